// include/kpeg/Markers.hpp -- JFIF marker bytes (second byte after 0xFF); the names follow
// the reference's include/Markers.hpp:10-80.
#ifndef KPEG_MARKERS_HPP
#define KPEG_MARKERS_HPP

#include "Types.hpp"

namespace kpeg
{
    const UInt8 JFIF_BYTE_0  = 0x00;
    const UInt8 JFIF_BYTE_FF = 0xFF;

    const UInt8 JFIF_SOF0 = 0xC0;  // baseline DCT
    const UInt8 JFIF_SOF1 = 0xC1;
    const UInt8 JFIF_SOF2 = 0xC2;
    const UInt8 JFIF_SOF3 = 0xC3;
    const UInt8 JFIF_DHT  = 0xC4;
    const UInt8 JFIF_SOF5 = 0xC5;
    const UInt8 JFIF_SOF6 = 0xC6;
    const UInt8 JFIF_SOF7 = 0xC7;
    const UInt8 JFIF_SOF9 = 0xC9;
    const UInt8 JFIF_SOF10 = 0xCA;
    const UInt8 JFIF_SOF11 = 0xCB;
    const UInt8 JFIF_DAC  = 0xCC;
    const UInt8 JFIF_SOF13 = 0xCD;
    const UInt8 JFIF_SOF14 = 0xCE;
    const UInt8 JFIF_SOF15 = 0xCF;

    const UInt8 JFIF_RST0 = 0xD0;
    const UInt8 JFIF_RST1 = 0xD1;
    const UInt8 JFIF_RST2 = 0xD2;
    const UInt8 JFIF_RST3 = 0xD3;
    const UInt8 JFIF_RST4 = 0xD4;
    const UInt8 JFIF_RST5 = 0xD5;
    const UInt8 JFIF_RST6 = 0xD6;
    const UInt8 JFIF_RST7 = 0xD7;

    const UInt8 JFIF_SOI  = 0xD8;
    const UInt8 JFIF_EOI  = 0xD9;
    const UInt8 JFIF_SOS  = 0xDA;
    const UInt8 JFIF_DQT  = 0xDB;
    const UInt8 JFIF_DNL  = 0xDC;
    const UInt8 JFIF_DRI  = 0xDD;
    const UInt8 JFIF_DHP  = 0xDE;
    const UInt8 JFIF_EXP  = 0xDF;

    const UInt8 JFIF_APP0 = 0xE0;
    const UInt8 JFIF_APP1 = 0xE1;
    const UInt8 JFIF_APP15 = 0xEF;

    const UInt8 JFIF_COM  = 0xFE;
}

#endif
