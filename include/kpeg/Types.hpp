// include/kpeg/Types.hpp -- basic types of the kpeg API (mirrors the reference's
// include/Types.hpp:12-126 so that code written against libKPEG compiles unchanged).
#ifndef KPEG_TYPES_HPP
#define KPEG_TYPES_HPP

#include <array>
#include <memory>
#include <utility>
#include <vector>

namespace kpeg
{
    typedef unsigned char  UInt8;
    typedef unsigned short UInt16;
    typedef unsigned int   UInt32;
    typedef char  Int8;
    typedef short Int16;
    typedef int   Int32;

    enum Components { COMP1, COMP2, COMP3 };
    enum RGBComponents { RED, GREEN, BLUE };
    enum YCbCrComponents { Y, Cb, Cr };

    /// Pixel with three integer components (reference Types.hpp:52-76).
    struct Pixel
    {
        Pixel() { comp[0] = comp[1] = comp[2] = 0; }
        Pixel( const Int16 c1, const Int16 c2, const Int16 c3 ) { comp[0] = c1; comp[1] = c2; comp[2] = c3; }
        Int16 comp[3];
    };

    /// Pixel with three float components (reference Types.hpp:85-109).
    struct FPixel
    {
        FPixel() { comp[0] = comp[1] = comp[2] = 0.f; }
        FPixel( const float c1, const float c2, const float c3 ) { comp[0] = c1; comp[1] = c2; comp[2] = c3; }
        float comp[3];
    };

    typedef std::shared_ptr<std::vector<std::vector<Pixel>>>  PixelPtr;
    typedef std::shared_ptr<std::vector<std::vector<FPixel>>> FPixelPtr;

    /// 16 entries, entry i = < number of codes of length i+1, their symbols > (Types.hpp:116)
    typedef std::array<std::pair<int, std::vector<UInt8>>, 16> HuffmanTable;

    const int HT_DC   = 0;
    const int HT_AC   = 1;
    const int HT_Y    = 0;
    const int HT_CbCr = 1;
}

#endif
