// C shim over the C++ host API (include/kpeg_host.h).
#include "../../../include/kpeg_host.h"

#include <cstring>
#include <fstream>
#include <string>

#include "Decoder.hpp"
#include "HuffmanTree.hpp"
#include "Image.hpp"
#include "Logger.hpp"
#include "Utility.hpp"

#ifndef KPEG_SRC_HASH
#define KPEG_SRC_HASH "unstamped"
#endif
extern "C" const char* kpeg_host_build_hash( void )
{
    static const char stamp[] = "KPEG_SRC_HASH=" KPEG_SRC_HASH;
    return stamp + 14;
}

extern "C" int kpeg_host_parse( const uint8_t* file, size_t size, unsigned flags, kpeg_frame* frame, uint8_t* scan, size_t scan_cap,
                                size_t* scan_len )
{
    if ( !file || !frame || !scan_len )
        return -1;
    try
    {
        kpeg::JPEGDecoder dec;
        dec.setParseOnly( true );
        dec.setRestartMarkerSupport( ( flags & KPEG_PARSE_ALLOW_DRI ) != 0 );
        dec.setGrayscaleSupport( ( flags & KPEG_PARSE_ALLOW_GRAY ) != 0 );
        dec.setAnySizeSupport( ( flags & KPEG_PARSE_ALLOW_ANY_SIZE ) != 0 );
        dec.set420Support( ( flags & KPEG_PARSE_ALLOW_420 ) != 0 );
        dec.openMemory( file, size, "memory.jpg" );
        const int rc = (int)dec.decodeImageFile();
        if ( rc != (int)kpeg::JPEGDecoder::DECODE_DONE )
            return rc;
        if ( !dec.frameInfo( frame ) )
            return -1;
        // what decodeScanData() answers for such a file without the extension (the parse alone does not look at the size)
        if ( !( flags & KPEG_PARSE_ALLOW_ANY_SIZE ) && frame->components != KPEG_FRAME_420 && ( ( frame->width & 7 ) || ( frame->height & 7 ) ) )
            return (int)kpeg::JPEGDecoder::ERROR;
        const std::vector<kpeg::UInt8>& s = dec.scanData();
        *scan_len = s.size();
        if ( scan && s.size() > scan_cap )
            return -1;
        if ( scan && !s.empty() )
            std::memcpy( scan, s.data(), s.size() );
        return rc;
    }
    catch ( ... )
    {
        return -1;
    }
}

extern "C" int kpeg_host_decode_file( const char* path, unsigned flags )
{
    try
    {
        kpeg::JPEGDecoder dec;
        dec.setRestartMarkerSupport( ( flags & KPEG_PARSE_ALLOW_DRI ) != 0 );
        dec.setGrayscaleSupport( ( flags & KPEG_PARSE_ALLOW_GRAY ) != 0 );
        dec.setAnySizeSupport( ( flags & KPEG_PARSE_ALLOW_ANY_SIZE ) != 0 );
        dec.set420Support( ( flags & KPEG_PARSE_ALLOW_420 ) != 0 );
        if ( !dec.open( path ) )
            return (int)kpeg::JPEGDecoder::ERROR;
        const int rc = (int)dec.decodeImageFile();
        if ( rc == (int)kpeg::JPEGDecoder::DECODE_DONE )
            dec.dumpRawData();
        return rc;
    }
    catch ( ... )
    {
        return (int)kpeg::JPEGDecoder::ERROR;
    }
}

// test hook (tests/test_gpu_decode.py): what holders of a decoded image see after the shared context has decoded something else.
// Decodes `path_a`, takes (1) a COPY of the decoder's image while its pixels are still on the GPU and (2) writes the PPM (which
// leaves the lazy source in place), decodes `path_b` with another decoder, then copies the pixels the copy and the first
// decoder's own image hold into out_copy / out_own (cap bytes each).  Returns the number of pixel bytes of image A, 0 on failure.
extern "C" size_t kpeg_host_test_image_holders( const char* path_a, const char* path_b, uint8_t* out_copy, uint8_t* out_own, size_t cap )
{
    try
    {
        kpeg::JPEGDecoder a;
        if ( !a.open( path_a ) || a.decodeImageFile() != kpeg::JPEGDecoder::DECODE_DONE )
            return 0;
        kpeg::Image copy = a.image();   // (must own its pixels from here on)
        a.dumpRawData();
        kpeg::Image assigned;
        assigned = a.image();
        {
            kpeg::JPEGDecoder b;
            if ( !b.open( path_b ) || b.decodeImageFile() != kpeg::JPEGDecoder::DECODE_DONE )
                return 0;
            b.dumpRawData();
        }
        const std::vector<kpeg::UInt8>& c = copy.getRGB8();
        const std::vector<kpeg::UInt8>& o = a.image().getRGB8();
        if ( c.size() > cap || o.size() != c.size() || assigned.getRGB8() != c )
            return 0;
        std::memcpy( out_copy, c.data(), c.size() );
        std::memcpy( out_own, o.data(), o.size() );
        return c.size();
    }
    catch ( ... )
    {
        return 0;
    }
}

extern "C" size_t kpeg_host_restart_offsets( const uint8_t* scan, size_t n, uint64_t* offsets, size_t cap )
{
    size_t k = 0;
    for ( size_t i = 0; i + 1 < n; ++i )
        if ( scan[i] == 0xFF && scan[i + 1] >= 0xD0 && scan[i + 1] <= 0xD7 )
        {
            if ( offsets && k < cap )
                offsets[k] = i;
            ++k;
            ++i;
        }
    return k;
}

extern "C" int kpeg_host_huffman_contains( const uint8_t counts[16], const uint8_t* symbols, const char* bits, char* out, size_t cap )
{
    kpeg::HuffmanTable t;
    int k = 0;
    for ( int i = 0; i < 16; ++i )
    {
        t[i].first = counts[i];
        t[i].second.assign( symbols + k, symbols + k + counts[i] );
        k += counts[i];
    }
    kpeg::HuffmanTree tree( t );
    const std::string r = tree.contains( bits );
    if ( r.size() + 1 > cap )
        return -1;
    std::memcpy( out, r.c_str(), r.size() + 1 );
    return (int)r.size();
}

extern "C" int kpeg_host_bitstring_to_value( const char* bits ) { return kpeg::bitStringtoValue( bits ); }

extern "C" int kpeg_host_value_to_bitstring( int value, char* out, size_t cap )
{
    const std::string r = kpeg::valueToBitString( (kpeg::Int16)value );
    if ( r.size() + 1 > cap )
        return -1;
    std::memcpy( out, r.c_str(), r.size() + 1 );
    return (int)r.size();
}

extern "C" int kpeg_host_is_valid_filename( const char* name ) { return kpeg::isValidFilename( name ) ? 1 : 0; }
