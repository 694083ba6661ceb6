// C shim over the C++ host API (include/kpeg_host.h).
#include "../../../include/kpeg_host.h"

#include <cstring>
#include <fstream>
#include <string>

#include "Decoder.hpp"
#include "Logger.hpp"

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
        dec.openMemory( file, size, "memory.jpg" );
        const int rc = (int)dec.decodeImageFile();
        if ( rc != (int)kpeg::JPEGDecoder::DECODE_DONE )
            return rc;
        if ( !dec.frameInfo( frame ) )
            return -1;
        const std::vector<kpeg::UInt8>& s = dec.scanData();
        *scan_len = s.size();
        if ( scan && s.size() <= scan_cap )
            std::memcpy( scan, s.data(), s.size() );
        else if ( scan )
            return -1;
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
