// libkpeg_amd/csrc/host/Batch.cpp -- kpeg::decodeFiles (include/kpeg/Batch.hpp): the batch front end.
#include "Batch.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstring>
#include <memory>

#include "../../../include/kpeg_hip.h"
#include "Decoder.hpp"
#include "HipContext.hpp"
#include "Logger.hpp"
#include "Utility.hpp"

namespace kpeg
{
    namespace
    {
        struct Item
        {
            std::unique_ptr<JPEGDecoder> dec;
            kpeg_frame frame;
            bool done = false;
        };

        bool isDirectory( const std::string& name )
        {
            struct stat st;
            return ::stat( name.c_str(), &st ) == 0 && S_ISDIR( st.st_mode );
        }

        void expand( const std::string& name, std::vector<std::string>& out )
        {
            if ( !isDirectory( name ) )
            {
                out.push_back( name );
                return;
            }
            std::vector<std::string> found;
            if ( DIR* d = ::opendir( name.c_str() ) )
            {
                while ( const dirent* e = ::readdir( d ) )
                {
                    const std::string f = e->d_name;
                    if ( isValidFilename( f ) )   // ends in ".jpg", as the single-file front end demands
                        found.push_back( name + ( name.empty() || name.back() == '/' ? "" : "/" ) + f );
                }
                ::closedir( d );
            }
            std::sort( found.begin(), found.end() );
            out.insert( out.end(), found.begin(), found.end() );
        }

        // one image through the single-image entry point
        bool decodeOne( kpeg_hip_ctx* ctx, Item& it )
        {
            const kpeg_frame& f = it.frame;
            std::vector<UInt8> rgb( (std::size_t)f.width * f.height * 3 );
            const std::vector<UInt8>& scan = it.dec->scanData();
            if ( kpeg_hip_decode_scan( ctx, &f, scan.data(), scan.size(), rgb.data() ) != KPEG_HIP_OK )
                return false;
            it.dec->image().adoptRGB8( std::move( rgb ) );
            return true;
        }
    }

    BatchResult decodeFiles( const std::vector<std::string>& names, bool allowDRI, bool allowGray, bool allowAnySize, bool allow420 )
    {
        BatchResult res;
        std::vector<std::string> files;
        for ( const std::string& n : names )
            expand( n, files );

        // 1. the marker parser, file by file (host)
        std::vector<Item> items;
        for ( const std::string& name : files )
        {
            if ( !isValidFilename( name ) )
            {
                LOG(Logger::Level::ERROR) << "Invalid input file name passed: " << name << std::endl;
                res.rejected++;
                continue;
            }
            Item it;
            it.dec.reset( new JPEGDecoder );
            it.dec->setRestartMarkerSupport( allowDRI );
            it.dec->setGrayscaleSupport( allowGray );
            it.dec->setAnySizeSupport( allowAnySize );
            it.dec->set420Support( allow420 );
            it.dec->setParseOnly( true );
            if ( !it.dec->open( name ) || it.dec->decodeImageFile() != JPEGDecoder::DECODE_DONE || !it.dec->decodable() ||
                 !it.dec->frameInfo( &it.frame ) )
            {
                LOG(Logger::Level::ERROR) << "Not decoded (rejected by the marker parser): " << name << std::endl;
                res.rejected++;
                continue;
            }
            items.push_back( std::move( it ) );
        }
        if ( items.empty() )
            return res;

        std::string why;
        kpeg_hip_ctx* ctx = hip::context( &why );
        hip::claimResident( nullptr );   // an image a JPEGDecoder left on the GPU is fetched before the batch reuses the buffers
        if ( !ctx )
        {
            LOG(Logger::Level::ERROR) << "[ FATAL ] " << why << std::endl;
            res.failed = items.size();
            return res;
        }

        // 2. groups of identical geometry and tables, in order of first appearance (kpeg_frame is plain data, zero-filled
        //    by frameInfo before it is set: memcmp compares values)
        std::vector<std::vector<std::size_t>> groups;
        for ( std::size_t i = 0; i < items.size(); ++i )
        {
            bool placed = false;
            for ( auto& g : groups )
                if ( std::memcmp( &items[g[0]].frame, &items[i].frame, sizeof( kpeg_frame ) ) == 0 )
                {
                    g.push_back( i );
                    placed = true;
                    break;
                }
            if ( !placed )
                groups.push_back( std::vector<std::size_t>( 1, i ) );
        }

        // 3. the GPU path, group by group
        for ( const auto& g : groups )
        {
            res.groups++;
            const kpeg_frame& f = items[g[0]].frame;
            const std::size_t bytes = (std::size_t)f.width * f.height * 3;
            bool batched = false;
            if ( g.size() > 1 )
            {
                std::vector<std::vector<UInt8>> rgbs( g.size(), std::vector<UInt8>( bytes ) );
                std::vector<const uint8_t*> scans( g.size() );
                std::vector<size_t> lens( g.size() );
                std::vector<uint8_t*> outs( g.size() );
                for ( std::size_t k = 0; k < g.size(); ++k )
                {
                    scans[k] = items[g[k]].dec->scanData().data();
                    lens[k] = items[g[k]].dec->scanData().size();
                    outs[k] = rgbs[k].data();
                }
                if ( kpeg_hip_decode_batch( ctx, (int)g.size(), &f, scans.data(), lens.data(), outs.data() ) == KPEG_HIP_OK )
                {
                    for ( std::size_t k = 0; k < g.size(); ++k )
                    {
                        items[g[k]].dec->image().adoptRGB8( std::move( rgbs[k] ) );
                        items[g[k]].done = true;
                    }
                    batched = true;
                }
                else
                {
                    // one corrupt stream fails the whole call: the group is decoded image by image, so that the others still get their PPM
                    LOG(Logger::Level::ERROR) << "Batch of " << g.size() << " failed (" << kpeg_hip_last_error( ctx )
                                              << "): decoding its images one by one" << std::endl;
                }
            }
            if ( !batched )
                for ( std::size_t i : g )
                    items[i].done = decodeOne( ctx, items[i] );
        }

        // 4. the PPM files
        for ( Item& it : items )
        {
            if ( !it.done )
            {
                LOG(Logger::Level::ERROR) << "[ FATAL ] GPU decode failed (corrupt entropy-coded data)" << std::endl;
                res.failed++;
                continue;
            }
            if ( it.dec->dumpRawData() )
                res.written++;
        }
        return res;
    }
}
