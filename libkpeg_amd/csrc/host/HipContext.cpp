#include "HipContext.hpp"

#include <cstdlib>
#include <mutex>
#include <string>

namespace kpeg
{
    namespace hip
    {
        namespace
        {
            struct Holder
            {
                kpeg_hip_ctx* ctx = nullptr;
                std::string error;
                ~Holder() { if ( ctx ) kpeg_hip_destroy( ctx ); }
            };
        }

        kpeg_hip_ctx* context( std::string* why )
        {
            static Holder h;
            static std::once_flag once;
            std::call_once( once, [] {
                int dev = 0;
                if ( const char* e = std::getenv( "KPEG_HIP_DEVICE" ) )
                    dev = std::atoi( e );
                const int rc = kpeg_hip_create( &h.ctx, dev );
                if ( rc != KPEG_HIP_OK )
                {
                    h.ctx = nullptr;
                    h.error = std::string( "kpeg_hip_create failed: " ) + kpeg_hip_strerror( rc ) +
                              " (this library has no CPU decode path; a gfx950 GPU is required)";
                }
            } );
            if ( !h.ctx && why )
                *why = h.error;
            return h.ctx;
        }

        namespace
        {
            Image* g_resident = nullptr;
            std::mutex g_resident_mutex;
        }

        void claimResident( Image* owner )
        {
            std::lock_guard<std::mutex> lock( g_resident_mutex );
            if ( g_resident && g_resident != owner )
                g_resident->materialise();   // the device buffer is about to be reused
            g_resident = owner;
        }

        void releaseResident( Image* owner )
        {
            std::lock_guard<std::mutex> lock( g_resident_mutex );
            if ( g_resident == owner )
                g_resident = nullptr;
        }

        const std::vector<kpeg_hip_ctx*>& contexts( std::string* why )
        {
            struct Many
            {
                std::vector<kpeg_hip_ctx*> all;
                std::string error;
                ~Many() { for ( std::size_t i = 1; i < all.size(); ++i ) kpeg_hip_destroy( all[i] ); }   // [0] belongs to context()
            };
            static Many m;
            static std::once_flag once;
            std::call_once( once, [] {
                kpeg_hip_ctx* first = context( &m.error );
                if ( !first )
                    return;
                m.all.push_back( first );
                int n = 1, dev0 = 0;
                if ( const char* e = std::getenv( "KPEG_HIP_DEVICES" ) )
                    n = std::atoi( e );
                if ( const char* e = std::getenv( "KPEG_HIP_DEVICE" ) )
                    dev0 = std::atoi( e );
                for ( int i = 1; i < n; ++i )
                {
                    kpeg_hip_ctx* c = nullptr;
                    const int rc = kpeg_hip_create( &c, dev0 + i );
                    if ( rc != KPEG_HIP_OK )
                    {
                        m.error = std::string( "kpeg_hip_create failed on device " ) + std::to_string( dev0 + i ) + ": " + kpeg_hip_strerror( rc );
                        for ( std::size_t k = 1; k < m.all.size(); ++k ) kpeg_hip_destroy( m.all[k] );
                        m.all.clear();
                        return;
                    }
                    m.all.push_back( c );
                }
            } );
            if ( m.all.empty() && why )
                *why = m.error;
            return m.all;
        }
    }
}
