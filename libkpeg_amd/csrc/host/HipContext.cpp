#include "HipContext.hpp"

#include <cstdlib>
#include <mutex>

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
    }
}
