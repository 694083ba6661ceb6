// Process-wide handle on the GPU path for the C++ host classes.
#pragma once
#include <string>

#include "../../../include/kpeg_hip.h"

namespace kpeg
{
    namespace hip
    {
        /// Lazily creates one kpeg_hip_ctx on device $KPEG_HIP_DEVICE (default 0).
        /// Returns nullptr and fills `why` if no gfx950 device / runtime is available.
        kpeg_hip_ctx* context( std::string* why );
    }
}
