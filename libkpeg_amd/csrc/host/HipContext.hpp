// Process-wide handle on the GPU path for the C++ host classes.
#pragma once
#include <string>
#include <vector>

#include "../../../include/kpeg_hip.h"
#include "Image.hpp"

namespace kpeg
{
    namespace hip
    {
        /// Lazily creates one kpeg_hip_ctx on device $KPEG_HIP_DEVICE (default 0).
        /// Returns nullptr and fills `why` if no gfx950 device / runtime is available.
        kpeg_hip_ctx* context( std::string* why );

        /// The contexts a restart-interval image is sharded over: $KPEG_HIP_DEVICES of them (default 1: no sharding),
        /// on devices $KPEG_HIP_DEVICE, +1, ...  The first one is context().  Empty (and `why` filled) on failure.
        const std::vector<kpeg_hip_ctx*>& contexts( std::string* why );

        /// The shared context's device buffer holds ONE decoded image.  An Image that left its pixels there registers
        /// itself before the decode; whoever decodes next makes the previous owner fetch its pixels first.
        void claimResident( Image* owner );
        void releaseResident( Image* owner );
    }
}
