// include/kpeg/MCU.hpp -- one 8x8 pixel block with its three component matrices.
//
// Same surface as the reference's include/MCU.hpp:26-87.  In the reference this class IS the
// hot loop (constructMCU -> computeIDCT -> performLevelShift -> convertYCbCrToRGB on the
// CPU, src/MCU.cpp:64-279).  Here the RLE walk and the never-reset DC predictors
// (MCU::DCDiff, src/MCU.cpp:53,97-108) stay on the host, and the arithmetic of the block is
// done by the GPU path (kpeg_hip_idct_colour on a one-MCU image) -- there is no CPU
// implementation of the transform in this library.  JPEGDecoder does not go through this
// class: it hands the whole scan to the GPU at once.
#ifndef KPEG_MCU_HPP
#define KPEG_MCU_HPP

#include <array>
#include <utility>
#include <vector>

#include "Transform.hpp"
#include "Types.hpp"

namespace kpeg
{
    typedef std::array< std::array< std::array< int, 8 >, 8 >, 3 > CompMatrices;
    typedef std::array< std::array< int, 8 >, 8 > Matrix8x8;

    class MCU
    {
        public:
            MCU();
            MCU( const std::array<std::vector<int>, 3>& compRLE, const std::vector<std::vector<UInt16>>& QTables );

            /// Throws std::runtime_error if the GPU path is unavailable.
            void constructMCU( const std::array<std::vector<int>, 3>& compRLE, const std::vector<std::vector<UInt16>>& QTables );

            const CompMatrices& getAllMatrices() const;
            const Matrix8x8 getYMatrix() const;
            const Matrix8x8 getCbMatrix() const;
            const Matrix8x8 getCrMatrix() const;

            /// Builds an MCU from already decoded R, G, B values (row-major 8x8 each).
            static MCU fromRGB( const UInt8* rgb, std::size_t pitch );

        private:
            CompMatrices blocks_;            // after construction: R, G, B
            static int MCUCount_;
            static std::vector<std::vector<UInt16>> QTables_;
            static int DCDiff[3];               // never reset, as in the reference (quirk Q6)
    };
}

#endif
