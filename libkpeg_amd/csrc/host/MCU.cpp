// kpeg::MCU -- host bookkeeping of one block; the arithmetic runs on the GPU.
//
// constructMCU follows the reference's src/MCU.cpp:64-150 for what stays on the host:
//   * the RLE walk stops at the first (0,0) pair, the DC pair included (quirk Q1, :97-100),
//   * DCDiff[c] accumulates across calls and is never reset (:53, :107-108, quirk Q6),
//   * component 0 uses quantiser table 0, components 1 and 2 table 1 (:110).
// Dequantisation, IDCT, level shift and colour (:110-279) are done by kpeg_hip_idct_colour
// on a one-MCU image.
#include "MCU.hpp"

#include <cstring>
#include <stdexcept>

#include "HipContext.hpp"
#include "Logger.hpp"

namespace kpeg
{
    int MCU::MCUCount_ = 0;
    std::vector<std::vector<UInt16>> MCU::QTables_ = {};
    int MCU::DCDiff[3] = { 0, 0, 0 };

    MCU::MCU()
    {
        for ( auto& c : blocks_ )
            for ( auto& r : c )
                r.fill( 0 );
    }

    MCU::MCU( const std::array<std::vector<int>, 3>& compRLE, const std::vector<std::vector<UInt16>>& QTables )
    {
        constructMCU( compRLE, QTables );
    }

    void MCU::constructMCU( const std::array<std::vector<int>, 3>& compRLE, const std::vector<std::vector<UInt16>>& QTables )
    {
        QTables_ = QTables;
        MCUCount_++;
        if ( QTables.size() < 2 || QTables[0].size() < 64 || QTables[1].size() < 64 )
            throw std::runtime_error( "kpeg::MCU: two 64-entry quantisation tables are required" );

        int16_t coef[3][64];  // natural order
        std::memset( coef, 0, sizeof( coef ) );
        for ( int c = 0; c < 3; ++c )
        {
            int zz[64] = { 0 };
            int j = -1;
            const std::vector<int>& rle = compRLE[c];
            for ( std::size_t i = 0; i + 1 < rle.size(); i += 2 )
            {
                if ( rle[i] == 0 && rle[i + 1] == 0 )
                    break;
                j += rle[i] + 1;
                if ( j > 63 )
                    throw std::runtime_error( "kpeg::MCU: run-length data overruns the block" );
                zz[j] = rle[i + 1];
            }
            DCDiff[c] += zz[0];
            zz[0] = DCDiff[c];
            for ( int k = 0; k < 64; ++k )
            {
                auto rc = zzOrderToMatIndices( k );
                coef[c][rc.first * 8 + rc.second] = (int16_t)zz[k];
            }
        }

        std::string why;
        kpeg_hip_ctx* ctx = hip::context( &why );
        if ( !ctx )
            throw std::runtime_error( "kpeg::MCU: " + why );
        kpeg_frame f;
        std::memset( &f, 0, sizeof( f ) );
        f.width = f.height = 8;
        for ( int t = 0; t < 2; ++t )
            for ( int k = 0; k < 64; ++k )
                f.qt[t][k] = QTables[t][k];
        UInt8 rgb[192];
        const int rc = kpeg_hip_idct_colour( ctx, &f, &coef[0][0], rgb );
        if ( rc != KPEG_HIP_OK )
            throw std::runtime_error( std::string( "kpeg::MCU: GPU path failed: " ) + kpeg_hip_strerror( rc ) + ": " +
                                      kpeg_hip_last_error( ctx ) );
        *this = fromRGB( rgb, 24 );
    }

    MCU MCU::fromRGB( const UInt8* rgb, std::size_t pitch )
    {
        MCU m;
        for ( int r = 0; r < 8; ++r )
            for ( int x = 0; x < 8; ++x )
                for ( int c = 0; c < 3; ++c )
                    m.blocks_[c][r][x] = rgb[r * pitch + x * 3 + c];
        return m;
    }

    const CompMatrices& MCU::getAllMatrices() const { return blocks_; }
    const Matrix8x8 MCU::getYMatrix() const { return blocks_[0]; }
    const Matrix8x8 MCU::getCbMatrix() const { return blocks_[1]; }
    const Matrix8x8 MCU::getCrMatrix() const { return blocks_[2]; }
}
