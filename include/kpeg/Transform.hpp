// include/kpeg/Transform.hpp -- zig-zag index helpers (reference include/Transform.hpp:9-18).
#ifndef KPEG_TRANSFORM_HPP
#define KPEG_TRANSFORM_HPP

#include <utility>

namespace kpeg
{
    /// zig-zag position (0..63) -> (row, column) of the 8x8 matrix
    const std::pair<const int, const int> zzOrderToMatIndices( const int zzindex );

    /// (row, column) -> zig-zag position
    const int matIndicesToZZOrder( const int row, const int column );
}

#endif
