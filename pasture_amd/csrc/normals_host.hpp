// Host-side interface between the kNN driver (normals.hip) and the LDS box kernel (normals_tile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "normals_device.hpp"

namespace pstk {

struct TileShape { uint32_t bx = 0, by = 0, bz = 0, threads = 256, cap = 0; };
// false: no box fits (k > 32, or even a single query row with its halo exceeds the LDS budget at this density)
bool knn_tile_shape(const pstn::GridParams& g, uint64_t nf, uint64_t cells, uint32_t k, TileShape& t);
// Searches every query whose 5x5x5-cell neighbourhood fits the box kernel; the others are appended to fb_list / *fb_count
// (sorted indices) for knn_grid_kernel.  *fb_count must be zero on entry; fb_list must hold nf entries.
void launch_knn_tile(const TileShape& t, const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t k, uint32_t nf,
                     const pstn::RecOut& out, uint32_t* fb_list, uint32_t* fb_count, hipStream_t stream);

}  // namespace pstk
