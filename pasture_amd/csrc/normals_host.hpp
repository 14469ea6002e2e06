// Host-side interface between the kNN driver (normals.hip) and the LDS box kernel (normals_tile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "normals_device.hpp"

namespace pstk {

struct TileShape { uint32_t bx = 0, by = 0, bz = 0, threads = 256, cap = 0; char tag = '1'; };  // tag: which instance of the box kernel (normals_tile.hip)
// false: no box fits (k > 32, or even a single query row with its halo exceeds the LDS budget at this density)
// (measured: census kernels over the dense directory; scratch3 = 32 bytes of device memory; synchronises the stream)
// sink (nullable): every census also lists the boxes that hold a query; on success list / n describe the winning shape (list == null: none
// was written, e.g. a forced shape -- knn_box_list builds it then)
struct BoxListSink {
  std::function<uint32_t*(size_t)> alloc;  // device memory for one list (stays valid until the call ends)
  uint32_t* count_dev = nullptr;           // one device word
  const uint32_t* list = nullptr;
  uint32_t n = 0;
  const uint32_t* sorted_cells = nullptr;  // the points' cell numbers in sorted order (the index builder's keys), or null: a census of a grid
                                           // far larger than the cloud counts the queries of the boxes from these instead of the directory
};
bool knn_tile_shape(const pstn::GridParams& g, uint64_t nf, uint64_t cells, uint32_t k, bool volume_like, const uint32_t* cell_start,
                    unsigned long long* scratch3, hipStream_t stream, TileShape& t, BoxListSink* sink = nullptr);
// mean number of points within h / 2 and within h of a sampled point of the sorted cloud (synchronises the stream); false on failure
bool knn_probe(const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t nf, unsigned long long* scratch3, hipStream_t stream,
               double& mean_half, double& mean_full);
// Local scale without an index (normals_scale.hip): median radius at which the full cloud holds m_target points and median local dimension, from
// distance histograms of 512 sampled points against the subsample `cand` (n_c packed points = every `thinning`-th point of the cloud).
size_t knn_scale_scratch_bytes();
bool knn_scale_estimate(const double* cand, uint32_t n_c, double thinning, double diag2, double m_target, unsigned int* scratch, hipStream_t stream,
                        double& h_m, double& dim, uint32_t max_queries = 512);
// Searches every query whose 5x5x5-cell neighbourhood fits the box kernel; the others are appended to fb_list / *fb_count
// (sorted indices) for knn_grid_kernel.  *fb_count must be zero on entry; fb_list must hold nf entries.
// box_list / n_list: the boxes to search (knn_box_list: those that hold a query), or null: every box of the grid.
void launch_knn_tile(const TileShape& t, const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t k, uint32_t nf,
                     const pstn::RecOut& out, uint32_t* fb_list, uint32_t* fb_count, const uint32_t* box_list, uint32_t n_list, hipStream_t stream);
uint32_t knn_box_count(const TileShape& t, const pstn::GridParams& g);
uint32_t knn_box_list(const TileShape& t, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t* list, uint32_t* count_dev, hipStream_t stream);

}  // namespace pstk
