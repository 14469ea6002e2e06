// Host-side interface between the kNN driver (normals.hip) and the LDS box kernel (normals_tile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "normals_device.hpp"

namespace pstk {

struct TileShape { uint32_t bx = 0, by = 0, bz = 0, threads = 256, cap = 0; char tag = '1'; bool fit_seq = false; };  // tag: which instance of the box kernel (normals_tile.hip); fit_seq: its plane fit in the reference's order
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
// n_list_dev (nullable, with a box list): the list's length lives on the device and n_list is the capacity the grid is sized for
void launch_knn_tile(const TileShape& t, const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t k, uint32_t nf,
                     const pstn::RecOut& out, uint32_t* fb_list, uint32_t* fb_count, const uint32_t* box_list, uint32_t n_list, hipStream_t stream,
                     const uint32_t* n_list_dev = nullptr);
bool knn_box_list_async(const TileShape& t, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t* list, uint32_t* count_dev, hipStream_t stream);
uint32_t knn_box_count(const TileShape& t, const pstn::GridParams& g);
uint32_t knn_box_list(const TileShape& t, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t* list, uint32_t* count_dev, hipStream_t stream);

// What one synchronous compute_normals call decided, for clouds that took the LDS box search and left no query open: enough to run the same
// pipeline again -- on this cloud or on another of the same length and shape -- without measuring anything and without a host round trip
// (run_normals_replay).  The grid (frame, box, cell edges) is FIXED by the record: points outside it are clamped into boundary cells and
// their queries go to the exact search, so a replay is exact for any data; only its capacities (occupied boxes, hand-back list) can be
// exceeded, which the status word reports.
struct KnnPlanRecord {
  bool valid = false;
  uint64_t n = 0, nf = 0, cells = 0;
  uint32_t k = 0;
  unsigned key_bits = 0;
  pstn::GridParams g{};
  TileShape shape{};
  bool use_list = false;         // one workgroup per box that holds a query (clouds that do not fill their box)
  uint32_t n_list = 0, n_fb = 0; // what the recorded call saw: boxes listed, queries handed to the exact search
  const char* why_not = "";      // !valid: the reason
};
enum : uint32_t { KNN_STATUS_FINITE_COUNT = 1, KNN_STATUS_BOX_CAPACITY = 2, KNN_STATUS_FALLBACK_CAPACITY = 4, KNN_STATUS_OPEN_QUERIES = 8, KNN_STATUS_DEGENERATE = 16 };

}  // namespace pstk
