// K4 — kNN normal estimation on the device (gfx950): spatial index, global-memory search kernels and the host driver.
//
// Replaces compute_normals (pasture-algorithms/src/normal_estimation.rs:79-130): per point the k nearest neighbours
// (exact, f64 squared Euclidean distance, the point itself included, ascending distance — the contract of
// kd-tree 0.3.0 `nearests`, normal_estimation.rs:103-108), the UN-normalised covariance of the neighbourhood
// (:240-305), the closed-form eigen solve (:308-453) and the plane parameters (:456-467), reproduced quirk for quirk
// (eigenvalues of the unscaled matrix multiplied by the scale again :441-443; the diagonal subtraction :446-449 that
// has no effect; normal = largest of three row cross products, NOT normalised :395-426).
//
// Spatial index: points are binned into a uniform grid whose cell edge is chosen so that a sphere of one cell edge holds about k
// points, radix-sorted by cell key (radix_sort.hip) and reordered once, so that every search reads
// contiguous runs of sorted points.
//   volume-like clouds (cells <= 4 n): keys are row-major cell numbers (x fastest, 32-bit) with a DENSE directory cell_start[];
//       the search runs in normals_tile.hip (a box of cells staged in LDS per workgroup); queries it cannot finish come back as a list
//       for knn_grid_kernel<K, true> below;
//   sparse clouds: 63-bit Morton keys + an open-addressing hash table of occupied cells; knn_grid_kernel<K, false>: one lane per query
//       walks Chebyshev shells of cells, k best candidates SORTED IN REGISTERS, stops when the k-th distance is inside the searched cube.
// Search kernels write ONE aligned 32-byte result record per point at its original index (full-sector stores behind a compute-bound
// search); split_results_kernel streams the records into the caller's outputs and narrows the normal to the NORMAL attribute (Vec3f32,
// Rust `as`).
// The reference allocates a HashMapBuffer per point and goes through DMatrix; none of that survives: the 3x3 moment
// sums live in registers.  f64 throughout (sqrt / atan2 / cos / sin from the device math library); -ffp-contract=off.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <memory>
#include <vector>

#include "device_sort.hpp"
#include "kernels.hpp"
#include "normals_device.hpp"
#include "normals_host.hpp"
#include "normals_plan.hpp"

using namespace pstn;

namespace {

__device__ __forceinline__ uint64_t spread21(uint64_t v) {  // insert two zero bits between each of the low 21 bits
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) { return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2); }

// positions (any stride) -> packed xyz f64 + finite-only bounds partials
__global__ __launch_bounds__(kBlock) void gather_positions_kernel(const uint8_t* base, uint64_t stride, uint64_t n, double* __restrict__ xyz,
                                                                  double* __restrict__ partials) {
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    cgptr_t p = (cgptr_t)(uint64_t)base + i * stride;
    const double x = load_un<double>(p), y = load_un<double>(p + 8), z = load_un<double>(p + 16);
    if (xyz) { xyz[3 * i] = x; xyz[3 * i + 1] = y; xyz[3 * i + 2] = z; }  // null: the source already is a packed Vec3f64 array
    if (finite3(x, y, z)) {
      mn[0] = __builtin_fmin(mn[0], x); mx[0] = __builtin_fmax(mx[0], x);
      mn[1] = __builtin_fmin(mn[1], y); mx[1] = __builtin_fmax(mx[1], y);
      mn[2] = __builtin_fmin(mn[2], z); mx[2] = __builtin_fmax(mx[2], z);
    }
  }
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* o = partials + (uint64_t)blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
}

// Does the cloud fill its bounding box?  A 32^3 occupancy bitmask (per-block in LDS, OR-ed into global memory): the LDS box search is
// laid out for volume-like clouds; a surface in a 3-D box (a few percent of the coarse cells occupied) keeps the global-memory search.
constexpr uint32_t kAxisBins = 256;  // slices per axis of the point-count histograms behind the trimmed box
constexpr uint32_t kOccBins = 32, kOccWords = kOccBins * kOccBins * kOccBins / 32;
// Every stride_pts-th point is looked at (n = the number of those): the 32^3 bitmask and the tail masses of the trimmed box are statistics -- 2^23
// points say what 10^8 do, and a surface with strays takes this pass three times.
__global__ __launch_bounds__(kBlock) void occupancy_kernel(const double* __restrict__ xyz, uint64_t n, uint64_t stride_pts, double ox, double oy, double oz, double sx, double sy,
                                                           double sz, uint32_t* __restrict__ bits, double ax, double ay, double az, uint32_t* __restrict__ axis_hist,
                                                           GridParams frame) {
  __shared__ uint32_t local[kOccWords];
  __shared__ uint32_t hist[3 * kAxisBins];  // points per slice of the box along every axis (ax/ay/az = slices per unit; outside -> end slices)
  for (uint32_t i = threadIdx.x; i < kOccWords; i += kBlock) local[i] = 0;
  for (uint32_t i = threadIdx.x; i < 3 * kAxisBins; i += kBlock) hist[i] = 0;
  __syncthreads();
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    const uint64_t p = i * stride_pts;
    const double x0 = xyz[3 * p], y0 = xyz[3 * p + 1], z0 = xyz[3 * p + 2];
    if (!finite3(x0, y0, z0)) continue;
    double x, y, z;  // in the grid's frame
    grid_frame(frame, x0, y0, z0, x, y, z);
    const double fx = (x - ox) * sx, fy = (y - oy) * sy, fz = (z - oz) * sz;
    auto slice = [](double t) { return (uint32_t)(t < 0.0 ? 0.0 : t > (double)(kAxisBins - 1) ? (double)(kAxisBins - 1) : t); };
    atomicAdd(&hist[slice((x - ox) * ax)], 1u);
    atomicAdd(&hist[kAxisBins + slice((y - oy) * ay)], 1u);
    atomicAdd(&hist[2 * kAxisBins + slice((z - oz) * az)], 1u);
    if (fx < 0.0 || fy < 0.0 || fz < 0.0 || fx >= (double)kOccBins || fy >= (double)kOccBins || fz >= (double)kOccBins) continue;  // outside a trimmed box
    const uint32_t bx = min((uint32_t)fx, kOccBins - 1), by = min((uint32_t)fy, kOccBins - 1), bz = min((uint32_t)fz, kOccBins - 1);
    const uint32_t bit = (bz * kOccBins + by) * kOccBins + bx;
    atomicOr(&local[bit >> 5], 1u << (bit & 31u));
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kOccWords; i += kBlock) if (local[i]) atomicOr(&bits[i], local[i]);
  for (uint32_t i = threadIdx.x; i < 3 * kAxisBins; i += kBlock) if (hist[i]) atomicAdd(&axis_hist[i], hist[i]);
}

// First and second moments of the points of a subsample that lie inside a box (the principal axes of the cloud without its far points):
// sums[0..2] = sum x, y, z; sums[3..8] = sum xx, xy, xz, yy, yz, zz (about `centre`, which keeps the sums small); sums[9] = count.
__global__ __launch_bounds__(kBlock) void moments_kernel(const double* __restrict__ xyz, uint64_t n, uint64_t stride_pts, double bx0, double by0, double bz0, double bx1,
                                                         double by1, double bz1, double cx, double cy, double cz, double* __restrict__ sums) {
  double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    const uint64_t p = i * stride_pts;
    const double x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
    if (!finite3(x, y, z) || x < bx0 || x > bx1 || y < by0 || y > by1 || z < bz0 || z > bz1) continue;
    const double dx = x - cx, dy = y - cy, dz = z - cz;
    acc[0] += dx; acc[1] += dy; acc[2] += dz;
    acc[3] += dx * dx; acc[4] += dx * dy; acc[5] += dx * dz; acc[6] += dy * dy; acc[7] += dy * dz; acc[8] += dz * dz;
    acc[9] += 1.0;
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    double v = acc[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_any(v, off);
    if ((threadIdx.x & 63u) == 0 && v != 0.0) atomicAdd(&sums[q], v);
  }
}

// bounds, in the frame `f`, of the finite points inside an axis-aligned box (partials: six doubles per block, like gather_positions_kernel)
__global__ __launch_bounds__(kBlock) void framed_bounds_kernel(const double* __restrict__ xyz, uint64_t n, double bx0, double by0, double bz0, double bx1, double by1,
                                                               double bz1, GridParams f, double* __restrict__ partials) {
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!finite3(x, y, z) || x < bx0 || x > bx1 || y < by0 || y > by1 || z < bz0 || z > bz1) continue;
    double u, v, w;
    grid_frame(f, x, y, z, u, v, w);
    mn[0] = __builtin_fmin(mn[0], u); mx[0] = __builtin_fmax(mx[0], u);
    mn[1] = __builtin_fmin(mn[1], v); mx[1] = __builtin_fmax(mx[1], v);
    mn[2] = __builtin_fmin(mn[2], w); mx[2] = __builtin_fmax(mx[2], w);
  }
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* o = partials + (uint64_t)blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
}

// The points are walked in the radix sort's tiles (walk.tile_size consecutive points per workgroup and step); with walk.counts the digit
// histogram of the sort's first pass is counted on the way (radix_sort.hip: counts[digit][tile]).  idx == nullptr: the sort numbers the points.
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void keys_kernel(const double* __restrict__ xyz, uint64_t n, GridParams g, KeyT* __restrict__ keys,
                                                      uint32_t* __restrict__ idx, unsigned long long* __restrict__ n_finite, pstk::RadixFirstPass walk) {
  __shared__ uint32_t hist[512];
  unsigned long long local = 0;
  const uint32_t mask = walk.counts ? (1u << walk.bits) - 1u : 0u, steps = walk.tile_size / kBlock;
  for (uint64_t tile = blockIdx.x; tile < walk.tiles; tile += gridDim.x) {
    if (walk.counts) {
      for (uint32_t d = threadIdx.x; d <= mask; d += kBlock) hist[d] = 0;
      __syncthreads();
    }
    constexpr uint32_t U = 4;  // points per thread in flight (tile_size is a multiple of U * kBlock)
    for (uint32_t it = 0; it < steps; it += U) {
      const uint64_t i0 = tile * walk.tile_size + (uint64_t)it * kBlock + threadIdx.x;
      if (i0 >= n) break;
      double px[U], py[U], pz[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kBlock < n ? i0 + (uint64_t)u * kBlock : i0;
        px[u] = xyz[3 * i]; py[u] = xyz[3 * i + 1]; pz[u] = xyz[3 * i + 2];
      }
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kBlock;
        if (i >= n) break;
        const double x = px[u], y = py[u], z = pz[u];
        uint64_t key = kInvalidKey;
        if (finite3(x, y, z)) {
          double uu, v, w;
          grid_frame(g, x, y, z, uu, v, w);
          const uint32_t cx = cell_coord(uu, g.org[0], g.inv_hx, g.dim[0]), cy = cell_coord(v, g.org[1], g.inv_h, g.dim[1]),
                         cz = cell_coord(w, g.org[2], g.inv_h, g.dim[2]);
          key = g.dense ? ((uint64_t)cz * g.dim[1] + cy) * g.dim[0] + cx : morton3(cx, cy, cz);
          local += 1;
        } else if (g.dense) {
          key = (uint64_t)g.dim[0] * g.dim[1] * g.dim[2];  // one past the last cell: non-finite points sort to the end
        }
        keys[i] = (KeyT)key;
        if (idx) idx[i] = (uint32_t)i;
        if (walk.counts) atomicAdd(&hist[(uint32_t)key & mask], 1u);
      }
    }
    if (walk.counts) {
      __syncthreads();
      for (uint32_t d = threadIdx.x; d <= mask; d += kBlock) walk.counts[(uint64_t)d * walk.tiles + tile] = hist[d];
      __syncthreads();
    }
  }
  // ONE atomic per workgroup, from a launch of at most a few thousand workgroups: atomics on one address are served one after the other
  // (~10 ns each) -- one per wave of 12 207 workgroups was 0.5 of this kernel's 0.73 ms, of 97 656 workgroups 4 ms
  __shared__ unsigned long long wave_finite[kBlock / 64];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) local += shfl_xor_any(local, off);
  if ((threadIdx.x & 63u) == 0) wave_finite[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) sum += wave_finite[w];
    if (sum) atomicAdd(n_finite, sum);
  }
}

template <int UNROLL>
__global__ __launch_bounds__(kBlock) void reorder_kernel(const double* __restrict__ xyz, const uint32_t* __restrict__ idx, uint64_t n,
                                                         double* __restrict__ sorted_xyz) {
  // one random 24-byte read per point (8-byte aligned): a 16-byte + an 8-byte load, the store side is lane-contiguous.  UNROLL points per
  // thread and step: their indices are read first, then all their gathers are in flight together (the kernel runs at the memory system's
  // random-request rate; PST_REORDER_UNROLL is the A/B switch)
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  const uint64_t step = (uint64_t)gridDim.x * kBlock * UNROLL;
  for (uint64_t j0 = (uint64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; j0 < n; j0 += step) {
    uint64_t i[UNROLL];
    d2u xy[UNROLL];
    double z[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { const uint64_t j = j0 + (uint64_t)u * kBlock; i[u] = j < n ? idx[j] : 0; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { xy[u] = *reinterpret_cast<const d2u*>(xyz + 3 * i[u]); z[u] = xyz[3 * i[u] + 2]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t j = j0 + (uint64_t)u * kBlock;
      if (j < n) { *reinterpret_cast<d2u*>(sorted_xyz + 3 * j) = xy[u]; sorted_xyz[3 * j + 2] = z[u]; }
    }
  }
}

// ---- cell hash table: key -> first sorted point of the cell -------------------------------------------------
struct CellTable {
  uint64_t* keys;    // kInvalidKey = empty
  uint32_t* starts;
  uint32_t mask;     // capacity - 1 (power of two)
};
__device__ __forceinline__ uint32_t hash_key(uint64_t k, uint32_t mask) {
  k *= 0x9E3779B97F4A7C15ull;
  return (uint32_t)(k >> 32) & mask;
}
__global__ __launch_bounds__(kBlock) void count_cells_kernel(const uint64_t* __restrict__ keys, uint64_t nf, unsigned long long* __restrict__ n_cells) {
  unsigned long long local = 0;
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < nf; j += step)
    if (j == 0 || keys[j] != keys[j - 1]) local += 1;
  if (local) atomicAdd(n_cells, local);
}
__global__ __launch_bounds__(kBlock) void build_table_kernel(const uint64_t* __restrict__ keys, uint64_t nf, CellTable t) {
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < nf; j += step) {
    const uint64_t k = keys[j];
    if (j != 0 && k == keys[j - 1]) continue;
    uint32_t slot = hash_key(k, t.mask);
    while (true) {
      const unsigned long long prev = atomicCAS((unsigned long long*)&t.keys[slot], (unsigned long long)kInvalidKey, (unsigned long long)k);
      if (prev == kInvalidKey) { t.starts[slot] = (uint32_t)j; break; }
      slot = (slot + 1) & t.mask;
    }
  }
}
__device__ __forceinline__ uint32_t lookup_cell(const CellTable& t, uint64_t k) {
  uint32_t slot = hash_key(k, t.mask);
  while (true) {
    const uint64_t kk = t.keys[slot];
    if (kk == k) return t.starts[slot];
    if (kk == kInvalidKey) return kNoIndex;
    slot = (slot + 1) & t.mask;
  }
}

// Dense directory (volume-like clouds: cells <= a few n): cell_start[c] = first sorted point with key >= c, c in [0, cells].
// Sorted keys are row-major cell numbers, so the cells x0..x1 of one grid row are ONE contiguous range of sorted points.
__global__ __launch_bounds__(kBlock) void build_directory_kernel(const uint32_t* __restrict__ keys, uint64_t nf, uint64_t cells,
                                                                 uint32_t* __restrict__ cell_start) {
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j <= nf; j += step) {
    // run head j (or the end sentinel j == nf): every cell number in (previous key, this key] starts here
    const uint64_t k = j < nf ? keys[j] : cells;
    uint64_t lo;
    if (j == 0) lo = 0;
    else {
      const uint64_t prev = keys[j - 1];
      if (prev == k) continue;
      lo = prev + 1;
    }
    for (uint64_t c = lo; c <= k; ++c) cell_start[c] = (uint32_t)j;
  }
}

// The same directory for clouds that leave most cells empty (a surface in a 3-D box: whole grid rows without a point; 1.9 * 10^9 cells for 10^8
// points of the LiDAR-like sheet).  Round 2 scattered the run heads into a 0xFFFFFFFF-filled array and took a suffix minimum over ALL cells
// (a 7.5 GB fill plus a 15 GB scan: 7 ms).  Now the directory is written exactly once, in blocks of kDirBlock cells:
//   dir_block_heads_kernel   first sorted point of every block that holds a run head (atomicMin into one word per block);
//   (suffix minimum over the BLOCK words: 1/1024 of the cells)  -> block_first[b] = first sorted point with key >= b * kDirBlock;
//   dir_fill_kernel          a workgroup per block: the block's points are the sorted range [block_first[b], block_first[b + 1]); an empty
//                            block writes that one value 1024 times, the others find every cell's first point by binary search in the range.
constexpr uint32_t kDirBlock = 1024;
__global__ __launch_bounds__(kBlock) void dir_block_heads_kernel(const uint32_t* __restrict__ keys, uint64_t nf, uint64_t cells, uint32_t* __restrict__ block_first) {
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j <= nf; j += step) {
    const uint64_t k = j < nf ? keys[j] : cells;  // (the end sentinel: cell_start[cells] = nf)
    // the FIRST point of a block: exactly one writer per block word, no atomics
    if (j == 0 || keys[j - 1] / kDirBlock != k / kDirBlock) block_first[k / kDirBlock] = (uint32_t)j;
  }
}
constexpr uint32_t kDirPerGroup = 32;
__global__ __launch_bounds__(kBlock) void dir_fill_kernel(const uint32_t* __restrict__ keys, uint64_t nf, uint64_t cells, const uint32_t* __restrict__ block_first,
                                                          uint64_t n_blocks, uint32_t* __restrict__ cell_start) {
  static_assert(kDirBlock == 4 * kBlock, "four cells per thread");
  __shared__ uint32_t cs[kDirBlock];
  __shared__ uint32_t wmin[kBlock / 64];
  __shared__ uint32_t bf[kDirPerGroup + 1];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // kDirPerGroup consecutive blocks of cells per workgroup (1.8 * 10^6 workgroups of 4 KB each were bound by their launches: 2.2 ms for 7.5 GB).
  // Round 6: the group's block words are fetched ONCE, up front (every round of the loop used to start with two dependent global loads: 7.5 GB of
  // directory left at 2.1 TB/s), 32 blocks per group, the cells leave with non-temporal 16-byte stores.
  const uint64_t b_first = (uint64_t)blockIdx.x * kDirPerGroup;
  if (tid <= kDirPerGroup) bf[tid] = b_first + tid < n_blocks ? block_first[b_first + tid] : (uint32_t)nf;
  __syncthreads();
  for (uint64_t b = b_first; b < n_blocks && b < b_first + kDirPerGroup; ++b) {
  const uint64_t c0 = b * kDirBlock;
  const uint32_t j0 = bf[b - b_first], j1 = bf[b - b_first + 1];
  uint32_t v[4];
  if (j0 == j1) {  // no point in this block: every cell starts at the next block's first point
    v[0] = v[1] = v[2] = v[3] = j0;
  } else {
    // the block's run heads land on their cells in LDS; a suffix minimum carries each head back over the empty cells in front of it
    __syncthreads();  // (the previous block of cells is done with cs / wmin)
    for (uint32_t t = tid; t < kDirBlock; t += kBlock) cs[t] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t j = j0 + tid; j < j1; j += kBlock) {
      const uint32_t k = keys[j];
      if (j == j0 || keys[j - 1] != k) cs[k - (uint32_t)c0] = j;
    }
    __syncthreads();
    uint32_t run = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 3; u >= 0; --u) { run = min(run, cs[4 * tid + (uint32_t)u]); v[u] = run; }
    uint32_t suf = run;  // minimum over this thread's cells and those of the higher lanes of its wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_down((int)suf, off, 64);
      if (lane + (uint32_t)off < 64u) suf = min(suf, o);
    }
    if (lane == 0) wmin[wave] = suf;
    __syncthreads();
    uint32_t right = j1;  // everything to the right of this thread's cells: higher lanes, higher waves, then the next block
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) if (w > wave) right = min(right, wmin[w]);
    const uint32_t hi = (uint32_t)__shfl_down((int)suf, 1, 64);
    if (lane < 63u) right = min(right, hi);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = min(v[u], right);
  }
  const uint64_t c = c0 + 4ull * tid;
  if (c + 3 <= cells) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u4{v[0], v[1], v[2], v[3]}, reinterpret_cast<u4*>(cell_start + c));
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (c + (uint64_t)u <= cells) cell_start[c + u] = v[u];
  }
  }
}

struct NormalsOut {
  double* normals_f64;    // [n][3] or null
  double* curvature_f64;  // [n] or null
  long long* knn;         // [n][k] int64 or null
  uint32_t* knn_u32;      // [n][k] uint32 (0xFFFFFFFF = no neighbour) or null
  uint64_t normal_attr;   // device address of the NORMAL (Vec3f32) attribute of point 0, or 0
  uint64_t normal_stride;
  uint64_t curv_attr;     // device address of the Curvature (F64) attribute of point 0, or 0
  uint64_t curv_stride;
  int* error_count;       // neighbourhoods with fewer than 3 usable points
};

__device__ __forceinline__ void write_result(const NormalsOut& o, uint64_t orig, const Fit& f) {
  if (!f.ok) { atomicAdd(o.error_count, 1); return; }
  if (o.normals_f64) { o.normals_f64[3 * orig] = f.nx; o.normals_f64[3 * orig + 1] = f.ny; o.normals_f64[3 * orig + 2] = f.nz; }
  if (o.curvature_f64) o.curvature_f64[orig] = f.curvature;
  if (o.normal_attr) {  // f64 -> f32 narrowing of the normal = Rust `as` (RNE, overflow -> inf)
    gptr_t p = as_global(o.normal_attr) + orig * o.normal_stride;
    store_un<float>(p, (float)f.nx); store_un<float>(p + 4, (float)f.ny); store_un<float>(p + 8, (float)f.nz);
  }
  if (o.curv_attr) store_un<double>(as_global(o.curv_attr) + orig * o.curv_stride, f.curvature);
}

// ---- brute force (tiny inputs, NaN-heavy inputs): exact, distance NaN counts as +inf, ties -> lower index first --
template <int K>
__global__ __launch_bounds__(kBlock) void knn_bruteforce_kernel(const double* __restrict__ xyz, uint32_t n, uint32_t k, NormalsOut out) {
  const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const double qx = xyz[3 * q], qy = xyz[3 * q + 1], qz = xyz[3 * q + 2];
  KBest<K> best;
  best.init();
  uint32_t filled = 0;
  // the query itself first (distance 0 when finite), then every other point in index order
  for (uint32_t pass = 0; pass < 2; ++pass) {
    for (uint32_t j = (pass == 0 ? q : 0); j < (pass == 0 ? q + 1 : n); ++j) {
      if (pass == 1 && j == q) continue;
      const double dx = xyz[3 * j] - qx, dy = xyz[3 * j + 1] - qy, dz = xyz[3 * j + 2] - qz;
      double d = dx * dx + dy * dy + dz * dz;
      if (d != d) d = __builtin_inf();
      if (d == __builtin_inf()) {  // +inf never wins `<`: take them only while the list is not full
        if (filled < k) {
          // append at the first free slot (slots beyond `filled` hold +inf / kNoIndex)
#pragma unroll
          for (int t = 0; t < K; ++t) if ((uint32_t)t == filled) best.i[t] = j;
          filled += 1;
        }
      } else {
        const double kth = best.kth(k);
        if (d < kth) {
          // drop the k-th entry by inserting into a list limited to k: entries >= k are never read
          best.insert(d, j);
          if (filled < k) filled += 1;
        }
      }
    }
  }
  const uint32_t m = n < k ? n : k;
  if (out.knn || out.knn_u32)
    for (uint32_t t = 0; t < k; ++t) {
      long long v = -1;
#pragma unroll
      for (int u = 0; u < K; ++u) if ((uint32_t)u == t && t < m) v = (long long)best.i[u];
      if (out.knn) out.knn[(uint64_t)q * k + t] = v;
      if (out.knn_u32) out.knn_u32[(uint64_t)q * k + t] = (uint32_t)v;
    }
  const Fit f = plane_fit(m, [&](uint32_t t, double& x, double& y, double& z) {
    uint32_t j = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) if ((uint32_t)u == t) j = best.i[u];
    x = xyz[3 * j]; y = xyz[3 * j + 1]; z = xyz[3 * j + 2];
  });
  write_result(out, q, f);
}
// ---- 32-byte result records (original point order) -> the caller's outputs ---------------------------------------------------------
// Pure streaming: record i is read once (two 16-byte loads), every output is written coalesced.
__global__ __launch_bounds__(kBlock) void split_results_kernel(const double* __restrict__ rec, uint64_t n, NormalsOut out) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rec + 4 * i));
    const d2 b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rec + 4 * i + 2));
    if (out.normals_f64) { out.normals_f64[3 * i] = a.x; out.normals_f64[3 * i + 1] = a.y; out.normals_f64[3 * i + 2] = b.x; }
    if (out.curvature_f64) out.curvature_f64[i] = b.y;
    if (out.normal_attr) {  // f64 -> f32 narrowing of the normal = Rust `as` (RNE, overflow -> inf)
      gptr_t p = as_global(out.normal_attr) + i * out.normal_stride;
      store_un<float>(p, (float)a.x); store_un<float>(p + 4, (float)a.y); store_un<float>(p + 8, (float)b.x);
    }
    if (out.curv_attr) store_un<double>(as_global(out.curv_attr) + i * out.curv_stride, b.y);
  }
}

// sorted positions (of the CURRENT index) of the queries flagged as unresolved by original index; the flags are cleared on the way
__global__ __launch_bounds__(kBlock) void collect_unresolved_kernel(const uint32_t* __restrict__ sidx, uint32_t nf, uint8_t* __restrict__ flag, uint8_t which,
                                                                    uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= nf) return;
  const uint32_t o = sidx[j];
  if (flag[o] == which) { flag[o] = 0; list[atomicAdd(count, 1u)] = j; }
}

// the flags of listed open queries (sorted positions), cleared once the all-points search has taken them from the list
__global__ __launch_bounds__(kBlock) void clear_flags_kernel(const uint32_t* __restrict__ list, uint32_t n_q, const uint32_t* __restrict__ sidx, uint8_t* __restrict__ flag) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t < n_q) flag[sidx[list[t]]] = 0;
}

// ---- grid search over global memory -------------------------------------------------------------------------------------------------
// LIST: the queries are the sorted indices qlist[0 .. nq) (what the box kernel could not finish); otherwise all nf sorted points.
constexpr int kScanBatch = 4;    // candidates requested together by the one-lane-per-query search
template <int K, bool DENSE, bool LIST>
__global__ __launch_bounds__(kBlock, K <= 16 ? 4 : 1) void knn_grid_kernel(const double* __restrict__ sxyz, const uint64_t* __restrict__ skeys, uint32_t nf, uint32_t k,
                                                          GridParams g, CellTable table, const uint32_t* __restrict__ cell_start,
                                                          const uint32_t* __restrict__ qlist, uint32_t nq, RecOut out, int shell_cap,
                                                          uint8_t* __restrict__ unres_flag, uint32_t* __restrict__ unres_count, uint32_t crowd,
                                                          uint32_t* __restrict__ open_lists, uint32_t open_cap, const uint32_t* __restrict__ nq_dev) {
  const uint32_t t0 = blockIdx.x * kBlock + threadIdx.x;
  // nq_dev (stream-ordered replay of a plan): the list's length is only known on the device; nq is then the capacity the grid was sized for
  if (t0 >= (nq_dev ? (*nq_dev < nq ? *nq_dev : nq) : nq)) return;
  const uint32_t j = LIST ? qlist[t0] : t0;
  const double qx = sxyz[3 * (uint64_t)j], qy = sxyz[3 * (uint64_t)j + 1], qz = sxyz[3 * (uint64_t)j + 2];
  double qu, qv, qw;  // the query in the grid's frame: cells and shell margins; distances use (qx, qy, qz)
  grid_frame(g, qx, qy, qz, qu, qv, qw);
  const int cx = (int)cell_coord(qu, g.org[0], g.inv_hx, g.dim[0]), cy = (int)cell_coord(qv, g.org[1], g.inv_h, g.dim[1]),
            cz = (int)cell_coord(qw, g.org[2], g.inv_h, g.dim[2]);
  KBest<K> best;
  best.init();
  // candidates in batches: all coordinate triples of a batch are requested before the first insertion (the loop was waiting on one dependent
  // load per candidate); the insertion itself stays ONE inlined copy (a not-unrolled loop over the batch).  The two ranges of a row are walked
  // as one sequence, so the two single end cells of an interior row share a round trip.
  // crowd > 0 (coarser levels): a range longer than that is not walked by this one lane -- the query is flagged 2 and searched against all
  // points by a workgroup (a coarse cell over a dense part of the cloud holds millions of points)
  bool crowded = false;
  auto scan2 = [&](uint32_t p0, uint32_t p1, uint32_t q0, uint32_t q1) __attribute__((always_inline)) {
    const uint32_t lp = p1 - p0, total = lp + (q1 - q0);
    if (crowd && total > crowd) { crowded = true; return; }
    // (kScanBatch candidates per step, all requested before the first is tested: 1.59 -> 1.48 ms per 0.9 * 10^6 queries of the 10^8-point
    //  cloud against pairs.  Measured and dropped: a per-lane insertion queue in LDS, emptied when some lane's is nearly full -- an insertion
    //  runs for the whole wave when ONE lane inserts, but the walk is bound by its dependent directory and candidate round trips, not by
    //  instructions: 1.48 ms either way.)
    for (uint32_t v = 0; v < total; v += kScanBatch) {
      uint32_t pp[kScanBatch];
      double dd[kScanBatch];
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const uint32_t vu = v + (uint32_t)u < total ? v + (uint32_t)u : v;
        pp[u] = vu < lp ? p0 + vu : q0 + (vu - lp);
      }
      double cx_[kScanBatch], cy_[kScanBatch], cz_[kScanBatch];
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {  // a 16-byte and an 8-byte request per point
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        const d2u xy = *reinterpret_cast<const d2u*>(sxyz + 3 * (uint64_t)pp[u]);
        cx_[u] = xy.x; cy_[u] = xy.y; cz_[u] = sxyz[3 * (uint64_t)pp[u] + 2];
      }
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const double dx = cx_[u] - qx, dy = cy_[u] - qy, dz = cz_[u] - qz;
        dd[u] = v + (uint32_t)u < total ? dx * dx + dy * dy + dz * dz : __builtin_inf();
      }
#pragma nounroll
      for (int u = 0; u < kScanBatch; ++u) best.insert(dd[u], pp[u]);
    }
  };
  const int max_r = (int)max(g.dim[0], max(g.dim[1], g.dim[2]));
  for (int r = 0; r <= max_r; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= (int)g.dim[2]) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= (int)g.dim[1]) continue;
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);
        if constexpr (DENSE) {
          const uint64_t row = ((uint64_t)z * g.dim[1] + (uint64_t)y) * g.dim[0];
          // up to two contiguous ranges of sorted points per row, scanned by ONE inlined copy of the insertion code (three call
          // sites cost 60 VGPRs: 167 -> 3 waves per SIMD)
          uint32_t p0 = 0, p1 = 0, q0 = 0, q1 = 0;
          const int rr = r * (int)g.rx, ri = (r - 1) * (int)g.rx;  // shell r reaches r * rx fine cells along x
          const int x0 = cx - rr < 0 ? 0 : cx - rr, x1 = cx + rr >= (int)g.dim[0] ? (int)g.dim[0] - 1 : cx + rr;
          if (face) {  // the whole row segment [cx - rr, cx + rr] is one contiguous range
            p0 = cell_start[row + (uint32_t)x0]; p1 = cell_start[row + (uint32_t)x1 + 1];
          } else {     // interior rows of the shell: only the two end slabs [cx - rr, cx - ri - 1] and [cx + ri + 1, cx + rr]
            const int a1 = cx - ri - 1, b0 = cx + ri + 1;
            if (a1 >= x0) { p0 = cell_start[row + (uint32_t)x0]; p1 = cell_start[row + (uint32_t)a1 + 1]; }
            if (b0 <= x1) { q0 = cell_start[row + (uint32_t)b0]; q1 = cell_start[row + (uint32_t)x1 + 1]; }
          }
          scan2(p0, p1, q0, q1);
        } else {
          const int xstep = face ? 1 : (2 * r > 0 ? 2 * r : 1);  // interior rows of the shell: only the two end cells
          for (int dx = -r; dx <= r; dx += xstep) {
            const int x = cx + dx;
            if (x < 0 || x >= (int)g.dim[0]) continue;
            const uint64_t key = morton3((uint32_t)x, (uint32_t)y, (uint32_t)z);
            uint32_t p = lookup_cell(table, key);
            if (p == kNoIndex) continue;
            // the cell's points two at a time: keys and coordinates of both are requested before anything is tested or inserted
            const uint32_t p_first = p;
            for (; p < nf; p += 2) {
              if (crowd && p - p_first > crowd) { crowded = true; break; }
              const uint32_t pb = p + 1 < nf ? p + 1 : p;
              const uint64_t ka = skeys[p], kb = skeys[pb];
              const double ax = sxyz[3 * (uint64_t)p], ay = sxyz[3 * (uint64_t)p + 1], az = sxyz[3 * (uint64_t)p + 2];
              const double bx = sxyz[3 * (uint64_t)pb], by = sxyz[3 * (uint64_t)pb + 1], bz = sxyz[3 * (uint64_t)pb + 2];
              if (ka != key) break;
              const bool two = pb != p && kb == key;
              const double adx = ax - qx, ady = ay - qy, adz = az - qz, bdx = bx - qx, bdy = by - qy, bdz = bz - qz;
              const double da = adx * adx + ady * ady + adz * adz;
              const double db = two ? bdx * bdx + bdy * bdy + bdz * bdz : __builtin_inf();
#pragma nounroll
              for (int u = 0; u < 2; ++u) best.insert(u ? db : da, u ? pb : p);
              if (!two) break;
            }
          }
        }
      }
    }
    // (an open query is flagged by its ORIGINAL index -- a coarser level re-sorts the points -- and, for the all-points search of THIS index,
    //  listed by its sorted position: open_lists[0 .. cap) hand-backs, [cap .. 2 cap) crowded ones)
    if (crowded) { unres_flag[out.sidx[j]] = 2; const uint32_t at = atomicAdd(unres_count + 1, 1u); if (at < open_cap) open_lists[open_cap + at] = j; return; }
    if (shell_done(g, qu, qv, qw, cx, cy, cz, r, best.kth(k))) break;
    // shell_cap > 0: a query that is still open after that many shells (an outlier, a point of a region far sparser than the grid was made
    // for: shell r costs (2 r + 1)^2 rows) is handed back -- flagged by its original index -- and searched again on a coarser grid
    if (shell_cap > 0 && r == shell_cap && r < max_r) {
      unres_flag[out.sidx[j]] = 1;
      const uint32_t at = atomicAdd(unres_count, 1u);
      if (at < open_cap) open_lists[at] = j;
      return;
    }
  }
  const uint32_t m = nf < k ? nf : k;
  const uint64_t orig = out.sidx[j];
  if (out.knn || out.knn_u32)
    for (uint32_t t = 0; t < k; ++t) {
      uint32_t v = kNoIndex;
#pragma unroll
      for (int u = 0; u < K; ++u) if ((uint32_t)u == t && t < m) v = best.i[u];
      write_knn(out, orig, k, t, v == kNoIndex ? kNoIndex : out.sidx[v]);
    }
  const Fit f = plane_fit<K, true>(m, [&](uint32_t t, double& x, double& y, double& z) __attribute__((always_inline)) {
    uint32_t p = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) if ((uint32_t)u == t) p = best.i[u];
    x = sxyz[3 * (uint64_t)p]; y = sxyz[3 * (uint64_t)p + 1]; z = sxyz[3 * (uint64_t)p + 2];
  });
  write_record(out, orig, f);
}

// ---- exact search of the queries no grid resolves cheaply (far outliers), against all finite points: bound, filter, select -----------------
// On a grid coarse enough to reach a far outlier's neighbours a cell holds millions of points, and a grid search walks a cell with ONE
// lane (64 outliers around 10^7 points: 67 s on the fourth level).  Instead:
//   1. knn_bound_kernel: the exact k-th distance of every open query within a SUBSAMPLE of the cloud -- an upper bound of the true one;
//   2. knn_filter_kernel: every workgroup of 1024 consecutive sorted points (a few grid cells: a small box) against the open queries whose
//      bound reaches that box (27.9 -> 1.x ms for 1000 queries and 10^8 points); points in registers, queries broadcast from LDS, ~10
//      instructions per pair; a point inside a query's bound is appended to that query's candidate list (about k times the thinning);
//   3. knn_select_kernel: one workgroup per query picks the k nearest of its candidates (every thread the k best of its share, then k
//      rounds of a workgroup-wide minimum over the list heads; ties: lower sorted index), fits the plane and writes the record.  A list
//      that overflowed is replaced by a scan of all points.
template <int K, typename Visit>
__device__ __forceinline__ void block_select_and_fit(const double* __restrict__ sxyz, uint32_t nf, uint32_t k, uint32_t j, const RecOut& out, double* kth_out, Visit&& visit) {
  __shared__ double wd[kBlock / 64];
  __shared__ uint32_t wi[kBlock / 64];
  __shared__ uint32_t res[64];
  __shared__ uint32_t win;
  KBest<K> best;
  best.init();
  visit([&](double d, uint32_t p) __attribute__((always_inline)) { if (d < best.kth(k)) best.insert(d, p); });
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  double last = __builtin_inf();
  for (uint32_t t = 0; t < k; ++t) {
    double d = best.d[0];
    uint32_t i = best.i[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = shfl_xor_any(d, off);
      const uint32_t oi = (uint32_t)__shfl_xor((int)i, off, 64);
      if (od < d || (od == d && oi < i)) { d = od; i = oi; }
    }
    if (lane == 0) { wd[wave] = d; wi[wave] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double bd = wd[0];
      uint32_t bi = wi[0];
      for (int w = 1; w < kBlock / 64; ++w) if (wd[w] < bd || (wd[w] == bd && wi[w] < bi)) { bd = wd[w]; bi = wi[w]; }
      res[t] = bi;
      win = bi;
      wd[0] = bd;
    }
    __syncthreads();
    last = wd[0];
    if (best.i[0] == win && win != kNoIndex) {  // the winner drops its head
#pragma unroll
      for (int u = 0; u + 1 < K; ++u) { best.d[u] = best.d[u + 1]; best.i[u] = best.i[u + 1]; }
      best.d[K - 1] = __builtin_inf(); best.i[K - 1] = kNoIndex;
    }
    __syncthreads();
  }
  if (kth_out) { if (threadIdx.x == 0) *kth_out = last; return; }  // (+inf when fewer than k points were visited)
  if (wave != 0) return;
  // the plane fit as a wave-level operation (plane_fit_wave): lane t fetches neighbour t -- ONE parallel gather instead of 2 m dependent
  // loads by a single lane --, the sums run across the lanes in point order (v_readlane): the reference's order of operations
  const uint32_t m = nf < k ? nf : k;
  const uint64_t orig = out.sidx[j];
  const bool have = lane < m;
  const uint32_t pn = have ? res[lane] : j;
  const double nx = sxyz[3 * (uint64_t)pn], ny = sxyz[3 * (uint64_t)pn + 1], nz = sxyz[3 * (uint64_t)pn + 2];
  if ((out.knn || out.knn_u32) && lane < k) write_knn(out, orig, k, lane, have ? out.sidx[pn] : kNoIndex);
  const Fit f = plane_fit_wave(m, nx, ny, nz);
  if (lane == 0) write_record(out, orig, f);
}

template <int K>
__global__ __launch_bounds__(kBlock) void knn_bound_kernel(const double* __restrict__ sxyz, const uint32_t* __restrict__ qlist, uint32_t nq, const double* __restrict__ sub,
                                                           uint32_t n_sub, uint32_t k, double* __restrict__ bound) {
  if (blockIdx.x >= nq) return;
  const uint32_t j = qlist[blockIdx.x];
  const double qx = sxyz[3 * (uint64_t)j], qy = sxyz[3 * (uint64_t)j + 1], qz = sxyz[3 * (uint64_t)j + 2];
  // eight points per thread and step, all requested before the first is tested (one workgroup walks the whole subsample)
  constexpr int U = 8;
  auto walk = [&](uint32_t n_walk, auto&& each) __attribute__((always_inline)) {
    for (uint32_t p0 = threadIdx.x; p0 < n_walk; p0 += kBlock * U) {
      double x[U], y[U], z[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = p0 + (uint32_t)u * kBlock < n_walk ? p0 + (uint32_t)u * kBlock : p0;
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        const d2u xy = *reinterpret_cast<const d2u*>(sub + 3 * (uint64_t)p);
        x[u] = xy.x; y[u] = xy.y; z[u] = sub[3 * (uint64_t)p + 2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double dx = x[u] - qx, dy = y[u] - qy, dz = z[u] - qz;
        const double d = dx * dx + dy * dy + dz * dz;  // (NaN for a non-finite point of the subsample: fails every comparison below)
        if (p0 + (uint32_t)u * kBlock < n_walk) each(d, p0 + (uint32_t)u * kBlock);
      }
    }
  };
  // Pass 1: every thread's nearest point.  The k-th smallest of these 256 minima (k <= 64 distinct points) is at least the k-th distance
  // of the whole subsample, so pass 2 inserts only what lies within it -- a few dozen points per workgroup.  Without it every candidate of
  // every thread walked the sorted list (an insertion runs for the whole wave when ONE lane inserts: 4096 x 110 instructions per wave, 3.2 ms
  // for 1000 queries against 10^6 points).
  __shared__ double mins[kBlock];
  __shared__ double limit_s;
  double mine = __builtin_inf();
  // (pass 1 over the first eighth of the subsample -- itself a uniform thinning, in input order: its k-th distance bounds the whole subsample's,
  //  a few hundred points per workgroup pass instead of a few dozen, and the 1000 workgroups read 28 GB from L2 instead of 50)
  const uint32_t n_first = n_sub / 8 > 64u * kBlock ? n_sub / 8 : (n_sub < 64u * kBlock ? n_sub : 64u * kBlock);
  walk(n_first, [&](double d, uint32_t) __attribute__((always_inline)) { mine = d < mine ? d : mine; });
  mins[threadIdx.x] = mine;
  if (threadIdx.x == 0) limit_s = __builtin_inf();
  __syncthreads();
  {
    uint32_t below = 0;  // minima ordered before this thread's (ties: lower thread first)
    for (uint32_t t = 0; t < kBlock; ++t) { const double o = mins[t]; below += (o < mine || (o == mine && t < threadIdx.x)) ? 1u : 0u; }
    if (below == k - 1 && k <= kBlock) limit_s = mine;  // (fewer than k finite minima: +inf, everything is inserted)
  }
  __syncthreads();
  const double limit = limit_s;
  RecOut none{};
  block_select_and_fit<K>(sxyz, n_sub, k, j, none, bound + blockIdx.x, [&](auto&& take) __attribute__((always_inline)) {
    walk(n_sub, [&](double d, uint32_t p) __attribute__((always_inline)) { if (d <= limit) take(d, p); });
  });
}

constexpr uint32_t kFilterPts = 4;     // points per thread of the filter (per 10^8 points and 1000 queries: 2 -> 1.57 ms, 4 -> 1.56 ms, 8 -> 2.40 ms)
constexpr uint32_t kFilterQPer = 4, kFilterChunk = kFilterQPer * 256;  // queries tested per lane and per step
// qpack[q] = {x, y, z, bound} of open query q (knn_pack_queries_kernel): every workgroup of the filter stages every query, and fetching them
// through the query list (index -> point -> coordinates, a dependent round trip per chunk and workgroup) was most of its 1.7 ms
__global__ __launch_bounds__(kBlock) void knn_pack_queries_kernel(const double* __restrict__ sxyz, const uint32_t* __restrict__ qlist, uint32_t nq,
                                                                  const double* __restrict__ bound, double* __restrict__ qpack) {
  const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= nq) return;
  const uint32_t j = qlist[q];
  qpack[4 * (uint64_t)q] = sxyz[3 * (uint64_t)j]; qpack[4 * (uint64_t)q + 1] = sxyz[3 * (uint64_t)j + 1]; qpack[4 * (uint64_t)q + 2] = sxyz[3 * (uint64_t)j + 2];
  qpack[4 * (uint64_t)q + 3] = bound[q];
}
__global__ __launch_bounds__(kBlock) void knn_filter_kernel(const double* __restrict__ sxyz, uint32_t nf, const double* __restrict__ qpack, uint32_t nq, uint32_t cap,
                                                            uint32_t* __restrict__ cand_count, uint32_t* __restrict__ cand) {
  __shared__ double box[6];                // bounding box of the workgroup's points: consecutive SORTED points, a few grid cells
  __shared__ double scratch[(kBlock / 64) * 6];
  __shared__ uint16_t act[kFilterChunk];   // the step's queries whose ball reaches the box
  __shared__ uint32_t n_act;
  const uint32_t p0 = (blockIdx.x * kBlock + threadIdx.x) * kFilterPts;
  double px[kFilterPts], py[kFilterPts], pz[kFilterPts];
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  if (p0 + kFilterPts <= nf) {  // the thread's points are 24 * kFilterPts contiguous bytes (16-byte aligned: kFilterPts is even): 16-byte requests
    static_assert(kFilterPts % 2 == 0, "whole 16-byte pieces");
    typedef double d2a __attribute__((ext_vector_type(2)));
    double w[3 * kFilterPts];
#pragma unroll
    for (uint32_t i = 0; i < 3 * kFilterPts / 2; ++i) { const d2a v = *reinterpret_cast<const d2a*>(sxyz + 3 * (uint64_t)p0 + 2 * i); w[2 * i] = v.x; w[2 * i + 1] = v.y; }
#pragma unroll
    for (uint32_t u = 0; u < kFilterPts; ++u) { px[u] = w[3 * u]; py[u] = w[3 * u + 1]; pz[u] = w[3 * u + 2]; }
  } else {
#pragma unroll
    for (uint32_t u = 0; u < kFilterPts; ++u) {
      const uint32_t p = p0 + u < nf ? p0 + u : nf - 1;
      px[u] = sxyz[3 * (uint64_t)p]; py[u] = sxyz[3 * (uint64_t)p + 1]; pz[u] = sxyz[3 * (uint64_t)p + 2];
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < kFilterPts; ++u) {
    mn[0] = __builtin_fmin(mn[0], px[u]); mx[0] = __builtin_fmax(mx[0], px[u]);
    mn[1] = __builtin_fmin(mn[1], py[u]); mx[1] = __builtin_fmax(mx[1], py[u]);
    mn[2] = __builtin_fmin(mn[2], pz[u]); mx[2] = __builtin_fmax(mx[2], pz[u]);
  }
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) { box[0] = mn[0]; box[1] = mn[1]; box[2] = mn[2]; box[3] = mx[0]; box[4] = mx[1]; box[5] = mx[2]; }
  // kFilterChunk queries per step, kFilterQPer per lane, all requested before the first is tested; only the numbers of the queries whose ball
  // reaches the box go to LDS (a stray far from the cloud reaches a handful of the 10^5 boxes), their data is read again where it is used
  for (uint32_t q0 = 0; q0 < nq; q0 += kFilterChunk) {
    const uint32_t cnt = min(kFilterChunk, nq - q0);
    __syncthreads();
    if (threadIdx.x == 0) n_act = 0;
    __syncthreads();
    typedef double d2a __attribute__((ext_vector_type(2)));
    d2a xy[kFilterQPer], zb[kFilterQPer];
#pragma unroll
    for (uint32_t u = 0; u < kFilterQPer; ++u) {
      const uint32_t ql = u * kBlock + threadIdx.x < cnt ? u * kBlock + threadIdx.x : 0u;
      xy[u] = *reinterpret_cast<const d2a*>(qpack + 4 * (uint64_t)(q0 + ql));
      zb[u] = *reinterpret_cast<const d2a*>(qpack + 4 * (uint64_t)(q0 + ql) + 2);
    }
#pragma unroll
    for (uint32_t u = 0; u < kFilterQPer; ++u) {
      const double qx = xy[u].x, qy = xy[u].y, qz = zb[u].x, b = zb[u].y;
      // squared distance from the query to the box: only a query whose bound reaches the box can find a candidate here
      const double ex = __builtin_fmax(0.0, __builtin_fmax(box[0] - qx, qx - box[3])), ey = __builtin_fmax(0.0, __builtin_fmax(box[1] - qy, qy - box[4])),
                   ez = __builtin_fmax(0.0, __builtin_fmax(box[2] - qz, qz - box[5]));
      if (u * kBlock + threadIdx.x < cnt && ex * ex + ey * ey + ez * ez <= b) act[atomicAdd(&n_act, 1u)] = (uint16_t)(u * kBlock + threadIdx.x);
    }
    __syncthreads();
    const uint32_t na = n_act;
    for (uint32_t a = 0; a < na; ++a) {  // every lane reads the same query
      const uint32_t q = act[a];
      const double qx = qpack[4 * (uint64_t)(q0 + q)], qy = qpack[4 * (uint64_t)(q0 + q) + 1], qz = qpack[4 * (uint64_t)(q0 + q) + 2], b = qpack[4 * (uint64_t)(q0 + q) + 3];
#pragma unroll
      for (uint32_t u = 0; u < kFilterPts; ++u) {
        const double dx = px[u] - qx, dy = py[u] - qy, dz = pz[u] - qz;
        if (dx * dx + dy * dy + dz * dz <= b && p0 + u < nf) {
          const uint32_t pos = atomicAdd(&cand_count[q0 + q], 1u);
          if (pos < cap) cand[(uint64_t)(q0 + q) * cap + pos] = p0 + u;
        }
      }
    }
  }
}

template <int K>
__global__ __launch_bounds__(kBlock) void knn_select_kernel(const double* __restrict__ sxyz, uint32_t nf, uint32_t k, const uint32_t* __restrict__ qlist, uint32_t nq,
                                                            const uint32_t* __restrict__ cand_count, const uint32_t* __restrict__ cand, uint32_t cap, RecOut out) {
  if (blockIdx.x >= nq) return;
  const uint32_t j = qlist[blockIdx.x];
  const double qx = sxyz[3 * (uint64_t)j], qy = sxyz[3 * (uint64_t)j + 1], qz = sxyz[3 * (uint64_t)j + 2];
  const uint32_t cnt = cand_count ? cand_count[blockIdx.x] : 0xFFFFFFFFu;
  const bool listed = cnt <= cap && cnt >= (nf < k ? nf : k);  // (overflow, or no filter at all: every point is a candidate)
  const uint32_t total = listed ? cnt : nf;
  block_select_and_fit<K>(sxyz, nf, k, j, out, (double*)nullptr, [&](auto&& take) __attribute__((always_inline)) {
    for (uint32_t c = threadIdx.x; c < total; c += kBlock) {
      const uint32_t p = listed ? cand[(uint64_t)blockIdx.x * cap + c] : c;
      const double dx = sxyz[3 * (uint64_t)p] - qx, dy = sxyz[3 * (uint64_t)p + 1] - qy, dz = sxyz[3 * (uint64_t)p + 2] - qz;
      take(dx * dx + dy * dy + dz * dz, p);
    }
  });
}

// non-finite query points (sorted positions [nf, n)): neighbourhood = itself + the first k-1 finite points
__global__ __launch_bounds__(kBlock) void knn_nonfinite_kernel(const double* __restrict__ xyz, const double* __restrict__ sxyz, uint32_t nf, uint32_t n,
                                                               uint32_t k, RecOut out) {
  const uint32_t j = nf + blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  const uint64_t orig = out.sidx[j];
  const uint32_t m = (nf + 1 < k) ? nf + 1 : k;
  if (out.knn || out.knn_u32)
    for (uint32_t t = 0; t < k; ++t) write_knn(out, orig, k, t, t == 0 ? (uint32_t)orig : (t < m ? out.sidx[t - 1] : kNoIndex));
  const Fit f = plane_fit(m, [&](uint32_t t, double& x, double& y, double& z) {
    if (t == 0) { x = xyz[3 * orig]; y = xyz[3 * orig + 1]; z = xyz[3 * orig + 2]; }
    else { x = sxyz[3 * (uint64_t)(t - 1)]; y = sxyz[3 * (uint64_t)(t - 1) + 1]; z = sxyz[3 * (uint64_t)(t - 1) + 2]; }
  });
  write_record(out, orig, f);
}

}  // namespace

namespace pstk {

namespace {
struct XyzRef { const double* p; template <typename T> const T* as() const { return (const T*)p; } };

// Scratch of one compute_normals call (~90 bytes per point in a dozen arrays).  The arrays of a call have the same sizes as those of the
// previous call on the same cloud, so they are kept in a per-thread cache of device blocks keyed by size and handed out again: the driver is
// not asked for memory in the steady state.  (Measured with hipMallocAsync / hipFreeAsync per array instead: every ~14th call of a 10^8-point
// cloud stalled for 0.7-0.9 s inside one of the allocations -- three per stalled call -- with the GPU idle; rocprofv3 showed no long kernel.)
// Blocks are only reused after the call that used them has synchronised its stream (run_normals ends with a read-back), blocks that went
// unused while the cache holds more than twice what the last call needed are returned to the driver.
struct ScratchCache {
  struct Block { void* p; size_t bytes; bool busy; unsigned idle_calls; };
  std::vector<Block> blocks;
  void* take(size_t bytes) {
    bytes = bytes ? (bytes + 255) & ~(size_t)255 : 256;
    Block* best = nullptr;
    for (Block& b : blocks)
      if (!b.busy && b.bytes >= bytes && b.bytes <= bytes + bytes / 8 + 4096 && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; best->idle_calls = 0; return best->p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      // out of memory: the blocks this cache holds for nobody go back first, then the pool's (device_malloc_retry), then the request is repeated;
      // a second failure leaves the HIP error for the caller's report
      (void)hipGetLastError();
      for (size_t i = 0; i < blocks.size();) {
        if (!blocks[i].busy) { (void)hipFree(blocks[i].p); blocks[i] = blocks.back(); blocks.pop_back(); }
        else ++i;
      }
      if (device_malloc_retry(&p, bytes) != hipSuccess) return nullptr;
    }
    blocks.push_back(Block{p, bytes, true, 0});
    return p;
  }
  void end_call() {
    size_t used = 0, held = 0;
    for (Block& b : blocks) { held += b.bytes; if (b.busy) used += b.bytes; else b.idle_calls += 1; b.busy = false; }
    if (held > 2 * used) {
      for (size_t i = 0; i < blocks.size();) {
        if (blocks[i].idle_calls >= 2) { held -= blocks[i].bytes; (void)hipFree(blocks[i].p); blocks[i] = blocks.back(); blocks.pop_back(); }
        else ++i;
      }
    }
    // CAP: a thread never keeps more than PST_SCRATCH_MAX_BYTES (default 16 GiB: a 10^8-point surface with its 7.5 GB directory keeps 13.6) between calls.
    // Beyond it the largest blocks go back to the driver first -- a larger cloud then pays the allocations on every call.
    const size_t cap = (size_t)knn_tuning().scratch_max;
    while (held > cap && !blocks.empty()) {
      size_t big = 0;
      for (size_t i = 1; i < blocks.size(); ++i) if (blocks[i].bytes > blocks[big].bytes) big = i;
      held -= blocks[big].bytes;
      (void)hipFree(blocks[big].p);
      blocks[big] = blocks.back();
      blocks.pop_back();
    }
  }
  size_t held_bytes() const { size_t h = 0; for (const Block& b : blocks) h += b.bytes; return h; }
  void release_all() {  // (no call of this thread is in flight: every call synchronises its stream before it returns)
    for (Block& b : blocks) (void)hipFree(b.p);  // (hipFree of another device's pointer is valid from any current device)
    blocks.clear();
  }
  ScratchCache() = default;
  ScratchCache(ScratchCache&& o) noexcept : blocks(std::move(o.blocks)) { o.blocks.clear(); }
  ScratchCache(const ScratchCache&) = delete;
  ~ScratchCache() { release_all(); }
};
// one cache per (thread, device): after pst_set_device(d) a thread must get blocks that live on GPU d
ScratchCache& scratch_cache() {
  static thread_local std::vector<ScratchCache> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if ((size_t)dev >= per_device.size()) per_device.resize((size_t)dev + 1);
  return per_device[(size_t)dev];
}
struct CacheBuf {  // same surface as DevBuf
  void* p = nullptr;
  hipError_t alloc(size_t bytes, hipStream_t) { p = scratch_cache().take(bytes); return p ? hipSuccess : hipErrorOutOfMemory; }
  template <typename T> T* as() { return (T*)p; }
};
// A second stream per calling thread and device: the permutation of the points (reorder_kernel: bound by random read requests) runs there while
// the caller's stream builds the cell directory from the sorted keys (bound by streaming writes: 7.5 GB for a LiDAR sheet of 10^8 points) --
// the two need nothing from each other.  fork / join: events that order the side stream behind, and the caller's stream after, that work.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
};
SideStream& side_stream() {
  thread_local std::vector<SideStream> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if ((size_t)dev >= per_device.size()) per_device.resize((size_t)dev + 1);
  SideStream& ss = per_device[(size_t)dev];
  if (!ss.s) {
    ss.ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
    if (!ss.ok) (void)hipGetLastError();
  }
  return ss;
}
struct CallGuard {
  hipStream_t stream;
  ~CallGuard() { (void)hipStreamSynchronize(stream); scratch_cache().end_call(); }  // blocks go back only when nothing in flight uses them
};
}  // namespace

// Frees the device blocks the calling thread's kNN calls keep between calls (about 55 bytes per point of the largest recent cloud, capped at PST_SCRATCH_MAX_BYTES, default 16 GiB).
void release_normals_scratch() { scratch_cache().release_all(); }  // (the current device's cache)

// Returns 0 on success, -1 on a HIP failure (hipGetLastError has it), -2 for inputs beyond the 32-bit point indices of the spatial index,
// or the number of degenerate neighbourhoods (> 0).
long long run_normals(const uint8_t* pos_base, uint64_t pos_stride, uint64_t n, uint32_t k, double* out_normals_dev, double* out_curv_dev,
                      long long* out_knn_dev, uint32_t* out_knn_u32_dev, uint64_t normal_attr, uint64_t normal_stride, uint64_t curv_attr,
                      uint64_t curv_stride, hipStream_t stream, KnnPlanRecord* record) {
#define NCK(x) do { if ((x) != hipSuccess) return -1; } while (0)
  if (record) { *record = KnnPlanRecord{}; record->why_not = "the call did not take the LDS box search"; }
  bool rec_open = false;         // some query was handed back by a capped search (coarser levels / all-points search ran)
  bool rec_tiled = false, rec_list = false;
  uint32_t rec_n_list = 0, rec_n_fb = 0;
  TileShape rec_shape{};
  GridParams rec_g{};
  uint64_t rec_cells = 0, rec_nf = 0;
  if (n >= 0xFFFFFFF0ull) return -2;  // sorted indices and directory entries are uint32_t
  const unsigned cus = (unsigned)device_cus();
  const unsigned sgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)cus * 8));
  // a packed, 8-byte aligned Vec3f64 array (a HashMapBuffer column, an XYZ-only VectorBuffer) is searched in place: no 24 n-byte copy
  const bool packed_source = pos_stride == 24 && ((uintptr_t)pos_base & 7u) == 0;
  CallGuard scratch_guard{stream};  // every exit of this function hands the scratch blocks back to the cache (all of them synchronise the stream or fail)
  CacheBuf xyz_own, partials, counters;
  if (!packed_source) NCK(xyz_own.alloc(n * 24, stream));
  XyzRef xyz{packed_source ? (const double*)pos_base : (const double*)xyz_own.p};
  NCK(partials.alloc((size_t)sgrid * 48, stream));
  NCK(counters.alloc(128, stream));
  NCK(hipMemsetAsync(counters.p, 0, 128, stream));
  hipLaunchKernelGGL(gather_positions_kernel, dim3(sgrid), dim3(kBlock), 0, stream, pos_base, pos_stride, n, packed_source ? (double*)nullptr : xyz_own.as<double>(),
                     partials.as<double>());
  std::vector<double> hp((size_t)sgrid * 6);
  NCK(hipMemcpyAsync(hp.data(), partials.p, hp.size() * 8, hipMemcpyDeviceToHost, stream));
  NCK(hipStreamSynchronize(stream));
  double mn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308}, mx[3] = {-mn[0], -mn[0], -mn[0]};
  for (unsigned b = 0; b < sgrid; ++b)
    for (int c = 0; c < 3; ++c) { mn[c] = std::fmin(mn[c], hp[b * 6 + c]); mx[c] = std::fmax(mx[c], hp[b * 6 + 3 + c]); }

  NormalsOut out{};
  out.normals_f64 = out_normals_dev; out.curvature_f64 = out_curv_dev; out.knn = out_knn_dev; out.knn_u32 = out_knn_u32_dev;
  out.normal_attr = normal_attr; out.normal_stride = normal_stride; out.curv_attr = curv_attr; out.curv_stride = curv_stride;
  unsigned long long* n_finite = (unsigned long long*)counters.p;
  unsigned long long* n_cells = n_finite + 1;
  uint32_t* fb_count = (uint32_t*)((uint8_t*)counters.p + 16);
  out.error_count = (int*)((uint8_t*)counters.p + 32);

  const bool any_finite = mn[0] <= mx[0];
  const bool brute = n <= 2048 || !any_finite;
#define KNN_DISPATCH_T(GRID, K1, K2, K3, K4, ...)                                                       \
  do {                                                                                                  \
    if (k <= 8) hipLaunchKernelGGL((K1), dim3(GRID), dim3(kBlock), 0, stream, __VA_ARGS__);            \
    else if (k <= 16) hipLaunchKernelGGL((K2), dim3(GRID), dim3(kBlock), 0, stream, __VA_ARGS__);      \
    else if (k <= 32) hipLaunchKernelGGL((K3), dim3(GRID), dim3(kBlock), 0, stream, __VA_ARGS__);      \
    else hipLaunchKernelGGL((K4), dim3(GRID), dim3(kBlock), 0, stream, __VA_ARGS__);                   \
  } while (0)
#define KNN_DISPATCH(KERNEL, GRID, ...) KNN_DISPATCH_T(GRID, KERNEL<8>, KERNEL<16>, KERNEL<32>, KERNEL<64>, __VA_ARGS__)
#define KNN_DISPATCH_GRID(DENSE, LIST, GRID, ...)                                                                                          \
  KNN_DISPATCH_T(GRID, (knn_grid_kernel<8, DENSE, LIST>), (knn_grid_kernel<16, DENSE, LIST>), (knn_grid_kernel<32, DENSE, LIST>),          \
                 (knn_grid_kernel<64, DENSE, LIST>), __VA_ARGS__)
  if (brute) {
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    KNN_DISPATCH(knn_bruteforce_kernel, grid, xyz.as<double>(), (uint32_t)n, k, out);
  } else {
    // cell edge: a sphere of radius h should hold about k points  =>  (4/3 pi) h^3 * density ~ k
    GridParams frame{};  // the frame the grids live in: the cloud's own axes, or (below) its principal axes
    const double full_mn[3] = {mn[0], mn[1], mn[2]}, full_mx[3] = {mx[0], mx[1], mx[2]};  // the bounding box (mn / mx may become a trimmed or rotated one)
    // the box the grids are laid over: the bounding box, or (below) a trimmed one when a few far points stretch it
    const KnnTuning& tune = knn_tuning();
    BoxStats bs{};
    double (&ext)[3] = bs.ext;
    double& maxext = bs.maxext;
    auto set_box = [&]() { bs = BoxStats::of(mn, mx); };
    set_box();
    // Points per cell.  With the hash table every cell costs a probe, so few fat cells win: ~k/3 points per cell (the first
    // shell of 27 cells almost always suffices).  With the dense directory a whole row of cells is one range, and small cells win
    // because the searched cube approximates the k-sphere better: ~k/12 points per cell, two shells
    // (global-memory search at k = 16, 10^8 points: 5.33 -> 126 ms, 2.2 -> 116, 1.3 -> 100, 0.8 -> 107, 0.4 -> 145).
    auto grid_for = [&](double h, uint32_t rx, GridParams& g) -> uint64_t {
      if (tune.cell > 0) h = tune.cell;
      const double min_h = maxext * (double)rx / 2000000.0;  // <= 2^21 cells per axis
      if (!(h > min_h)) h = min_h;
      g.h = h; g.inv_h = 1.0 / h; g.rx = rx; g.hx = h / (double)rx; g.inv_hx = (double)rx / h;
      g.rotated = frame.rotated;
      for (int c = 0; c < 9; ++c) g.rot[c] = frame.rot[c];
      for (int c = 0; c < 3; ++c) g.rot_c[c] = frame.rot_c[c];
      for (int c = 0; c < 3; ++c) {
        g.org[c] = mn[c];
        double d = std::floor(ext[c] * (c == 0 ? g.inv_hx : g.inv_h)) + 1.0;
        if (d > 2097151.0) d = 2097151.0;
        g.dim[c] = (uint32_t)d;
      }
      return (uint64_t)g.dim[0] * g.dim[1] * g.dim[2];
    };
    auto edge_for = [&](double per_cell) { return bs.edge_for(per_cell, n); };
    auto is_dense = [&](uint64_t cells) { return dense_directory_ok(cells, n); };
    const double per_cell_env = tune.per_cell;
    const bool debug = tune.debug;
    const bool trace = tune.trace;  // host wall time of every phase (each mark synchronises the stream)
    auto t_prev = std::chrono::steady_clock::now();
    std::string trace_line;
    auto mark = [&](const char* what) {
      if (!trace) return;
      (void)hipStreamSynchronize(stream);
      const auto now = std::chrono::steady_clock::now();
      char buf[64];
      snprintf(buf, sizeof buf, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      trace_line += buf;
      t_prev = now;
    };

    CacheBuf keys, keys2, idx, idx2, sorted_xyz, tmp, rec, directory, dir_blocks, tkeys, tstarts, fb_list;
    NCK(keys.alloc(n * 8, stream)); NCK(keys2.alloc(n * 8, stream)); NCK(idx.alloc(n * 4, stream)); NCK(idx2.alloc(n * 4, stream));
    NCK(sorted_xyz.alloc(n * 24, stream));
    GridParams g{};
    uint64_t cells = 0, nf = 0;
    CellTable table{nullptr, nullptr, 0};
    // The spatial index for a given grid: keys, radix sort, reorder, and the dense directory or the hash table.  Returns false on a HIP failure.
    // (src, cnt): the points to index -- all of them, or the subsample the density estimate below works on
    auto build_index = [&](double h, uint32_t rx, bool dense, const double* src = nullptr, uint64_t cnt = 0) -> bool {
#define BCK(x) do { if ((x) != hipSuccess) return false; } while (0)
      struct Join {  // every way out of this function: the caller's stream waits for what was sent to the side stream
        hipStream_t stream;
        SideStream* ss = nullptr;
        ~Join() { if (ss && hipStreamWaitEvent(stream, ss->join, 0) != hipSuccess) { (void)hipStreamSynchronize(ss->s); } }
      } join{stream};
      if (!src) { src = xyz.as<double>(); cnt = n; }
      cells = grid_for(h, rx, g);
      g.dense = dense ? 1u : 0u;
      unsigned key_bits = 64;
      if (dense) { key_bits = 1; while (key_bits < 32 && (1ull << key_bits) <= cells) ++key_bits; }  // keys 0 .. cells (< 2^32)
      BCK(hipMemsetAsync(counters.p, 0, 16, stream));
      size_t tmp_bytes = 0;
      pstk::RadixFirstPass walk{nullptr, (uint32_t)((cnt + 8191) / 8192), 0, 8192};
      if (dense) {
        // the library's own sort numbers the points itself and takes the histogram of its first pass from the key kernel
        BCK(sort_pairs_u32(nullptr, tmp_bytes, keys.as<uint32_t>(), keys2.as<uint32_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), cnt, key_bits, stream));
        BCK(tmp.alloc(tmp_bytes, stream));
        const pstk::RadixFirstPass first = pstk::sort_first_pass(tmp.p, cnt, key_bits);
        if (first.counts) walk = first;
        hipLaunchKernelGGL(keys_kernel<uint32_t>, dim3(std::max(1u, std::min(walk.tiles, cus * 16u))), dim3(kBlock), 0, stream, src, cnt, g, keys.as<uint32_t>(), (uint32_t*)nullptr, n_finite, walk);
        BCK(sort_pairs_u32(tmp.p, tmp_bytes, keys.as<uint32_t>(), keys2.as<uint32_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), cnt, key_bits, stream, true, &first));
      } else {
        hipLaunchKernelGGL(keys_kernel<uint64_t>, dim3(std::max(1u, std::min(walk.tiles, cus * 16u))), dim3(kBlock), 0, stream, src, cnt, g, keys.as<uint64_t>(), idx.as<uint32_t>(), n_finite, walk);
        BCK(sort_pairs_u64(nullptr, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), cnt, key_bits, stream));
        BCK(tmp.alloc(tmp_bytes, stream));
        BCK(sort_pairs_u64(tmp.p, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), cnt, key_bits, stream));
      }
      {
        const int unroll = tune.reorder_unroll;  // (same box, whole kNN call at 10^8 points: 32.6 / 32.3 / 32.4 ms for 1 / 2 / 4)
        // one step per workgroup (a grid of cnt / (256 * unroll) workgroups): a plain random gather of 10^8 points measured 2.41 ms that way and
        // 2.58 ms with 2048 workgroups looping (tools/exp_locality.hip)
        const int u = unroll >= 4 ? 4 : (unroll == 2 ? 2 : 1);
        const unsigned rgrid = (unsigned)std::max<uint64_t>(1, (cnt + (uint64_t)kBlock * u - 1) / ((uint64_t)kBlock * u));
        // on the side stream, behind the sort; the caller's stream goes on to the directory and waits for the points when this function returns
        hipStream_t rs = stream;
        if (tune.side_stream) {
          SideStream& ss = side_stream();
          if (ss.ok && hipEventRecord(ss.fork, stream) == hipSuccess && hipStreamWaitEvent(ss.s, ss.fork, 0) == hipSuccess) { rs = ss.s; join.ss = &ss; }
        }
        if (u == 4) hipLaunchKernelGGL(reorder_kernel<4>, dim3(rgrid), dim3(kBlock), 0, rs, src, idx2.as<uint32_t>(), cnt, sorted_xyz.as<double>());
        else if (u == 2) hipLaunchKernelGGL(reorder_kernel<2>, dim3(rgrid), dim3(kBlock), 0, rs, src, idx2.as<uint32_t>(), cnt, sorted_xyz.as<double>());
        else hipLaunchKernelGGL(reorder_kernel<1>, dim3(rgrid), dim3(kBlock), 0, rs, src, idx2.as<uint32_t>(), cnt, sorted_xyz.as<double>());
        if (join.ss && hipEventRecord(join.ss->join, rs) != hipSuccess) { (void)hipStreamSynchronize(rs); join.ss = nullptr; return false; }
      }
      unsigned long long h_counts[2] = {0, 0};
      BCK(hipMemcpyAsync(&h_counts[0], n_finite, 8, hipMemcpyDeviceToHost, stream));
      BCK(hipStreamSynchronize(stream));
      nf = h_counts[0];
      if (!nf) return true;
      if (dense) {
        BCK(directory.alloc((cells + 2) * 4, stream));
        if (cells > 3 * nf) {
          const uint64_t n_dblocks = (cells + 1 + kDirBlock - 1) / kDirBlock;
          BCK(dir_blocks.alloc(n_dblocks * 4, stream));
          BCK(hipMemsetAsync(dir_blocks.p, 0xFF, n_dblocks * 4, stream));
          hipLaunchKernelGGL(dir_block_heads_kernel, dim3(sgrid), dim3(kBlock), 0, stream, keys2.as<uint32_t>(), nf, cells, dir_blocks.as<uint32_t>());
          size_t sb = 0;
          BCK(suffix_min_u32(nullptr, sb, dir_blocks.as<uint32_t>(), n_dblocks, stream));
          BCK(tmp.alloc(sb, stream));
          BCK(suffix_min_u32(tmp.p, sb, dir_blocks.as<uint32_t>(), n_dblocks, stream));
          hipLaunchKernelGGL(dir_fill_kernel, dim3((unsigned)((n_dblocks + kDirPerGroup - 1) / kDirPerGroup)), dim3(kBlock), 0, stream, keys2.as<uint32_t>(), nf, cells, (const uint32_t*)dir_blocks.as<uint32_t>(),
                             n_dblocks, directory.as<uint32_t>());
        } else {
          hipLaunchKernelGGL(build_directory_kernel, dim3(sgrid), dim3(kBlock), 0, stream, keys2.as<uint32_t>(), nf, cells, directory.as<uint32_t>());
        }
      } else {
        hipLaunchKernelGGL(count_cells_kernel, dim3(sgrid), dim3(kBlock), 0, stream, keys2.as<uint64_t>(), nf, n_cells);
        BCK(hipMemcpyAsync(&h_counts[1], n_cells, 8, hipMemcpyDeviceToHost, stream));
        BCK(hipStreamSynchronize(stream));
        uint64_t cap = 64;
        while (cap < 2 * h_counts[1]) cap <<= 1;
        BCK(tkeys.alloc(cap * 8, stream)); BCK(tstarts.alloc(cap * 4, stream));
        BCK(hipMemsetAsync(tkeys.p, 0xFF, cap * 8, stream));
        table = CellTable{tkeys.as<uint64_t>(), tstarts.as<uint32_t>(), (uint32_t)(cap - 1)};
        hipLaunchKernelGGL(build_table_kernel, dim3(sgrid), dim3(kBlock), 0, stream, keys2.as<uint64_t>(), nf, table);
      }
      return true;
#undef BCK
    };

    // What is decided here, in this order: is the cloud what its bounding box says (a quick scale estimate against the box's volume)?  The
    // box the grids are laid over (the bounding box; a trimmed one when far points stretch it; the box along the principal axes when the
    // cloud is thin in a direction that is no coordinate axis).  The cell edge h = the radius of the ball that holds M = 1.75 k points --
    // from the box's volume for clouds that fill it, else measured (normals_scale.hip), and checked by a probe of the built index.  Then
    // 1. the LDS box search (normals_tile.hip) over a dense directory with x cells rx times finer than h, when the directory fits its
    //    budget; what it cannot finish goes to 2 as a list;
    // 2. otherwise the global-memory search over the dense directory with ~k/12 points per cubic cell, when the grid is not much larger
    //    than the cloud -- 3. else over Morton keys + a hash table with ~k/3 points per cell;
    // 4. queries 2 / 3 hand back after kShellCap shells: coarser grids over the full bounding box, then an exact search against all points.
    TileShape shape;
    BoxListSink box_sink;
    bool tiled = false;
    double h_est = 0.0, d_est = 3.0;  // measured (clouds that do not fill their box): the radius holding M = 1.75 k points, the local dimension
    unsigned long long* scratch3 = (unsigned long long*)((uint8_t*)counters.p + 88);  // four counters of the probe / census kernels (88 .. 120)
    double m_target = 1.75 * (double)k;  // points the ball of radius h should hold
    if (tune.tau_m > 0) m_target = tune.tau_m;
    // GATE: is the cloud what its bounding box says?  A quick scale estimate on 2^17 points, 256 queries (normals_scale.hip; 0.3 ms) against the radius
    // the box's volume predicts.  The 32^3 occupancy mask alone is fooled by a thin uniform halo around a dense core (1 % of the points
    // spread over 10^5 times the core's volume fill every coarse cell: the grid was laid for the halo and every cell of the core held
    // 400 000 points -- 32 s for 10^7 points).
    double h_gate = 0.0, d_gate = 3.0;
    bool concentrated = false;
    if (n >= 4096 && !tune.forced_scale() && !tune.no_scale) {
      const uint64_t S = (n + (1u << 17) - 1) >> 17, n_g = n / S;
      CacheBuf xyz_g, hist_g;
      NCK(xyz_g.alloc(n_g * 24, stream));
      NCK(hist_g.alloc(knn_scale_scratch_bytes(), stream));
      hipLaunchKernelGGL(gather_positions_kernel, dim3(sgrid), dim3(kBlock), 0, stream, (const uint8_t*)xyz.as<double>(), 24 * S, n_g, xyz_g.as<double>(), partials.as<double>());
      if (knn_scale_estimate(xyz_g.as<double>(), (uint32_t)n_g, (double)S, ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2], m_target, hist_g.as<unsigned int>(), stream,
                             h_gate, d_gate, 256)) {
        const double h_box = edge_for(m_target / kBallVolume);
        concentrated = cloud_is_concentrated(h_gate, h_box);
        if (debug) fprintf(stderr, "[pst knn gate] %.1f points within h=%g (dimension %.2f); the box's volume says %g%s\n", m_target, h_gate, d_gate, h_box,
                           concentrated ? ": concentrated" : "");
      }
      mark("gate");
    }
    // fraction of the 32^3 coarse cells of the box that hold a point (flat axes count as one layer), and -- from point counts per slice of
    // every axis, taken in the same pass -- a TRIMMED box: the smallest slice ranges that hold all but 0.05 % of the points at either end,
    // one slice added on each side.  It replaces the bounding box only when it is at least 8 times smaller, i.e. when a few far points
    // stretch the bounding box (64 outliers around 10^7 points: the cloud filled 0.001 % of it, no dense directory fitted, 42 ms instead
    // of 5).  Points outside it are clamped into the boundary cells like the box's own last points; the searches stay exact (a clamped
    // point lies beyond its cell, never nearer), the box kernel hands queries outside the box to the global-memory search.
    double occupancy = 0.0;
    auto measure_box = [&]() -> bool {
#define MCK(x) do { if ((x) != hipSuccess) return false; } while (0)
    for (int pass = 0; pass < 4; ++pass) {
      CacheBuf occ;
      const size_t occ_bytes = (size_t)kOccWords * 4 + 3 * kAxisBins * 4;
      MCK(occ.alloc(occ_bytes, stream));
      MCK(hipMemsetAsync(occ.p, 0, occ_bytes, stream));
      double sc[3], ax[3];
      for (int c = 0; c < 3; ++c) {
        sc[c] = ext[c] > maxext * 1e-9 ? (double)kOccBins / ext[c] * (1.0 - 1e-12) : 0.0;
        ax[c] = ext[c] > maxext * 1e-9 ? (double)kAxisBins / ext[c] * (1.0 - 1e-12) : 0.0;
      }
      uint32_t* axis_hist = occ.as<uint32_t>() + kOccWords;
      const uint64_t S_occ = tune.occupancy_all ? 1 : (n + ((uint64_t)1 << 23) - 1) >> 23;  // (every S-th point: at most 2^23 of them)
      hipLaunchKernelGGL(occupancy_kernel, dim3(std::min(sgrid, cus * 4)), dim3(kBlock), 0, stream, xyz.as<double>(), n / std::max<uint64_t>(S_occ, 1), std::max<uint64_t>(S_occ, 1), mn[0], mn[1], mn[2], sc[0], sc[1], sc[2],
                         occ.as<uint32_t>(), ax[0], ax[1], ax[2], axis_hist, frame);
      std::vector<uint32_t> hb(kOccWords + 3 * kAxisBins);
      MCK(hipMemcpyAsync(hb.data(), occ.p, occ_bytes, hipMemcpyDeviceToHost, stream));
      MCK(hipStreamSynchronize(stream));
      uint64_t set = 0;
      for (uint32_t w = 0; w < kOccWords; ++w) set += (uint64_t)__builtin_popcount(hb[w]);
      double bins = 1.0;
      for (int c = 0; c < 3; ++c) bins *= sc[c] > 0 ? (double)kOccBins : 1.0;
      occupancy = (double)set / bins;
      if (debug) fprintf(stderr, "[pst knn] occupancy of the box at 32^3: %.3f\n", occupancy);
      // The trimmed box for a tail mass `cut` (a fraction of the points at either end of every axis).  0.05 % always; for a cloud the gate
      // found concentrated also 0.5 % and 5 % -- a tenfold cut is taken when it buys at least an eightfold smaller box (a halo).
      auto trimmed = [&](double cut_frac, double (&tmn)[3], double (&tmx)[3]) { return trimmed_box<kAxisBins>(hb.data() + kOccWords, mn, mx, ax, cut_frac, tmn, tmx); };
      double tmn[3], tmx[3];
      double shrink = trimmed(0.0005, tmn, tmx);
      if (concentrated) {
        for (double cut_frac : {0.005, 0.05}) {
          double umn[3], umx[3];
          const double sh = trimmed(cut_frac, umn, umx);
          if (sh <= shrink / 8.0) { shrink = sh; for (int c = 0; c < 3; ++c) { tmn[c] = umn[c]; tmx[c] = umx[c]; } }
          else break;
        }
      }
      // (once a box has been trimmed, the slices are finer and a second and third look may tighten it further: any gain above 40 % is taken)
      if (take_trimmed_box(pass, shrink) && !tune.no_trim) {
        if (debug) fprintf(stderr, "[pst knn] trimmed box: %.3g of the volume: [%g, %g] x [%g, %g] x [%g, %g]\n", shrink, tmn[0], tmx[0], tmn[1], tmx[1], tmn[2], tmx[2]);
        for (int c = 0; c < 3; ++c) { mn[c] = tmn[c]; mx[c] = tmx[c]; }
        set_box();
        continue;
      }
      break;
    }
    return true;
#undef MCK
    };
    if (!measure_box()) return -1;
    mark("occupancy");
    // PRINCIPAL AXES.  A cloud that is thin along a direction which is not a coordinate axis (a tilted facade, a diagonal flight strip, a
    // helix) fills little of any axis-aligned box: the dense directory over that box would exceed its budget and the search fall back to
    // the hash directory.  The covariance of a subsample (inside the trimmed box) gives the principal axes; if the box in THAT frame is
    // at most a third of the volume, the grid is laid in it: cells, rows and trims use rotated coordinates (grid_frame), distances the
    // original ones.
    if (consider_rotation(occupancy, n, tune)) {
      const uint64_t S_m = std::max<uint64_t>(1, n >> 20), n_m = n / S_m;
      CacheBuf sums;
      NCK(sums.alloc(80, stream));
      NCK(hipMemsetAsync(sums.p, 0, 80, stream));
      const double ctr[3] = {0.5 * (mn[0] + mx[0]), 0.5 * (mn[1] + mx[1]), 0.5 * (mn[2] + mx[2])};
      hipLaunchKernelGGL(moments_kernel, dim3(std::min(sgrid, cus * 4)), dim3(kBlock), 0, stream, xyz.as<double>(), n_m, S_m, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], ctr[0],
                         ctr[1], ctr[2], sums.as<double>());
      double hs[10];
      NCK(hipMemcpyAsync(hs, sums.p, 80, hipMemcpyDeviceToHost, stream));
      NCK(hipStreamSynchronize(stream));
      if (hs[9] >= 1000.0) {
        const double c = hs[9], m0 = hs[0] / c, m1 = hs[1] / c, m2 = hs[2] / c;
        double A[3][3] = {{hs[3] / c - m0 * m0, hs[4] / c - m0 * m1, hs[5] / c - m0 * m2}, {0, hs[6] / c - m1 * m1, hs[7] / c - m1 * m2}, {0, 0, hs[8] / c - m2 * m2}};
        A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
        double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // columns = eigenvectors (cyclic Jacobi)
        for (int sweep = 0; sweep < 30; ++sweep) {
          const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
          if (!(off > 1e-300) || off < 1e-14 * (std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]))) break;
          for (int pi = 0; pi < 2; ++pi)
            for (int qi = pi + 1; qi < 3; ++qi) {
              if (A[pi][qi] == 0.0) continue;
              const double theta = (A[qi][qi] - A[pi][pi]) / (2.0 * A[pi][qi]);
              const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
              for (int r = 0; r < 3; ++r) { const double arp = A[r][pi], arq = A[r][qi]; A[r][pi] = cs * arp - sn * arq; A[r][qi] = sn * arp + cs * arq; }
              for (int r = 0; r < 3; ++r) { const double apr = A[pi][r], aqr = A[qi][r]; A[pi][r] = cs * apr - sn * aqr; A[qi][r] = sn * apr + cs * aqr; }
              for (int r = 0; r < 3; ++r) { const double vrp = V[r][pi], vrq = V[r][qi]; V[r][pi] = cs * vrp - sn * vrq; V[r][qi] = sn * vrp + cs * vrq; }
            }
        }
        int order[3] = {0, 1, 2};  // largest variance first: the grid's rows (x) run along the cloud's longest direction
        std::sort(order, order + 3, [&](int a, int b) { return A[a][a] > A[b][b]; });
        GridParams cand{};
        cand.rotated = 1;
        for (int cc = 0; cc < 3; ++cc) cand.rot_c[cc] = ctr[cc];
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) cand.rot[3 * r + cc] = V[cc][order[r]];
        double align = 1.0;  // the smallest of the rows' largest components: 1 = the principal axes ARE the coordinate axes (in some order)
        for (int r = 0; r < 3; ++r) align = std::fmin(align, std::fmax(std::fabs(cand.rot[3 * r]), std::fmax(std::fabs(cand.rot[3 * r + 1]), std::fabs(cand.rot[3 * r + 2]))));
        if (axes_are_coordinate_axes(align)) { mark("axes"); goto axes_done; }  // within 6 degrees: nothing to gain, no pass over the points
        hipLaunchKernelGGL(framed_bounds_kernel, dim3(sgrid), dim3(kBlock), 0, stream, xyz.as<double>(), n, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], cand, partials.as<double>());
        NCK(hipMemcpyAsync(hp.data(), partials.p, hp.size() * 8, hipMemcpyDeviceToHost, stream));
        NCK(hipStreamSynchronize(stream));
        double rmn[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308}, rmx[3] = {-rmn[0], -rmn[0], -rmn[0]};
        for (unsigned b = 0; b < sgrid; ++b)
          for (int cc = 0; cc < 3; ++cc) { rmn[cc] = std::fmin(rmn[cc], hp[b * 6 + cc]); rmx[cc] = std::fmax(rmx[cc], hp[b * 6 + 3 + cc]); }
        if (rmn[0] <= rmx[0]) {
          double v_now = 1.0, v_rot = 1.0;
          double rext = std::fmax(rmx[0] - rmn[0], std::fmax(rmx[1] - rmn[1], rmx[2] - rmn[2]));
          for (int cc = 0; cc < 3; ++cc) { v_now *= std::fmax(ext[cc], 1e-6 * maxext); v_rot *= std::fmax(rmx[cc] - rmn[cc], 1e-6 * rext); }
          if (debug) fprintf(stderr, "[pst knn] principal axes: box %.3g of the axis-aligned one (%g x %g x %g)\n", v_rot / v_now, rmx[0] - rmn[0], rmx[1] - rmn[1], rmx[2] - rmn[2]);
          if (take_rotated_box(v_rot, v_now)) {
            frame = cand;
            for (int cc = 0; cc < 3; ++cc) { mn[cc] = rmn[cc]; mx[cc] = rmx[cc]; }
            set_box();
            if (!measure_box()) return -1;
          }
        }
      }
      mark("axes");
    }
    axes_done:;
    CacheBuf xyz_s;  // a subsample of the cloud (packed xyz; may hold non-finite points): the scale estimate and the bounds of the all-points search
    uint64_t n_sub = 0;
    // The bounding box's volume gives the right h only for clouds that fill it.  For the others (a surface: the first guess is several
    // times too large; separate clusters: orders of magnitude) the scale is MEASURED first, without an index (normals_scale.hip): distance
    // histograms of 512 sampled points against a subsample of 2^20 to 2^22 points give the radius at which the cloud holds M points
    // around a typical point, and its local dimension.
    const double h_box_now = edge_for(m_target / kBallVolume);
    const bool fills = cloud_fills_box(occupancy, h_gate, h_box_now);  // the cloud is what its (trimmed) box says
    // (up to 4 million points the gate's subsample is thinned by at most 32 -- as good as the full estimate: taken as it is)
    const bool gate_enough = gate_is_enough(h_gate, n);
    if (!fills && gate_enough) { h_est = h_gate; d_est = d_gate; }
    if (!fills && !gate_enough && n >= 4096 && !tune.forced_scale() && !tune.no_scale) {
      // the subsample: a sixteenth of the cloud, at least 2^18 and at most 2^22 points (the further the thinning, the longer the extrapolation
      // down to the radius of M points: at 1 in 96 the sheet's h came out 8 % low)
      const uint64_t cap_s = std::min<uint64_t>(1u << 22, std::max<uint64_t>(1u << 18, n / 16));
      const uint64_t S = (n + cap_s - 1) / cap_s, n_s = n / S;
      CacheBuf hist_s;
      NCK(xyz_s.alloc(n_s * 24, stream));
      n_sub = n_s;
      NCK(hist_s.alloc(knn_scale_scratch_bytes(), stream));
      hipLaunchKernelGGL(gather_positions_kernel, dim3(sgrid), dim3(kBlock), 0, stream, (const uint8_t*)xyz.as<double>(), 24 * S, n_s, xyz_s.as<double>(),
                         partials.as<double>());
      double h_m = 0, dim_m = 3;
      if (knn_scale_estimate(xyz_s.as<double>(), (uint32_t)n_s, (double)S, ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2], m_target, hist_s.as<unsigned int>(), stream,
                             h_m, dim_m)) {
        if (debug) fprintf(stderr, "[pst knn scale] %llu of %llu points: %.1f points within h=%g, dimension %.2f (by the box's volume: %g)\n", (unsigned long long)n_s,
                           (unsigned long long)n, m_target, h_m, dim_m, edge_for(m_target / kBallVolume));
        h_est = h_m; d_est = dim_m;
      }
      mark("scale");
    }
    if (try_box_search(k, occupancy, n, tune)) {
      // fine x cells per h: 4 for clouds that fill their box; 2 for the others (a surface: the same box holds fewer points, the 31-cell limit of a
      // box row then makes boxes too short at rx = 4: 6.3 against 3.9 ms per 10^7 points of the sheet in tools/exp_normals_surface.py)
      uint32_t rx = fine_cells_per_h(h_est, d_est, occupancy, tune);
      // points per (cubic) cell of edge R0: M = (4/3 pi) R0^3 * density  =>  R0^3 * density = M / (4/3 pi)
      double h = h_est > 0.0 ? h_est : edge_for(per_cell_env > 0 ? per_cell_env : m_target / kBallVolume);
      for (int round = 0; round < 3; ++round) {
        GridParams trial{};
        // (clustered clouds and surfaces leave cells empty: 4 bytes each, up to 20 per point are accepted here)
        uint64_t trial_cells = grid_for(h, rx, trial);
        // (default 20 cells per point: 80 bytes per point; 12 left the 10^8-point sheet at rx = 1: 84 against 71 ms)
        const uint64_t cell_budget = directory_budget(n, tune);
        while (rx > 1 && !directory_fits(trial_cells, cell_budget)) { rx >>= 1; trial_cells = grid_for(h, rx, trial); }  // coarser x cells before giving up
        if (!directory_fits(trial_cells, cell_budget)) break;
        if (!build_index(h, rx, true)) return -1;
        mark("index");
        if (!nf) break;
        double m_half = 0, m_full = 0;
        if (!knn_probe(sorted_xyz.as<double>(), directory.as<uint32_t>(), g, (uint32_t)nf, scratch3, stream, m_half, m_full)) return -1;
        // N(r) ~ r^D through (h/2, m_half) and (h, m_full); the radius that holds M points
        mark("probe");
        const ProbeFit pf = probe_fit(g.h, m_half, m_full, m_target);
        const double D = pf.dim, h_new = pf.h_new;
        if (debug) fprintf(stderr, "[pst knn probe] h=%g: %.1f points within h/2, %.1f within h (target %.1f), dimension %.2f -> h=%g\n", g.h, m_half, m_full, m_target, D, h_new);
        if (probe_accepts(round, g.h, h_new, tune)) {
          // (clouds that do not fill their box: the census of the winning shape also lists the boxes that hold a query)
          box_sink.alloc = [&](size_t bytes) -> uint32_t* { CacheBuf b; return b.alloc(bytes, stream) == hipSuccess ? b.as<uint32_t>() : nullptr; };
          box_sink.count_dev = (uint32_t*)((uint8_t*)counters.p + 76);
          box_sink.list = nullptr; box_sink.n = 0;
          box_sink.sorted_cells = keys2.as<uint32_t>();  // (dense grids: 32-bit row-major cell numbers, sorted)
          tiled = knn_tile_shape(g, nf, cells, k, fills, directory.as<uint32_t>(), scratch3, stream, shape, (!fills && tune.box_list) ? &box_sink : nullptr);
          // Which plane fit the box search runs.  One pass about the query (plane_fit_pivot) agrees with the reference's two passes to a few
          // ulps of the covariance -- enough wherever the neighbourhood spans three dimensions.  On surfaces and strips the neighbourhoods are
          // nearly planar, the smallest eigenvalue is the difference of large numbers in the reference's cubic solver, and those ulps become
          // 1e-8 of the curvature (the deep fuzz: one curvature of 1.4 10^5 on a strip, 1.4e-12 absolute): such clouds keep the reference's
          // ORDER of operations, which reproduces its rounding.
          shape.fit_seq = tune.fit >= 0 ? tune.fit == 1 : !fills;
          mark("census");
          break;
        }
        h = h_new;
      }
    }
    bool dense = tiled;
    if (!tiled) {
      // Cubic cells of ~k/12 points (dense directory) or ~k/3 points (hash table) -- by the bounding box's volume, or, where the scale was
      // MEASURED above (a cloud that does not fill its box and did not fit the box search's directory budget either), by that: the ball of
      // radius h_est holds M points and N(r) ~ r^D; a cubic cell of edge h holds about what a ball of radius c_D h does (c = 0.62 in 3-D,
      // 0.56 in 2-D, 0.5 in 1-D).  A cloud of dimension D occupies 3^D of the 27 cells of the first shell, so the cell's share is scaled by
      // 3^(3-D): the first shell then holds the same number of candidates whatever the dimension.
      const FallbackEdges fe = fallback_edges(bs, n, k, h_est, d_est, m_target, tune);
      const double h_dense = fe.dense, h_hash = fe.hash;
      if (debug && h_est > 0.0 && !tune.forced_scale())
        fprintf(stderr, "[pst knn] measured scale: cell edge %g (dense) / %g (hash) instead of the bounding box's\n", h_dense, h_hash);
      GridParams trial{};
      dense = is_dense(grid_for(h_dense, 1, trial));
      if (tune.dense == 0) dense = false;
      if (!build_index(dense ? h_dense : h_hash, 1, dense)) return -1;
    }
    CacheBuf unres, open_lists;  // one byte per point, by ORIGINAL index: the query was handed back by a capped search; and their sorted positions
    constexpr uint32_t kOpenCap = 1u << 22;  // listed open queries per kind (beyond that the flags are collected in a pass over all points)
    NCK(open_lists.alloc((size_t)2 * kOpenCap * 4, stream));
    NCK(unres.alloc(n, stream));
    NCK(hipMemsetAsync(unres.p, 0, n, stream));
    uint32_t* unres_count = (uint32_t*)((uint8_t*)counters.p + 64);
    // results: straight into the caller's outputs (default), or as 32-byte records + split_results_kernel (PST_KNN_DIRECT=0: the A/B switch)
    const bool direct_out = tune.direct_out;
    if (!direct_out) NCK(rec.alloc(n * 32, stream));
    RecOut sorted{direct_out ? nullptr : rec.as<double>(), idx2.as<uint32_t>(), out.knn, out.knn_u32, out.error_count,
                  out.normals_f64, out.curvature_f64, out.normal_attr, out.normal_stride, out.curv_attr, out.curv_stride};
    if (nf) {
      if (dense) {
        const uint32_t* cell_start = directory.as<uint32_t>();
        if (tiled) {
          // box kernel first; what it hands back (k-th distance beyond tau0, ambiguous packed keys, boxes denser than its LDS budget) goes to
          // the global-memory search as a list
          NCK(fb_list.alloc(nf * 4, stream));
          NCK(hipMemsetAsync(fb_count, 0, 4, stream));
          // clouds that do not fill their box launch one workgroup per box that holds a query (a sheet leaves two thirds of them empty)
          CacheBuf box_list;
          uint32_t n_list = 0;
          const uint32_t* list_ptr = nullptr;
          if (!fills && tune.box_list && box_sink.list) {
            list_ptr = box_sink.list;
            n_list = box_sink.n;
            if (debug) fprintf(stderr, "[pst knn] %u of %u boxes hold a query (listed by the census)\n", n_list, knn_box_count(shape, g));
          } else if (!fills && tune.box_list) {
            NCK(box_list.alloc((size_t)knn_box_count(shape, g) * 4, stream));
            n_list = knn_box_list(shape, cell_start, g, box_list.as<uint32_t>(), unres_count + 3, stream);
            if (n_list == 0xFFFFFFFFu) return -1;
            list_ptr = box_list.as<uint32_t>();
            if (debug) fprintf(stderr, "[pst knn] %u of %u boxes hold a query\n", n_list, knn_box_count(shape, g));
            mark("box-list");
          }
          launch_knn_tile(shape, sorted_xyz.as<double>(), cell_start, g, k, (uint32_t)nf, sorted, fb_list.as<uint32_t>(), fb_count, list_ptr, n_list, stream);
          uint32_t n_fb = 0;
          NCK(hipMemcpyAsync(&n_fb, fb_count, 4, hipMemcpyDeviceToHost, stream));
          NCK(hipStreamSynchronize(stream));
          rec_tiled = true; rec_list = list_ptr != nullptr; rec_n_list = n_list; rec_n_fb = n_fb; rec_shape = shape; rec_g = g; rec_cells = cells; rec_nf = nf;
          mark("box-search");
          if (debug)
            fprintf(stderr, "[pst knn] fit=%s n=%llu nf=%llu cells=%llu dim=%ux%ux%u h=%g box=%ux%ux%u kernel=%c threads=%u cap=%u fallback=%u\n", shape.fit_seq ? "reference-order" : "one-pass+guard", (unsigned long long)n,
                    (unsigned long long)nf, (unsigned long long)cells, g.dim[0], g.dim[1], g.dim[2], g.h, shape.bx, shape.by, shape.bz, shape.tag, shape.threads,
                    shape.cap, n_fb);
          if (n_fb) {
            // The box kernel appends its leftovers in the order its workgroups finish: 64 consecutive entries come from 64 different boxes.
            // Sorted by position (= by cell) the lanes of a wave search neighbouring cells and share their candidates' cache lines
            // (PST_KNN_SORT_FALLBACK=0: the A/B switch; same box, 10^8 points: uniform cloud 32.9 -> 32.45 ms, but the sheet 45.0 -> 46.2 --
            // volume-like clouds only).  The unsorted key / index buffers of the index build are free by now.
            const uint32_t* fb_q = fb_list.as<uint32_t>();
            const bool sort_fb = tune.sort_fallback;
            if (sort_fb && fills && n_fb >= 4096) {
              uint32_t *ka = fb_list.as<uint32_t>(), *kb = keys.as<uint32_t>(), *va = keys.as<uint32_t>() + n, *vb = idx.as<uint32_t>();
              unsigned bits = 1;
              while (bits < 32 && (1ull << bits) <= nf) ++bits;
              size_t sb = 0;
              // (only the sorted keys are used: iota = true lets the first scatter number the values instead of reading `va`, which is uninitialised scratch)
              NCK(sort_pairs_u32(nullptr, sb, ka, kb, va, vb, n_fb, bits, stream, true));
              NCK(tmp.alloc(sb, stream));
              NCK(sort_pairs_u32(tmp.p, sb, ka, kb, va, vb, n_fb, bits, stream, true));
              fb_q = kb;
            }
            const unsigned grid = (unsigned)((n_fb + kBlock - 1) / kBlock);
            KNN_DISPATCH_GRID(true, true, grid, sorted_xyz.as<double>(), (const uint64_t*)nullptr, (uint32_t)nf, k, g, table, cell_start,
                              fb_q, n_fb, sorted, kShellCap, unres.as<uint8_t>(), unres_count, 0u, open_lists.as<uint32_t>(), kOpenCap, (const uint32_t*)nullptr);
          }
        } else {
          const unsigned grid = (unsigned)((nf + kBlock - 1) / kBlock);
          KNN_DISPATCH_GRID(true, false, grid, sorted_xyz.as<double>(), (const uint64_t*)nullptr, (uint32_t)nf, k, g, table, cell_start,
                            (const uint32_t*)nullptr, (uint32_t)nf, sorted, kShellCap, unres.as<uint8_t>(), unres_count, 0u, open_lists.as<uint32_t>(), kOpenCap, (const uint32_t*)nullptr);
        }
      } else {
        const unsigned grid = (unsigned)((nf + kBlock - 1) / kBlock);
        KNN_DISPATCH_GRID(false, false, grid, sorted_xyz.as<double>(), (const uint64_t*)keys2.as<uint64_t>(), (uint32_t)nf, k, g, table,
                          (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t)nf, sorted, kShellCap, unres.as<uint8_t>(), unres_count, 0u, open_lists.as<uint32_t>(), kOpenCap, (const uint32_t*)nullptr);
      }
      // Queries a capped search handed back (flag 1: outliers; regions far sparser than the grid was made for): again on a grid with six
      // times the cell edge, laid over the FULL bounding box (the trimmed or rotated box of the first level clamps exactly the points these
      // queries are made of) -- its first shell covers what six shells of the last one did.  On such a grid a cell over a dense part of
      // the cloud holds millions of points and a grid search walks a cell with ONE lane: a query that meets a range of more than kCrowd
      // points gives up (flag 2) and is searched exactly against all points by a workgroup (bound / filter / select above), as are the
      // last few open ones, whose all-points search costs less than another index.
      auto all_points = [&](uint8_t which, uint32_t n_q) -> bool {
#define ACK(x) do { if ((x) != hipSuccess) return false; } while (0)
        const uint32_t* open_q = nullptr;
        if (n_q <= kOpenCap) {
          // the search that flagged them listed their sorted positions (same index: nothing was re-sorted in between): no pass over all points
          open_q = open_lists.as<uint32_t>() + (which == 2 ? kOpenCap : 0u);
          hipLaunchKernelGGL(clear_flags_kernel, dim3((n_q + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, open_q, n_q, (const uint32_t*)idx2.as<uint32_t>(), unres.as<uint8_t>());
        } else {
          ACK(fb_list.alloc((size_t)nf * 4, stream));
          ACK(hipMemsetAsync(unres_count + 2, 0, 4, stream));
          hipLaunchKernelGGL(collect_unresolved_kernel, dim3((unsigned)((nf + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, (const uint32_t*)idx2.as<uint32_t>(), (uint32_t)nf,
                             unres.as<uint8_t>(), which, fb_list.as<uint32_t>(), unres_count + 2);
          open_q = fb_list.as<uint32_t>();
        }
        if (debug) fprintf(stderr, "[pst knn] %u open queries against all %llu points\n", n_q, (unsigned long long)nf);
        if (!n_sub) {  // (clouds that fill their box had no scale estimate: take the subsample now)
          const uint64_t cap_s = std::min<uint64_t>(1u << 22, std::max<uint64_t>(1u << 20, n / 16));
          const uint64_t S = (n + cap_s - 1) / cap_s;
          n_sub = n / S;
          ACK(xyz_s.alloc(n_sub * 24, stream));
          hipLaunchKernelGGL(gather_positions_kernel, dim3(sgrid), dim3(kBlock), 0, stream, (const uint8_t*)xyz.as<double>(), 24 * S, n_sub, xyz_s.as<double>(),
                             partials.as<double>());
        }
        const uint32_t cand_cap = 4096, batch = 16384;  // (16 KB of candidate list per open query: 256 MB per batch)
        CacheBuf bound, cand_count, cand, qpack;
        const uint32_t n_b = std::min(n_q, batch);
        ACK(bound.alloc((size_t)n_b * 8, stream));
        ACK(qpack.alloc((size_t)n_b * 32, stream));
        ACK(cand_count.alloc((size_t)n_b * 4, stream));
        ACK(cand.alloc((size_t)n_b * cand_cap * 4, stream));
        for (uint32_t off = 0; off < n_q; off += batch) {
          const uint32_t cnt = std::min(batch, n_q - off);
          const uint32_t* ql = open_q + off;
          ACK(hipMemsetAsync(cand_count.p, 0, (size_t)cnt * 4, stream));
          // (the first quarter of the subsample -- itself a uniform thinning, in input order -- is enough for the bound: four times the
          // candidates per query, which the culled filter and the select kernel barely notice, for a quarter of the scan)
          const uint32_t n_bound = (uint32_t)std::max<uint64_t>(n_sub / 4, std::min<uint64_t>(n_sub, 1u << 18));
          KNN_DISPATCH(knn_bound_kernel, cnt, sorted_xyz.as<double>(), ql, cnt, (const double*)xyz_s.as<double>(), n_bound, k, bound.as<double>());
          hipLaunchKernelGGL(knn_pack_queries_kernel, dim3((cnt + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, (const double*)sorted_xyz.as<double>(), ql, cnt,
                             (const double*)bound.as<double>(), qpack.as<double>());
          hipLaunchKernelGGL(knn_filter_kernel, dim3((unsigned)((nf + kBlock * kFilterPts - 1) / (kBlock * kFilterPts))), dim3(kBlock), 0, stream,
                             (const double*)sorted_xyz.as<double>(), (uint32_t)nf, (const double*)qpack.as<double>(), cnt, cand_cap, cand_count.as<uint32_t>(),
                             cand.as<uint32_t>());
          KNN_DISPATCH(knn_select_kernel, cnt, sorted_xyz.as<double>(), (uint32_t)nf, k, ql, cnt, (const uint32_t*)cand_count.as<uint32_t>(),
                       (const uint32_t*)cand.as<uint32_t>(), cand_cap, sorted);
        }
        mark("all-points");
        return true;
#undef ACK
      };
      for (int level = 1; nf; ++level) {
        uint32_t n_open[2] = {0, 0};  // handed back by the shell cap / by the crowd guard
        NCK(hipMemcpyAsync(n_open, unres_count, 8, hipMemcpyDeviceToHost, stream));
        NCK(hipStreamSynchronize(stream));
        NCK(hipMemsetAsync(unres_count, 0, 8, stream));
        if (n_open[0] || n_open[1]) rec_open = true;
        if (n_open[1] && !all_points(2, n_open[1])) return -1;
        const uint32_t n_un = n_open[0];
        if (!n_un) break;
        if (level == 1) {  // from here on: the cloud's own axes and its full bounding box
          frame = GridParams{};
          for (int c = 0; c < 3; ++c) { mn[c] = full_mn[c]; mx[c] = full_mx[c]; }
          set_box();
        }
        const double h_up = g.h * (double)kShellCap;
        GridParams trial{};
        const uint64_t up_cells = grid_for(h_up, 1, trial);
        const bool last = std::max(trial.dim[0], std::max(trial.dim[1], trial.dim[2])) <= (uint32_t)kShellCap + 1u;
        const bool up_dense = is_dense(up_cells);
        // all points or another level?  The all-points search does ~2e12 pairs per second; an index costs ~2 ns per point with a dense
        // directory and ~4.5 ns with the hash table (64-bit Morton keys, eight radix passes, the table), and may leave queries open.
        if (search_all_points(level, n_un, nf, up_dense)) {
          if (!all_points(1, n_un)) return -1;
          break;
        }
        if (debug) fprintf(stderr, "[pst knn] level %d: %u open queries, cell edge %g (%s)%s\n", level, n_un, h_up, up_dense ? "dense" : "hash", last ? ", uncapped" : "");
        if (!build_index(h_up, 1, up_dense)) return -1;
        dense = up_dense;
        NCK(fb_list.alloc((size_t)nf * 4, stream));
        NCK(hipMemsetAsync(unres_count + 2, 0, 4, stream));
        hipLaunchKernelGGL(collect_unresolved_kernel, dim3((unsigned)((nf + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, (const uint32_t*)idx2.as<uint32_t>(), (uint32_t)nf,
                           unres.as<uint8_t>(), (uint8_t)1, fb_list.as<uint32_t>(), unres_count + 2);
        const unsigned grid = (unsigned)((n_un + kBlock - 1) / kBlock);
        const int cap = last ? 0 : kShellCap;
        if (up_dense) {
          KNN_DISPATCH_GRID(true, true, grid, sorted_xyz.as<double>(), (const uint64_t*)nullptr, (uint32_t)nf, k, g, table, (const uint32_t*)directory.as<uint32_t>(),
                            (const uint32_t*)fb_list.as<uint32_t>(), n_un, sorted, cap, unres.as<uint8_t>(), unres_count, kCrowd, open_lists.as<uint32_t>(), kOpenCap, (const uint32_t*)nullptr);
        } else {
          KNN_DISPATCH_GRID(false, true, grid, sorted_xyz.as<double>(), (const uint64_t*)keys2.as<uint64_t>(), (uint32_t)nf, k, g, table, (const uint32_t*)nullptr,
                            (const uint32_t*)fb_list.as<uint32_t>(), n_un, sorted, cap, unres.as<uint8_t>(), unres_count, kCrowd, open_lists.as<uint32_t>(), kOpenCap, (const uint32_t*)nullptr);
        }
        mark("coarser");
      }
    }
    if (nf < n) {
      // non-finite query points: every distance is NaN (-> +inf), so "the k nearest" is the reference's kd-tree tie order
      // (unpinned).  Chosen here: the point itself, then the first k-1 finite points in sorted order.
      const unsigned grid = (unsigned)((n - nf + kBlock - 1) / kBlock);
      hipLaunchKernelGGL(knn_nonfinite_kernel, dim3(grid), dim3(kBlock), 0, stream, xyz.as<double>(), sorted_xyz.as<double>(), (uint32_t)nf, (uint32_t)n, k,
                         sorted);
    }
    mark("fallback");
    if (!direct_out) {
      hipLaunchKernelGGL(split_results_kernel, dim3(sgrid), dim3(kBlock), 0, stream, (const double*)rec.as<double>(), n, out);
      mark("split");
    }
    if (trace) fprintf(stderr, "[pst knn trace]%s\n", trace_line.c_str());
    NCK(hipGetLastError());
  }
  NCK(hipGetLastError());
  int errors = 0;
  NCK(hipMemcpyAsync(&errors, out.error_count, 4, hipMemcpyDeviceToHost, stream));
  NCK(hipStreamSynchronize(stream));
  if (record && rec_tiled) {
    if (rec_open) record->why_not = "some queries were handed back by a capped search (far points, sparse regions): the coarser levels are host-driven";
    else if (rec_nf != n) record->why_not = "the cloud holds non-finite points";
    else if (!knn_tuning().direct_out) record->why_not = "PST_KNN_DIRECT=0";
    else if (out_knn_dev) record->why_not = "int64 neighbour lists are a host-side format";
    else {
      record->valid = true; record->why_not = "";
      record->n = n; record->nf = rec_nf; record->cells = rec_cells; record->k = k; record->g = rec_g; record->shape = rec_shape;
      record->use_list = rec_list; record->n_list = rec_n_list; record->n_fb = rec_n_fb;
      unsigned kb = 1; while (kb < 32 && (1ull << kb) <= rec_cells) ++kb;
      record->key_bits = kb;
    }
  }
#undef KNN_DISPATCH_GRID
#undef KNN_DISPATCH
#undef KNN_DISPATCH_T
#undef NCK
  return errors;
}


// ---- stream-ordered replay of a recorded call ------------------------------------------------------------------------------------------------
struct KnnPlan {
  KnnPlanRecord rec;
  bool packed_source = true;
  DevBuf xyz_own, partials, keys, keys2, idx, idx2, sorted_xyz, tmp, directory, dir_blocks, fb_list, box_list, counters, unres, open_lists;
  size_t tmp_sort = 0, tmp_suffix = 0;
  uint32_t list_cap = 0, fb_cap = 0, all_boxes = 0;
  uint64_t n_dblocks = 0;
  unsigned sgrid = 1;
};

namespace {
constexpr uint32_t kReplayOpenCap = 1u << 16;
// the device-side end of a replay: what the synchronous call checks on the host
__global__ void knn_replay_status_kernel(const unsigned long long* __restrict__ n_finite, unsigned long long nf, const uint32_t* __restrict__ fb_count, uint32_t fb_cap,
                                         const uint32_t* __restrict__ box_count, uint32_t list_cap, const uint32_t* __restrict__ unres_count,
                                         const int* __restrict__ error_count, unsigned long long* __restrict__ status2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long st = 0;
  if (*n_finite != nf) st |= KNN_STATUS_FINITE_COUNT;
  if (box_count && *box_count > list_cap) st |= KNN_STATUS_BOX_CAPACITY;
  if (*fb_count > fb_cap) st |= KNN_STATUS_FALLBACK_CAPACITY;
  if (unres_count[0] || unres_count[1]) st |= KNN_STATUS_OPEN_QUERIES;
  if (*error_count) st |= KNN_STATUS_DEGENERATE;
  status2[0] = st;
  status2[1] = (unsigned long long)*error_count;
}
}  // namespace

KnnPlan* knn_plan_create(const KnnPlanRecord& rec, bool packed_source, hipStream_t stream) {
  if (!rec.valid) return nullptr;
  auto p = std::make_unique<KnnPlan>();
  p->rec = rec;
  p->packed_source = packed_source;
  const uint64_t n = rec.n;
  const unsigned cus = (unsigned)device_cus();
  p->sgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)cus * 8));
#define PCK(x) do { if ((x) != hipSuccess) return nullptr; } while (0)
  if (!packed_source) { PCK(p->xyz_own.alloc(n * 24, stream)); PCK(p->partials.alloc((size_t)p->sgrid * 48, stream)); }
  PCK(p->counters.alloc(128, stream));
  PCK(p->keys.alloc(n * 4, stream)); PCK(p->keys2.alloc(n * 4, stream)); PCK(p->idx.alloc(n * 4, stream)); PCK(p->idx2.alloc(n * 4, stream));
  PCK(p->sorted_xyz.alloc(n * 24, stream));
  PCK(sort_pairs_u32(nullptr, p->tmp_sort, p->keys.as<uint32_t>(), p->keys2.as<uint32_t>(), p->idx.as<uint32_t>(), p->idx2.as<uint32_t>(), n, rec.key_bits, stream));
  PCK(p->directory.alloc((rec.cells + 2) * 4, stream));
  if (rec.cells > 3 * rec.nf) {
    p->n_dblocks = (rec.cells + 1 + kDirBlock - 1) / kDirBlock;
    PCK(p->dir_blocks.alloc(p->n_dblocks * 4, stream));
    PCK(suffix_min_u32(nullptr, p->tmp_suffix, p->dir_blocks.as<uint32_t>(), p->n_dblocks, stream));
  }
  PCK(p->tmp.alloc(std::max(p->tmp_sort, p->tmp_suffix), stream));
  PCK(p->fb_list.alloc(rec.nf * 4, stream));
  p->all_boxes = knn_box_count(rec.shape, rec.g);
  if (rec.use_list) {
    PCK(p->box_list.alloc((size_t)p->all_boxes * 4, stream));
    p->list_cap = (uint32_t)std::min<uint64_t>(p->all_boxes, (uint64_t)rec.n_list + rec.n_list / 4 + 1024);  // a quarter more occupied boxes than recorded
  }
  p->fb_cap = (uint32_t)std::min<uint64_t>(rec.nf, (uint64_t)rec.n_fb + rec.n_fb / 2 + 65536);  // half as many more hand-backs than recorded
  PCK(p->unres.alloc(n, stream));
  PCK(p->open_lists.alloc((size_t)2 * kReplayOpenCap * 4, stream));
#undef PCK
  return p.release();
}
void knn_plan_free(KnnPlan* p) { delete p; }
bool knn_plan_accepts(const KnnPlan* p, const uint8_t* pos_base, uint64_t pos_stride) {
  return (pos_stride == 24 && ((uintptr_t)pos_base & 7u) == 0) || p->xyz_own.p != nullptr;
}
const KnnPlanRecord& knn_plan_record(const KnnPlan* p) { return p->rec; }

bool run_normals_replay(KnnPlan* p, const uint8_t* pos_base, uint64_t pos_stride, double* out_normals_dev, double* out_curv_dev, uint32_t* out_knn_u32_dev,
                        uint64_t normal_attr, uint64_t normal_stride, uint64_t curv_attr, uint64_t curv_stride, unsigned long long* status2, hipStream_t stream) {
#define RCK(x) do { if ((x) != hipSuccess) return false; } while (0)
  const KnnPlanRecord& r = p->rec;
  const uint64_t n = r.n, nf = r.nf, cells = r.cells;
  const uint32_t k = r.k;
  const GridParams& g = r.g;
  const unsigned cus = (unsigned)device_cus(), sgrid = p->sgrid;
  RCK(hipMemsetAsync(p->counters.p, 0, 128, stream));
  // packed or not is a property of THIS call's buffer, not of the one the plan was made on (another cloud of the same length may sit in a
  // VectorBuffer): a plan made on a packed column has no staging copy and takes packed sources only (knn_plan_accepts)
  const bool packed_now = pos_stride == 24 && ((uintptr_t)pos_base & 7u) == 0;
  if (!packed_now && !p->xyz_own.p) return false;
  const double* src = (const double*)pos_base;
  if (!packed_now) {
    hipLaunchKernelGGL(gather_positions_kernel, dim3(sgrid), dim3(kBlock), 0, stream, pos_base, pos_stride, n, p->xyz_own.as<double>(), p->partials.as<double>());
    src = p->xyz_own.as<double>();
  }
  unsigned long long* n_finite = (unsigned long long*)p->counters.p;
  uint32_t* fb_count = (uint32_t*)((uint8_t*)p->counters.p + 16);
  int* error_count = (int*)((uint8_t*)p->counters.p + 32);
  uint32_t* unres_count = (uint32_t*)((uint8_t*)p->counters.p + 64);
  uint32_t* box_count = (uint32_t*)((uint8_t*)p->counters.p + 76);
  // index: keys (+ the sort's first histogram), sort, permutation, directory -- what build_index does for a dense grid, with the recorded grid
  {
    RadixFirstPass walk{nullptr, (uint32_t)((n + 8191) / 8192), 0, 8192};
    const RadixFirstPass first = sort_first_pass(p->tmp.p, n, r.key_bits);
    if (first.counts) walk = first;
    hipLaunchKernelGGL(keys_kernel<uint32_t>, dim3(std::max(1u, std::min(walk.tiles, cus * 16u))), dim3(kBlock), 0, stream, src, n, g, p->keys.as<uint32_t>(), (uint32_t*)nullptr,
                       n_finite, walk);
    size_t tb = p->tmp_sort;
    RCK(sort_pairs_u32(p->tmp.p, tb, p->keys.as<uint32_t>(), p->keys2.as<uint32_t>(), p->idx.as<uint32_t>(), p->idx2.as<uint32_t>(), n, r.key_bits, stream, true, &first));
    const unsigned rgrid = (unsigned)std::max<uint64_t>(1, (n + (uint64_t)kBlock * 2 - 1) / ((uint64_t)kBlock * 2));
    hipLaunchKernelGGL(reorder_kernel<2>, dim3(rgrid), dim3(kBlock), 0, stream, src, p->idx2.as<uint32_t>(), n, p->sorted_xyz.as<double>());
    if (cells > 3 * nf) {
      RCK(hipMemsetAsync(p->dir_blocks.p, 0xFF, p->n_dblocks * 4, stream));
      hipLaunchKernelGGL(dir_block_heads_kernel, dim3(sgrid), dim3(kBlock), 0, stream, p->keys2.as<uint32_t>(), nf, cells, p->dir_blocks.as<uint32_t>());
      size_t sb = p->tmp_suffix;
      RCK(suffix_min_u32(p->tmp.p, sb, p->dir_blocks.as<uint32_t>(), p->n_dblocks, stream));
      hipLaunchKernelGGL(dir_fill_kernel, dim3((unsigned)((p->n_dblocks + kDirPerGroup - 1) / kDirPerGroup)), dim3(kBlock), 0, stream, p->keys2.as<uint32_t>(), nf, cells,
                         (const uint32_t*)p->dir_blocks.as<uint32_t>(), p->n_dblocks, p->directory.as<uint32_t>());
    } else {
      hipLaunchKernelGGL(build_directory_kernel, dim3(sgrid), dim3(kBlock), 0, stream, p->keys2.as<uint32_t>(), nf, cells, p->directory.as<uint32_t>());
    }
  }
  const uint32_t* cell_start = p->directory.as<uint32_t>();
  const uint32_t* list_ptr = nullptr;
  if (r.use_list) {
    if (!knn_box_list_async(r.shape, cell_start, g, p->box_list.as<uint32_t>(), box_count, stream)) return false;
    list_ptr = p->box_list.as<uint32_t>();
  }
  RCK(hipMemsetAsync(p->unres.p, 0, n, stream));
  RecOut sorted{nullptr, p->idx2.as<uint32_t>(), nullptr, out_knn_u32_dev, error_count, out_normals_dev, out_curv_dev, normal_attr, normal_stride, curv_attr, curv_stride};
  launch_knn_tile(r.shape, p->sorted_xyz.as<double>(), cell_start, g, k, (uint32_t)nf, sorted, p->fb_list.as<uint32_t>(), fb_count, list_ptr, p->list_cap, stream,
                  list_ptr ? box_count : nullptr);
  // what the box kernel handed back: the exact search over the dense directory, its length read on the device (the list is not sorted by
  // position here: that sort takes its length on the host)
  {
    const unsigned grid = (unsigned)((p->fb_cap + kBlock - 1) / kBlock);
    CellTable table{nullptr, nullptr, 0};
#define RDISPATCH(K) hipLaunchKernelGGL((knn_grid_kernel<K, true, true>), dim3(grid), dim3(kBlock), 0, stream, (const double*)p->sorted_xyz.as<double>(), (const uint64_t*)nullptr, \
                                        (uint32_t)nf, k, g, table, cell_start, (const uint32_t*)p->fb_list.as<uint32_t>(), p->fb_cap, sorted, kShellCap, p->unres.as<uint8_t>(), \
                                        unres_count, 0u, p->open_lists.as<uint32_t>(), kReplayOpenCap, (const uint32_t*)fb_count)
    if (k <= 8) RDISPATCH(8); else if (k <= 16) RDISPATCH(16); else if (k <= 32) RDISPATCH(32); else RDISPATCH(64);
#undef RDISPATCH
  }
  hipLaunchKernelGGL(knn_replay_status_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)n_finite, (unsigned long long)nf, (const uint32_t*)fb_count, p->fb_cap,
                     r.use_list ? (const uint32_t*)box_count : (const uint32_t*)nullptr, p->list_cap, (const uint32_t*)unres_count, (const int*)error_count, status2);
  return hipGetLastError() == hipSuccess;
#undef RCK
}

}  // namespace pstk
