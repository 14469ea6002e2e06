// Voxel-grid down-sampling for gfx950: voxelgrid_filter, pasture-algorithms/src/voxel_grid.rs:109-689.
//
// The reference keeps a Vec<Voxel> sorted by (x, y, z) leaf position with binary-search inserts (O(n * voxels)) and then
// reduces every attribute of every voxel with per-point view lookups and String-keyed HashMaps.  Here:
//
//   voxel_keys_kernel    per point: find_leaf (:21-52) = lower_bound over the axis markers + "better fitting marker" step;
//                        key = x | y | z packed x-major in as few bits as the marker counts need (fewer radix passes)
//   radix sort           (key, point index) pairs (radix_sort.hip: 32- and 64-bit keys); stable, so points keep ascending index order inside a voxel
//   voxel_heads_*        run heads of the sorted keys: per-tile counts, exclusive scan, voxel starts (two light passes over the keys)
//   voxel_reduce_kernel  a wave per 64 voxels: averages / max-pools of small voxels one voxel per LANE (sequential loop = the
//                        reference's loop); most-common attributes and large voxels one voxel per WAVE (set_all_attributes :459-689):
//                          average:     64 values fetched in parallel, then ONE sequential f64 addition chain in point order
//                                       (v_readlane + v_add_f64) so the sums round exactly like the reference's loop
//                          max-pool:    per-lane strict '>' from 0.0, wave max (order-independent, NaN never wins)
//                          most common: <= 64 points: match-any by 16 ballots; <= kMidVoxel points: readlane compare loops;
//                                       larger voxels: voxel_mode_big_kernel (one block per voxel, 65536-bin histogram)
//                                       ties -> smallest value (the reference's HashMap iteration order is random)
// Gather-bound (points of a voxel are scattered in the source) + sort-bound: no MFMA.
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <vector>

#include "device_common.hpp"
#include "device_sort.hpp"
#include "kernels.hpp"

using namespace pstd;

namespace {

constexpr uint32_t kMidVoxel = 2048;   // up to here the wave counts matches itself
constexpr uint32_t kBigBlocks = 128;   // blocks (and 256 KiB histograms) of the big-voxel pass
constexpr int kMaxVoxelAttrs = 24;

struct VoxelAttr {
  uint64_t src, dst;               // attribute of source point 0 / of target point 0
  uint32_t src_stride, dst_stride;
  uint32_t reduce;                 // pstk::VX_*
  uint32_t kind;                   // datatype kind of the attribute (PST_U8 ...)
};
// Stream-ordered form (pst_voxelgrid_filter_async): what the host of the synchronous call reads back between kernels stays in device memory.
// Written by voxel_plan_markers_kernel / voxel_count_kernel, read by the key, head and reduction kernels of the same call.
struct VoxelDyn {
  double origin[3];               // the cloud's minimum (find_leaf's arithmetic guess starts from it)
  uint32_t n_markers[3];          // markers per axis; the three arrays are consecutive in the marker buffer
  uint32_t status;                // pstk::VX_STATUS_* bits; 0 = the results are the reference's
  unsigned long long n_voxels;    // occupied voxels (clamped to the plan's capacity when VX_STATUS_VOXEL_CAPACITY is set)
};

struct VoxelArgs {
  const VoxelDyn* dyn;                   // null: n_voxels below is the host's count
  const uint32_t* sorted_idx;            // point indices sorted by voxel key
  const unsigned long long* starts;      // [n_voxels + 1] offsets into sorted_idx
  uint64_t n_voxels;
  uint64_t dst_first;                    // first new point of the target buffer
  uint32_t* big_list;                    // voxels with more than kMidVoxel points ...
  unsigned int* big_count;               // ... and how many
  uint32_t n_attrs;
  VoxelAttr attrs[kMaxVoxelAttrs];
};

// find_leaf :21-52 for one axis.  markers[0..n) strictly increasing, markers[i] ~ origin + (i + 1) * leaf (accumulated sums, so
// the arithmetic guess is only a starting point; the two fix-up loops make the result exactly the reference's linear scan).
__device__ __forceinline__ uint32_t find_leaf_axis(double p, const double* __restrict__ markers, uint32_t n, double origin, double inv_leaf) {
  if (n == 0) return 0;
  const double t = (p - origin) * inv_leaf;
  uint32_t index = t >= 1.0 ? (t < (double)(n - 1) ? (uint32_t)t : n - 1) : 0u;  // NaN -> 0
  // The guess is off by at most one in all but freak cases (the markers are accumulated sums), so the three markers around it are read
  // TOGETHER -- one LDS round trip instead of a dependent load per step -- and decide: lb = first index with !(markers[i] < p), never past
  // the last marker (it is >= max >= p); NaN -> 0.  Anything the three cannot decide walks the array as before.
  const double a = index > 0 ? markers[index - 1] : 0.0, b = markers[index], c = index + 1 < n ? markers[index + 1] : 0.0;
  double lo, hi;  // markers[lb - 1], markers[lb]
  bool have = false;
  if (index == 0 || a < p) {
    if (!(b < p) || index + 1 == n) { lo = a; hi = b; have = true; }                                     // lb = index
    else if (index + 2 == n || !(c < p)) { index += 1; lo = b; hi = c; have = true; }                    // lb = index + 1
  }
  if (!have) {
    while (index + 1 < n && markers[index] < p) index += 1;
    while (index > 0 && !(markers[index - 1] < p)) index -= 1;
    hi = markers[index];
    lo = index > 0 ? markers[index - 1] : 0.0;
  }
  if (index > 0 && p - lo < hi - p) index -= 1;  // "clamp values to the better fitting marker"
  return index;
}

struct AxisGrid { const double* markers; uint32_t n; uint32_t shift; double origin, inv_leaf; };

// LDS_MARKERS: the three marker arrays are staged in LDS first (they are walked twice per point and axis: the arithmetic guess is
// usually off by at most one, but every step is a dependent load) -- dynamic LDS of exactly their size, so that a grid of a few hundred markers
// leaves the CU to eight workgroups; grids with more markers than kLdsMarkers read them from global memory.
// The points are walked in the radix sort's tiles (first.tile_size consecutive points per workgroup and step) and the digit histogram of the
// sort's first pass is counted on the way (radix_sort.hip: counts[digit][tile]) -- the sort then starts with its scatter.  idx == nullptr: the
// sort numbers the points itself.
constexpr uint32_t kLdsMarkers = 6144;  // 48 KiB
template <typename KeyT, bool LDS_MARKERS>
__global__ __launch_bounds__(kBlock) void voxel_keys_kernel(const uint8_t* __restrict__ pos_base, uint64_t pos_stride, uint64_t n, AxisGrid gx, AxisGrid gy,
                                                            AxisGrid gz, KeyT* __restrict__ keys, uint32_t* __restrict__ idx, pstk::RadixFirstPass first,
                                                            const VoxelDyn* __restrict__ dyn) {
  extern __shared__ double lds_markers[];
  __shared__ uint32_t hist[512];
  if (dyn) {  // stream-ordered form: marker counts and the origin were worked out on the device (voxel_plan_markers_kernel)
    gx.n = dyn->n_markers[0]; gy.n = dyn->n_markers[1]; gz.n = dyn->n_markers[2];
    gx.origin = dyn->origin[0]; gy.origin = dyn->origin[1]; gz.origin = dyn->origin[2];
    gy.markers = gx.markers + gx.n; gz.markers = gy.markers + gy.n;
  }
  if constexpr (LDS_MARKERS) {
    // gx.markers, gy.markers, gz.markers are consecutive in one device array (voxel_grid_build)
    const uint32_t total = gx.n + gy.n + gz.n;
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) lds_markers[i] = gx.markers[i];
    __syncthreads();
    gx.markers = lds_markers; gy.markers = lds_markers + gx.n; gz.markers = lds_markers + gx.n + gy.n;
  }
  const uint32_t mask = first.counts ? (1u << first.bits) - 1u : 0u, steps = first.tile_size / kBlock;
  for (uint64_t tile = blockIdx.x; tile < first.tiles; tile += gridDim.x) {
    if (first.counts) {
      for (uint32_t d = threadIdx.x; d <= mask; d += kBlock) hist[d] = 0;
      __syncthreads();
    }
    constexpr uint32_t U = 4;  // points per thread in flight (tile_size is a multiple of U * kBlock)
    for (uint32_t it = 0; it < steps; it += U) {
      const uint64_t i0 = tile * first.tile_size + (uint64_t)it * kBlock + threadIdx.x;
      if (i0 >= n) break;
      double px[U], py[U], pz[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kBlock < n ? i0 + (uint64_t)u * kBlock : i0;
        cgptr_t p = (cgptr_t)((uint64_t)(uintptr_t)pos_base + i * pos_stride);
        px[u] = load_un<double>(p); py[u] = load_un<double>(p + 8); pz[u] = load_un<double>(p + 16);
      }
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint64_t i = i0 + (uint64_t)u * kBlock;
        if (i >= n) break;
        const uint64_t kx = find_leaf_axis(px[u], gx.markers, gx.n, gx.origin, gx.inv_leaf), ky = find_leaf_axis(py[u], gy.markers, gy.n, gy.origin, gy.inv_leaf),
                       kz = find_leaf_axis(pz[u], gz.markers, gz.n, gz.origin, gz.inv_leaf);
        const KeyT key = (KeyT)((kx << gx.shift) | (ky << gy.shift) | kz);  // x-major: integer order == the reference's (x, y, z) tuple order
        keys[i] = key;
        if (idx) idx[i] = (uint32_t)i;
        if (first.counts) atomicAdd(&hist[(uint32_t)key & mask], 1u);
      }
    }
    if (first.counts) {
      __syncthreads();
      for (uint32_t d = threadIdx.x; d <= mask; d += kBlock) first.counts[(uint64_t)d * first.tiles + tile] = hist[d];
      __syncthreads();
    }
  }
}

// ---- voxel segmentation: run heads of the sorted keys -------------------------------------------------------------------------------
// A tile = kBlock * kHeadsPerThread consecutive sorted keys.  Pass 1 counts the heads of every tile; an exclusive scan of the counts
// gives every tile the rank of its first voxel; pass 2 writes starts[rank] = position for every head, and starts[n_voxels] = n.
constexpr int kHeadsPerThread = 8;
template <typename KeyT, bool WRITE>
__global__ __launch_bounds__(kBlock) void voxel_heads_kernel(const KeyT* __restrict__ keys, uint64_t n, uint32_t* __restrict__ tile_counts,
                                                             const unsigned long long* __restrict__ tile_first, unsigned long long* __restrict__ starts,
                                                             unsigned long long starts_cap) {  // WRITE: starts[] holds starts_cap + 1 entries
  __shared__ uint32_t wave_sums[kBlock / 64];
  const uint64_t tile0 = (uint64_t)blockIdx.x * (kBlock * kHeadsPerThread);
  const uint64_t j0 = tile0 + (uint64_t)threadIdx.x * kHeadsPerThread;
  uint32_t flags = 0, cnt = 0;
  KeyT prev = j0 > 0 && j0 <= n ? keys[j0 - 1] : KeyT(0);
  KeyT kk[kHeadsPerThread];
  if (j0 + kHeadsPerThread <= n) {  // the thread's eight keys as 16-byte vector loads (32 / 64 contiguous bytes per lane)
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v* v = reinterpret_cast<const u32x4v*>(keys + j0);
    constexpr int NV = (int)(sizeof(KeyT) * kHeadsPerThread / 16);
    u32x4v w[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) w[q] = v[q];
    __builtin_memcpy(kk, w, sizeof(kk));
  } else {
#pragma unroll
    for (int u = 0; u < kHeadsPerThread; ++u) kk[u] = j0 + u < n ? keys[j0 + u] : KeyT(0);
  }
#pragma unroll
  for (int u = 0; u < kHeadsPerThread; ++u) {
    const uint64_t j = j0 + u;
    if (j < n) {
      const KeyT k = kk[u];
      if (j == 0 || k != prev) { flags |= 1u << u; cnt += 1; }
      prev = k;
    }
  }
  // block exclusive scan of cnt
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) { if ((uint32_t)w < wave) before += wave_sums[w]; total += wave_sums[w]; }
  if constexpr (!WRITE) {
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
  } else {
    unsigned long long rank = tile_first[blockIdx.x] + before + (inc - cnt);
#pragma unroll
    for (int u = 0; u < kHeadsPerThread; ++u)
      if (flags & (1u << u)) { if (rank < starts_cap) starts[rank] = j0 + u; ++rank; }
    // (more voxels than the stream-ordered plan provided for: the count kernel flags it; starts[cap] then ends the last voxel kept)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { const unsigned long long e = tile_first[blockIdx.x] + total; starts[e < starts_cap ? e : starts_cap] = n; }
  }
}

__device__ __forceinline__ double wave_bcast_f64(double v, uint32_t lane) {  // lane is wave-uniform
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, (int)lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)lane);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

// component `c` of the attribute of point `i` as f64 (`as f64` of voxel_grid.rs:178-209, :344-376, :401-431)
__device__ __forceinline__ double load_as_f64(const VoxelAttr& a, uint32_t scalar_kind, uint64_t i, uint32_t c) {
  cgptr_t p = (cgptr_t)as_global(a.src) + i * a.src_stride;
  switch (scalar_kind) {
    case CT_U8: return (double)load_un<uint8_t>(p + c);
    case CT_I8: return (double)load_un<int8_t>(p + c);
    case CT_U16: return (double)load_un<uint16_t>(p + 2 * c);
    case CT_I16: return (double)load_un<int16_t>(p + 2 * c);
    case CT_U32: return (double)load_un<uint32_t>(p + 4 * c);
    case CT_I32: return (double)load_un<int32_t>(p + 4 * c);
    case CT_U64: return (double)load_un<uint64_t>(p + 8 * c);
    case CT_I64: return (double)load_un<int64_t>(p + 8 * c);
    case CT_F32: return (double)load_un<float>(p + 4 * c);
    default: return load_un<double>(p + 8 * c);
  }
}
__device__ __forceinline__ int32_t load_as_int(const VoxelAttr& a, uint64_t i) {  // most-common attributes: u8 / i8 / u16 / i16
  cgptr_t p = (cgptr_t)as_global(a.src) + i * a.src_stride;
  switch (a.kind) {
    case 0: return load_un<uint8_t>(p);
    case 1: return load_un<int8_t>(p);
    case 2: return load_un<uint16_t>(p);
    default: return load_un<int16_t>(p);
  }
}
// scalar component type of a datatype kind (PST_* order: U8 I8 U16 I16 U32 I32 U64 I64 F32 F64 Vec3u8 Vec3u16 Vec3f32 Vec3i32 Vec3f64)
__device__ __forceinline__ uint32_t scalar_of(uint32_t kind) {
  switch (kind) {
    case 10: return CT_U8;
    case 11: return CT_U16;
    case 12: return CT_F32;
    case 13: return CT_I32;
    case 14: return CT_F64;
    default: return kind;
  }
}

__device__ __forceinline__ void store_mode(const VoxelAttr& a, uint64_t out_point, int32_t value) {
  gptr_t d = as_global(a.dst) + out_point * a.dst_stride;
  if (a.reduce == pstk::VX_MOST_COMMON_BOOL) store_un<uint8_t>(d, (uint8_t)(value != 0));  // `!= 0` then bytes_of(bool) :548, :563
  else if (a.kind <= 1) store_un<uint8_t>(d, (uint8_t)value);
  else store_un<uint16_t>(d, (uint16_t)value);
}

// (count, value) -> key whose maximum is: highest count, then smallest value
__device__ __forceinline__ uint64_t mode_key(uint32_t count, int32_t value) { return ((uint64_t)count << 32) | (uint32_t)(0x7FFFFFFF - (value + 32768)); }
__device__ __forceinline__ int32_t mode_value(uint64_t key) { return (int32_t)(0x7FFFFFFF - (uint32_t)key) - 32768; }
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint64_t o = shfl_xor_any(v, off);
    v = o > v ? o : v;
  }
  return v;
}

constexpr uint32_t kSmallVoxel = 48;  // averages / max-pools of voxels up to this size run one voxel per LANE

// averages and max-pools of ONE voxel computed by ONE lane: a sequential loop over the voxel's points — exactly the reference's
// loop (same f64 addition order) — with 64 independent voxels per wave hiding the gather latency.
__device__ __forceinline__ void reduce_voxel_by_lane(const VoxelAttr& at, const uint32_t* __restrict__ idx, uint32_t m, uint64_t out) {
  gptr_t d = as_global(at.dst) + out * at.dst_stride;
  const uint32_t sk = scalar_of(at.kind);
  if (at.reduce == pstk::VX_MAX_POOL) {
    double cur = 0.0;
    for (uint32_t j = 0; j < m; ++j) {
      const double x = load_as_f64(at, sk, idx[j], 0);
      if (x > cur) cur = x;
    }
    if (at.kind == 9) store_un<double>(d, cur);
    else if (at.kind == 6) store_un<uint64_t>(d, rust_as<uint64_t, double>(cur));
    else store_un<uint8_t>(d, rust_as<uint8_t, double>(cur));
    return;
  }
  const double np = (double)m;
  if (at.reduce == pstk::VX_AVG_NUM) {
    double sum = 0.0;
    for (uint32_t j = 0; j < m; ++j) sum += load_as_f64(at, sk, idx[j], 0);
    store_un<uint16_t>(d, rust_as<uint16_t, double>(sum / np));
    return;
  }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (uint32_t j = 0; j < m; ++j) {
    const uint64_t i = idx[j];
    s0 += load_as_f64(at, sk, i, 0); s1 += load_as_f64(at, sk, i, 1); s2 += load_as_f64(at, sk, i, 2);
  }
  const double a0 = s0 / np, a1 = s1 / np, a2 = s2 / np;
  if (at.kind == 14) { store_un<double>(d, a0); store_un<double>(d + 8, a1); store_un<double>(d + 16, a2); }
  else if (at.kind == 11) {
    store_un<uint16_t>(d, rust_as<uint16_t, double>(a0)); store_un<uint16_t>(d + 2, rust_as<uint16_t, double>(a1));
    store_un<uint16_t>(d + 4, rust_as<uint16_t, double>(a2));
  } else { store_un<float>(d, (float)a0); store_un<float>(d + 4, (float)a1); store_un<float>(d + 8, (float)a2); }
}

// The 64 voxels of group `grp`, by the waves of one workgroup WITHOUT staging: every wave computes the voxels' sizes, wave 0 takes the small
// voxels one per lane, and the waves share the voxels that need a whole wave (most-common attributes; everything of large voxels) round robin.
// `skip_avg`: the averages and max-pools of this group have been reduced from LDS already (voxel_reduce_kernel below).
__device__ __forceinline__ void reduce_group_by_waves(const VoxelArgs& a, uint64_t n_voxels, uint64_t grp, uint32_t lane, uint32_t wave, uint32_t n_waves, bool any_mode, bool skip_avg) {
  {
    // ---- phase 1: lane l owns voxel 64*grp + l; averages and max-pools of small voxels ----
    const uint64_t lv = grp * 64 + lane;
    uint32_t lm = 0;
    uint64_t ls = 0;
    if (lv < n_voxels) { ls = a.starts[lv]; lm = (uint32_t)(a.starts[lv + 1] - ls); }
    const bool small = lm != 0 && (lm <= kSmallVoxel || skip_avg);
    if (small && !skip_avg && wave == 0) {
      for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
        const VoxelAttr& at = a.attrs[ai];
        if (at.reduce == pstk::VX_AVG_VEC || at.reduce == pstk::VX_AVG_NUM || at.reduce == pstk::VX_MAX_POOL)
          reduce_voxel_by_lane(at, a.sorted_idx + ls, lm, a.dst_first + lv);
      }
    }
    // ---- phase 2: the waves walk the voxels that still need one: most-common attributes, and everything of large voxels ----
    uint64_t todo = __ballot(lm != 0 && (any_mode || !small));
    uint32_t turn = 0;
    while (todo) {
      const uint32_t vl = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1;
      if (turn++ % n_waves != wave) continue;
      const uint64_t v = grp * 64 + vl;
      const uint64_t s = a.starts[v];
      const uint32_t m = (uint32_t)(a.starts[v + 1] - s);  // n < 2^32
      const bool by_lane_done = m <= kSmallVoxel || skip_avg;
      const uint32_t* idx = a.sorted_idx + s;
      const uint64_t out = a.dst_first + v;
      if (m > kMidVoxel && lane == 0) a.big_list[atomicAdd(a.big_count, 1u)] = (uint32_t)v;
      for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
        const VoxelAttr& at = a.attrs[ai];
        if (at.reduce == pstk::VX_AVG_VEC || at.reduce == pstk::VX_AVG_NUM) {
          if (by_lane_done) continue;
          const uint32_t nc = at.reduce == pstk::VX_AVG_VEC ? 3u : 1u, sk = scalar_of(at.kind);
          double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0;
          for (uint32_t b = 0; b < m; b += 64) {
            const uint32_t cnt = m - b < 64u ? m - b : 64u;
            double x0 = 0.0, x1 = 0.0, x2 = 0.0;
            if (lane < cnt) {
              const uint64_t i = idx[b + lane];
              x0 = load_as_f64(at, sk, i, 0);
              if (nc == 3) { x1 = load_as_f64(at, sk, i, 1); x2 = load_as_f64(at, sk, i, 2); }
            }
            // the reference's loop: x_sum += v.x; ... one point after the other (:343-376, :400-431)
            if (nc == 3) {
              for (uint32_t j = 0; j < cnt; ++j) { sum0 += wave_bcast_f64(x0, j); sum1 += wave_bcast_f64(x1, j); sum2 += wave_bcast_f64(x2, j); }
            } else {
              for (uint32_t j = 0; j < cnt; ++j) sum0 += wave_bcast_f64(x0, j);
            }
          }
          if (lane == 0) {
            const double np = (double)m;
            gptr_t d = as_global(at.dst) + out * at.dst_stride;
            const double a0 = sum0 / np, a1 = sum1 / np, a2 = sum2 / np;
            if (at.reduce == pstk::VX_AVG_NUM) {
              store_un<uint16_t>(d, rust_as<uint16_t, double>(a0));  // `as u16` :489, :638
            } else if (at.kind == 14) {
              store_un<double>(d, a0); store_un<double>(d + 8, a1); store_un<double>(d + 16, a2);
            } else if (at.kind == 11) {  // ColorRGB :626
              store_un<uint16_t>(d, rust_as<uint16_t, double>(a0)); store_un<uint16_t>(d + 2, rust_as<uint16_t, double>(a1));
              store_un<uint16_t>(d + 4, rust_as<uint16_t, double>(a2));
            } else {  // Normal :676
              store_un<float>(d, (float)a0); store_un<float>(d + 4, (float)a1); store_un<float>(d + 8, (float)a2);
            }
          }
        } else if (at.reduce == pstk::VX_MAX_POOL) {
          if (by_lane_done) continue;
          const uint32_t sk = scalar_of(at.kind);
          double cur = 0.0;  // :175
          for (uint32_t j = lane; j < m; j += 64) {
            const double x = load_as_f64(at, sk, idx[j], 0);
            if (x > cur) cur = x;
          }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const double o = shfl_xor_any(cur, off);
            if (o > cur) cur = o;
          }
          if (lane == 0) {
            gptr_t d = as_global(at.dst) + out * at.dst_stride;
            if (at.kind == 9) store_un<double>(d, cur);
            else if (at.kind == 6) store_un<uint64_t>(d, rust_as<uint64_t, double>(cur));
            else store_un<uint8_t>(d, rust_as<uint8_t, double>(cur));
          }
        } else {  // most common
          if (m > kMidVoxel) continue;  // voxel_mode_big_kernel
          uint64_t best = 0;
          if (m <= 64) {
            const bool valid = lane < m;
            const int32_t x = valid ? load_as_int(at, idx[lane]) : 0;
            const uint64_t vmask = __ballot(valid);
            uint64_t same = vmask;
            const uint32_t ux = (uint32_t)(x + 32768);
#pragma unroll
            for (int bit = 0; bit < 17; ++bit) {
              const bool b1 = (ux >> bit) & 1u;
              const uint64_t bal = __ballot(b1);
              same &= b1 ? bal : ~bal;
            }
            if (valid) best = mode_key((uint32_t)__builtin_popcountll(same & vmask), x);
          } else {
            for (uint32_t ca = 0; ca < m; ca += 64) {  // candidates
              const bool valid = ca + lane < m;
              const int32_t x = valid ? load_as_int(at, idx[ca + lane]) : 0;
              uint32_t count = 0;
              for (uint32_t cb = 0; cb < m; cb += 64) {
                const uint32_t cnt = m - cb < 64u ? m - cb : 64u;
                const int32_t w = cb + lane < m ? load_as_int(at, idx[cb + lane]) : 0;
                for (uint32_t j = 0; j < cnt; ++j) count += (x == __builtin_amdgcn_readlane(w, (int)j)) ? 1u : 0u;
              }
              if (valid) { const uint64_t k = mode_key(count, x); best = k > best ? k : best; }
            }
          }
          best = wave_max_u64(best);
          if (lane == 0) store_mode(at, out, mode_value(best));
        }
      }
    }
  }
}

// One workgroup per group of 64 consecutive voxels.  Averages and max-pools are reduced from LDS: the group's points (consecutive in the sorted
// order: starts[64 grp] .. starts[64 grp + 64)) are fetched by ALL lanes of the workgroup, one point per lane and step with the index list read
// lane-contiguously -- as many independent random reads in flight as a plain permutation of the points has -- and stored by component; then lane l
// of wave c adds component c of voxel l's points in their sorted (= original) order, the reference's `x_sum += v.x` loop (:343-376, :400-431).
// The one-voxel-per-lane loop over global memory that this replaces kept one dependent index -> point chain per lane (3.3 ms per 10^8 points
// against 2.4 ms for a permutation).  Groups with more than `cap` points (dense voxels) go the unstaged way, shared by the four waves.
__global__ __launch_bounds__(kBlock) void voxel_reduce_kernel(const VoxelArgs a, const uint32_t cap) {
  extern __shared__ double staged[];  // [3][cap]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t grp = blockIdx.x;
  bool any_mode = false, any_avg = false;
  for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
    const uint32_t r = a.attrs[ai].reduce;
    any_mode = any_mode || r == pstk::VX_MOST_COMMON || r == pstk::VX_MOST_COMMON_BOOL;
    any_avg = any_avg || r == pstk::VX_AVG_VEC || r == pstk::VX_AVG_NUM || r == pstk::VX_MAX_POOL;
  }
  const uint64_t n_voxels = a.dyn ? a.dyn->n_voxels : a.n_voxels;
  if (grp * 64 >= n_voxels) return;  // (stream-ordered form: the grid is the plan's capacity)
  const uint64_t v0 = grp * 64, v1 = v0 + 64 < n_voxels ? v0 + 64 : n_voxels;
  const uint64_t p0 = a.starts[v0];
  const uint64_t cnt64 = a.starts[v1] - p0;
  const bool stage = any_avg && cnt64 <= (uint64_t)cap;
  if (stage) {
    const uint32_t cnt = (uint32_t)cnt64;
    const uint64_t lv = v0 + lane;
    uint32_t lm = 0, lo = 0;  // lane l: voxel v0 + l has lm points, the first is staged at `lo`
    if (lv < n_voxels) { const uint64_t ls = a.starts[lv]; lm = (uint32_t)(a.starts[lv + 1] - ls); lo = (uint32_t)(ls - p0); }
    const uint32_t* __restrict__ idx = a.sorted_idx + p0;
    for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
      const VoxelAttr& at = a.attrs[ai];
      if (!(at.reduce == pstk::VX_AVG_VEC || at.reduce == pstk::VX_AVG_NUM || at.reduce == pstk::VX_MAX_POOL)) continue;
      const uint32_t nc = at.reduce == pstk::VX_AVG_VEC ? 3u : 1u, sk = scalar_of(at.kind);
      // ---- fetch: four points per lane in flight --------------------------------------------------------------------------------------
      constexpr int U = 4;
      for (uint32_t j0 = tid; j0 < cnt; j0 += kBlock * U) {
        uint32_t i[U];
        double x0[U], x1[U], x2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t j = j0 + (uint32_t)u * kBlock; i[u] = j < cnt ? idx[j] : idx[j0]; }
        if (nc == 3 && sk == CT_F64) {
          typedef double d2u __attribute__((ext_vector_type(2), aligned(1)));
#pragma unroll
          for (int u = 0; u < U; ++u) {
            cgptr_t p = (cgptr_t)as_global(at.src) + (uint64_t)i[u] * at.src_stride;
            const d2u xy = *reinterpret_cast<const PST_AS_GLOBAL d2u*>(p);
            x0[u] = xy.x; x1[u] = xy.y; x2[u] = load_un<double>(p + 16);
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            x0[u] = load_as_f64(at, sk, i[u], 0);
            if (nc == 3) { x1[u] = load_as_f64(at, sk, i[u], 1); x2[u] = load_as_f64(at, sk, i[u], 2); }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t j = j0 + (uint32_t)u * kBlock;
          if (j < cnt) {
            staged[j] = x0[u];
            if (nc == 3) { staged[cap + j] = x1[u]; staged[2 * cap + j] = x2[u]; }
          }
        }
      }
      __syncthreads();
      // ---- reduce: wave c, lane l = component c of voxel v0 + l --------------------------------------------------------------------------
      if (wave < nc && lm != 0) {
        const double* __restrict__ col = staged + wave * cap + lo;
        gptr_t d = as_global(at.dst) + (a.dst_first + lv) * at.dst_stride;
        if (at.reduce == pstk::VX_MAX_POOL) {
          double cur = 0.0;  // :175
          for (uint32_t j = 0; j < lm; ++j) { const double x = col[j]; if (x > cur) cur = x; }
          if (at.kind == 9) store_un<double>(d, cur);
          else if (at.kind == 6) store_un<uint64_t>(d, rust_as<uint64_t, double>(cur));
          else store_un<uint8_t>(d, rust_as<uint8_t, double>(cur));
        } else {
          double sum = 0.0;
          for (uint32_t j = 0; j < lm; ++j) sum += col[j];
          const double avg = sum / (double)lm;
          if (at.reduce == pstk::VX_AVG_NUM) store_un<uint16_t>(d, rust_as<uint16_t, double>(avg));  // `as u16` :489, :638
          else if (at.kind == 14) store_un<double>(d + 8 * wave, avg);
          else if (at.kind == 11) store_un<uint16_t>(d + 2 * wave, rust_as<uint16_t, double>(avg));  // ColorRGB :626
          else store_un<float>(d + 4 * wave, (float)avg);                                              // Normal :676
        }
      }
      __syncthreads();
    }
    if (!any_mode) return;
  }
  reduce_group_by_waves(a, n_voxels, grp, lane, wave, kBlock / 64, any_mode, stage);
}

// One block per voxel with more than kMidVoxel points: 65536-bin histogram (ascending bins are ascending values) in the block's private global scratch.
__global__ __launch_bounds__(kBlock) void voxel_mode_big_kernel(const VoxelArgs a, uint32_t* __restrict__ hist_all) {
  uint32_t* hist = hist_all + (size_t)blockIdx.x * 65536u;
  __shared__ unsigned long long best_s[kBlock / 64];
  const uint32_t n_big = *a.big_count;
  for (uint32_t bi = blockIdx.x; bi < n_big; bi += gridDim.x) {
    const uint64_t v = a.big_list[bi];
    const uint64_t s = a.starts[v];
    const uint32_t m = (uint32_t)(a.starts[v + 1] - s);
    const uint32_t* idx = a.sorted_idx + s;
    for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
      const VoxelAttr& at = a.attrs[ai];
      if (at.reduce != pstk::VX_MOST_COMMON && at.reduce != pstk::VX_MOST_COMMON_BOOL) continue;
      for (uint32_t b = threadIdx.x; b < 65536u; b += kBlock) hist[b] = 0;
      __syncthreads();
      const int32_t bias = (at.kind == 1 || at.kind == 3) ? 32768 : 0;  // signed kinds: bin = value + 32768
      for (uint32_t j = threadIdx.x; j < m; j += kBlock) atomicAdd(&hist[(uint32_t)(load_as_int(at, idx[j]) + bias)], 1u);
      __syncthreads();
      uint64_t best = 0;
      for (uint32_t b = threadIdx.x; b < 65536u; b += kBlock) {
        const uint32_t c = hist[b];
        if (c) { const uint64_t k = mode_key(c, (int32_t)b - bias); best = k > best ? k : best; }
      }
      best = wave_max_u64(best);
      if ((threadIdx.x & 63u) == 0) best_s[threadIdx.x >> 6] = best;
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) best = best_s[w] > best ? best_s[w] : best;
        store_mode(at, a.dst_first + v, mode_value(best));
      }
      __syncthreads();
    }
  }
}


// Stream-ordered form, step 1: what voxelgrid_filter's host code does between calculate_bounds and the key kernel -- AABB::from_min_max's
// check (math/bounds.rs:21-26) and create_markers_for_axis (voxel_grid.rs:55-83: curr = min; while curr < max { curr += leaf; push }), the
// same sequential f64 additions, by ONE lane: a few hundred to a few thousand dependent adds (microseconds).  The three marker arrays are
// written back to back.  Overflowing the plan's capacities (total markers, key bits per axis) sets a status bit; the arrays are cut there.
__global__ void voxel_plan_markers_kernel(const double* __restrict__ bounds6, double lx, double ly, double lz, double* __restrict__ markers, uint32_t cap_markers,
                                          uint32_t bits_x, uint32_t bits_y, uint32_t bits_z, VoxelDyn* __restrict__ dyn) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t status = 0, total = 0;
  const double leaf[3] = {lx, ly, lz};
  const uint32_t bits[3] = {bits_x, bits_y, bits_z};
  if (bounds6[0] > bounds6[3] || bounds6[1] > bounds6[4] || bounds6[2] > bounds6[5]) status |= pstk::VX_STATUS_BOUNDS_INVALID;
  for (int a = 0; a < 3; ++a) {
    const double mn = bounds6[a], mx = bounds6[3 + a];
    dyn->origin[a] = mn;
    uint32_t cnt = 0;
    double curr = mn;
    while (!status && curr < mx) {
      const double next = curr + leaf[a];
      if (!(next > curr)) { status |= pstk::VX_STATUS_LEAF; break; }
      if (total >= cap_markers || cnt >= (1u << bits[a])) { status |= pstk::VX_STATUS_MARKER_CAPACITY; break; }
      curr = next;
      markers[total++] = curr;
      cnt += 1;
    }
    dyn->n_markers[a] = cnt;
  }
  dyn->status = status;
  dyn->n_voxels = 0;
}
// step 2, behind the scan of the run heads: the voxel count stays on the device
__global__ void voxel_count_kernel(const unsigned long long* __restrict__ total_runs, unsigned long long cap_voxels, VoxelDyn* __restrict__ dyn,
                                   unsigned long long* __restrict__ count_and_status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long runs = dyn->status ? 0ull : *total_runs;
  uint32_t status = dyn->status;
  if (runs > cap_voxels) { status |= pstk::VX_STATUS_VOXEL_CAPACITY; runs = cap_voxels; }
  dyn->n_voxels = runs;
  dyn->status = status;
  if (count_and_status) { count_and_status[0] = runs; count_and_status[1] = status; }
}

}  // namespace

namespace pstk {

struct VoxelGridState {
  DevBuf keys, keys2, idx, idx2, tmp, markers, unique, counts, starts, big_list, big_count, hist, dyn;
  uint64_t n = 0, n_voxels = 0;
  // stream-ordered plan (voxel_plan_create): every buffer above is allocated once for these capacities and no call allocates
  bool planned = false;
  VoxelPlanShape shape{};
  size_t tmp_sort = 0, tmp_scan = 0;
};

// Phase 1: keys, sort, voxel segmentation.  Returns the number of voxels (>= 1), or -1 on a HIP failure.
// KeyT = uint32_t when the packed (x, y, z) key fits 32 bits (half the key traffic of the radix sort), else uint64_t.
template <typename KeyT>
static long long voxel_grid_build_typed(VoxelGridState* st, const uint8_t* pos_base, uint64_t pos_stride, uint64_t n, AxisGrid gx, AxisGrid gy, AxisGrid gz,
                                        unsigned end_bit, hipStream_t stream, unsigned long long* count_and_status = nullptr) {
#define VCK(x) do { if ((x) != hipSuccess) return -1; } while (0)
  // planned (stream-ordered form): the buffers exist (voxel_plan_create), nothing is allocated and nothing is read back; the voxel count
  // stays in st->dyn and the run-head table is capped at the plan's capacity
  const bool planned = st->planned;
  const VoxelDyn* dyn = planned ? st->dyn.as<VoxelDyn>() : nullptr;
  const uint64_t tiles = (n + (uint64_t)kBlock * kHeadsPerThread - 1) / ((uint64_t)kBlock * kHeadsPerThread);
  size_t tmp_sort = st->tmp_sort, tmp_scan = st->tmp_scan;
  constexpr bool own_keys = sizeof(KeyT) == 4;  // (32-bit keys: the library's own sort numbers the points and takes its first histogram from the key kernel)
  RadixFirstPass first{nullptr, 0, 0, 0};
  auto sort = [&](void* tmp, size_t& bytes) {
    if constexpr (own_keys)
      return sort_pairs_u32(tmp, bytes, st->keys.as<uint32_t>(), st->keys2.as<uint32_t>(), st->idx.as<uint32_t>(), st->idx2.as<uint32_t>(), n, end_bit, stream, true, &first);
    else
      return sort_pairs_u64(tmp, bytes, st->keys.as<uint64_t>(), st->keys2.as<uint64_t>(), st->idx.as<uint32_t>(), st->idx2.as<uint32_t>(), n, end_bit, stream);
  };
  if (!planned) {
    VCK(st->keys.alloc(n * sizeof(KeyT), stream)); VCK(st->keys2.alloc(n * sizeof(KeyT), stream));
    VCK(st->idx.alloc(n * 4, stream)); VCK(st->idx2.alloc(n * 4, stream));
    tmp_sort = tmp_scan = 0;
    VCK(sort(nullptr, tmp_sort));
    VCK(st->counts.alloc((tiles + 1) * 4, stream));
    VCK(st->unique.alloc((tiles + 1) * 8, stream));  // exclusive scan of the tile counts (+ the total)
    VCK(exclusive_sum_u32_u64(nullptr, tmp_scan, st->counts.as<uint32_t>(), st->unique.as<unsigned long long>(), tiles + 1, stream));
    VCK(st->tmp.alloc(std::max(tmp_sort, tmp_scan), stream));
    st->tmp_sort = tmp_sort; st->tmp_scan = tmp_scan;
  }
  if constexpr (own_keys) first = sort_first_pass(st->tmp.p, n, end_bit);
  RadixFirstPass walk = first;  // (no histogram wanted: the same walk, nothing counted)
  if (!walk.counts) { walk.tile_size = 8192; walk.tiles = (uint32_t)((n + 8191) / 8192); walk.bits = 0; }
  uint32_t* idx_out = own_keys ? nullptr : st->idx.as<uint32_t>();
  const unsigned grid = (unsigned)std::max<uint32_t>(1, walk.tiles);
  const uint32_t n_markers = planned ? st->shape.cap_markers : gx.n + gy.n + gz.n;  // (planned: LDS for the plan's capacity)
  if (n_markers <= kLdsMarkers)
    hipLaunchKernelGGL((voxel_keys_kernel<KeyT, true>), dim3(grid), dim3(kBlock), (size_t)n_markers * 8, stream, pos_base, pos_stride, n, gx, gy, gz, st->keys.as<KeyT>(),
                       idx_out, walk, dyn);
  else
    hipLaunchKernelGGL((voxel_keys_kernel<KeyT, false>), dim3(grid), dim3(kBlock), 0, stream, pos_base, pos_stride, n, gx, gy, gz, st->keys.as<KeyT>(), idx_out, walk, dyn);
  VCK(sort(st->tmp.p, tmp_sort));
  VCK(hipMemsetAsync(st->counts.as<uint32_t>() + tiles, 0, 4, stream));
  hipLaunchKernelGGL((voxel_heads_kernel<KeyT, false>), dim3((unsigned)tiles), dim3(kBlock), 0, stream, (const KeyT*)st->keys2.as<KeyT>(), n,
                     st->counts.as<uint32_t>(), (const unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull);
  VCK(exclusive_sum_u32_u64(st->tmp.p, tmp_scan, st->counts.as<uint32_t>(), st->unique.as<unsigned long long>(), tiles + 1, stream));
  unsigned long long runs = 0, starts_cap = ~0ull;
  if (planned) {
    starts_cap = st->shape.cap_voxels;
    hipLaunchKernelGGL(voxel_count_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)(st->unique.as<unsigned long long>() + tiles), starts_cap,
                       st->dyn.as<VoxelDyn>(), count_and_status);
    st->n_voxels = starts_cap;
  } else {
    VCK(hipMemcpyAsync(&runs, st->unique.as<unsigned long long>() + tiles, 8, hipMemcpyDeviceToHost, stream));
    VCK(hipStreamSynchronize(stream));
    st->n_voxels = runs;
    VCK(st->starts.alloc((runs + 1) * 8, stream));
  }
  hipLaunchKernelGGL((voxel_heads_kernel<KeyT, true>), dim3((unsigned)tiles), dim3(kBlock), 0, stream, (const KeyT*)st->keys2.as<KeyT>(), n,
                     (uint32_t*)nullptr, (const unsigned long long*)st->unique.as<unsigned long long>(), st->starts.as<unsigned long long>(), starts_cap);
  return planned ? (long long)starts_cap : (long long)runs;
#undef VCK
}

long long voxel_grid_build(VoxelGridState*& st, const uint8_t* pos_base, uint64_t pos_stride, uint64_t n, const double* markers_x, uint32_t nx,
                           const double* markers_y, uint32_t ny, const double* markers_z, uint32_t nz, const double origin[3], const double leaf[3],
                           hipStream_t stream) {
  st = new VoxelGridState();
  st->n = n;
  if (st->markers.alloc(((size_t)nx + ny + nz + 1) * 8, stream) != hipSuccess) return -1;
  double* dm = st->markers.as<double>();
  if (nx && hipMemcpyAsync(dm, markers_x, (size_t)nx * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
  if (ny && hipMemcpyAsync(dm + nx, markers_y, (size_t)ny * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
  if (nz && hipMemcpyAsync(dm + nx + ny, markers_z, (size_t)nz * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
  auto bits_for = [](uint32_t count) { uint32_t b = 1; while (b < 21 && (1u << b) < count) ++b; return b; };  // indices 0 .. count-1
  const uint32_t bz = bits_for(nz), by = bits_for(ny), bx = bits_for(nx);
  const unsigned end_bit = bx + by + bz;
  AxisGrid gx{dm, nx, by + bz, origin[0], 1.0 / leaf[0]}, gy{dm + nx, ny, bz, origin[1], 1.0 / leaf[1]}, gz{dm + nx + ny, nz, 0, origin[2], 1.0 / leaf[2]};
  return end_bit <= 32 ? voxel_grid_build_typed<uint32_t>(st, pos_base, pos_stride, n, gx, gy, gz, end_bit, stream)
                       : voxel_grid_build_typed<uint64_t>(st, pos_base, pos_stride, n, gx, gy, gz, end_bit, stream);
}

// How many points of dynamic LDS (3 doubles each) voxel_reduce_kernel may stage on the CURRENT device: the opt-in is per device (the kernel's
// attribute is set once for each), and a device that offers less than the 144 KiB the largest groups use gets a smaller cap -- 0 (the
// unstaged path) when even the smallest staging does not fit.  gfx950: 160 KiB per workgroup => 6144 points.
static uint32_t voxel_stage_limit() {
  static uint32_t per_device[64];
  static bool known[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!known[dev]) {
    // the full 144 KiB first (gfx950 grants it); a device that refuses gets what its per-block limit allows, or nothing
    uint32_t pts = 6144;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(voxel_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pts * 24)) != hipSuccess) {
      (void)hipGetLastError();
      int limit = 0;
      if (hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); limit = 0; }
      pts = (uint32_t)std::max<long long>(0, ((long long)limit - 2048) / 24) & ~63u;  // (2 KiB of static LDS in the kernel)
      if (pts > 6144) pts = 6144;
      if (pts < 1024 || hipFuncSetAttribute(reinterpret_cast<const void*>(voxel_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pts * 24)) != hipSuccess) {
        (void)hipGetLastError();
        pts = 0;
      }
    }
    per_device[dev] = pts;
    known[dev] = true;
  }
  return per_device[dev];
}

// Phase 2: reductions into target points [dst_first, dst_first + n_voxels).
bool voxel_grid_reduce(VoxelGridState* st, const uint64_t* src_addr, const uint32_t* src_stride, const uint64_t* dst_addr, const uint32_t* dst_stride,
                       const uint32_t* reduce, const uint32_t* kind, int n_attrs, uint64_t dst_first, hipStream_t stream) {
  if (n_attrs > kMaxVoxelAttrs) return false;
  VoxelArgs a{};
  a.dyn = st->planned ? st->dyn.as<VoxelDyn>() : nullptr;
  a.sorted_idx = st->idx2.as<uint32_t>();
  a.starts = st->starts.as<unsigned long long>();
  a.n_voxels = st->n_voxels;  // (planned: the plan's capacity = the launch grid; the kernels read the count from a.dyn)
  a.dst_first = dst_first;
  const size_t max_big = (size_t)(st->n / kMidVoxel) + 1;
  if (!st->planned && (st->big_list.alloc(max_big * 4, stream) != hipSuccess || st->big_count.alloc(16, stream) != hipSuccess)) return false;
  if (hipMemsetAsync(st->big_count.p, 0, 16, stream) != hipSuccess) return false;
  a.big_list = st->big_list.as<uint32_t>();
  a.big_count = st->big_count.as<unsigned int>();
  a.n_attrs = (uint32_t)n_attrs;
  bool any_mode = false;
  for (int i = 0; i < n_attrs; ++i) {
    a.attrs[i] = VoxelAttr{src_addr[i], dst_addr[i], src_stride[i], dst_stride[i], reduce[i], kind[i]};
    any_mode = any_mode || reduce[i] == VX_MOST_COMMON || reduce[i] == VX_MOST_COMMON_BOOL;
  }
  // LDS for the points of 64 voxels: 1.6 x the average group (a Poisson-filled grid of 15 points per voxel: 960 +- 31 per group), between
  // 1024 and 6144 points (24 .. 144 KiB: four .. one workgroups per CU); groups beyond it take the unstaged path
  const uint64_t groups = (st->n_voxels + 63) / 64;
  if (groups == 0) return hipGetLastError() == hipSuccess;
  if (groups > 0x7FFFFFFFull) return false;
  static const uint32_t cap_env = [] { const char* e = std::getenv("PST_VOXEL_STAGE"); return e && *e ? (uint32_t)std::strtoul(e, nullptr, 10) : ~0u; }();  // 0 = never stage (A/B)
  uint32_t cap = (uint32_t)std::min<uint64_t>(6144, std::max<uint64_t>(1024, (st->n * 8 / 5) / groups + 63)) & ~63u;
  if (cap_env != ~0u) cap = std::min<uint32_t>(cap_env, 6144) & ~63u;
  if (st->planned) cap = st->shape.stage_cap;  // (sized for the voxel count the plan measured, not for its capacity)
  cap = std::min(cap, voxel_stage_limit());  // (0: this device cannot stage -- every group takes the unstaged path)
  const size_t lds = (size_t)std::max<uint32_t>(cap, 64) * 3 * sizeof(double);
  hipLaunchKernelGGL(voxel_reduce_kernel, dim3((unsigned)groups), dim3(kBlock), lds, stream, a, cap);
  if (any_mode && st->n > kMidVoxel && st->planned) {
    // stream-ordered: the big-voxel pass always runs on its fixed grid; its blocks read the count on the device and most find nothing
    hipLaunchKernelGGL(voxel_mode_big_kernel, dim3(kBigBlocks), dim3(kBlock), 0, stream, a, st->hist.as<uint32_t>());
  } else if (any_mode && st->n > kMidVoxel) {
    unsigned int n_big = 0;
    if (hipMemcpyAsync(&n_big, st->big_count.p, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
    if (hipStreamSynchronize(stream) != hipSuccess) return false;
    if (n_big) {
      const unsigned bgrid = std::min<unsigned>(n_big, kBigBlocks);
      if (st->hist.alloc((size_t)bgrid * 65536u * 4u, stream) != hipSuccess) return false;
      hipLaunchKernelGGL(voxel_mode_big_kernel, dim3(bgrid), dim3(kBlock), 0, stream, a, st->hist.as<uint32_t>());
    }
  }
  return hipGetLastError() == hipSuccess;
}

// ---- stream-ordered form -------------------------------------------------------------------------------------------------------------
// voxel_plan_create allocates every buffer of a call for the capacities in `shape` (chosen by voxel_api.cpp from ONE synchronous run over
// the cloud); voxel_grid_build_async then enqueues markers -> keys -> sort -> run heads -> count without a host round trip or an
// allocation (hipGraph-capturable), and voxel_grid_reduce works off the device-side count.
VoxelGridState* voxel_plan_create(const VoxelPlanShape& shape, hipStream_t stream) {
  auto st = std::make_unique<VoxelGridState>();
  st->n = shape.n;
  st->shape = shape;
  const uint64_t n = shape.n;
  const unsigned end_bit = shape.bits[0] + shape.bits[1] + shape.bits[2];
  const size_t key_bytes = end_bit <= 32 ? 4 : 8;
  const uint64_t tiles = (n + (uint64_t)kBlock * kHeadsPerThread - 1) / ((uint64_t)kBlock * kHeadsPerThread);
#define PCK(x) do { if ((x) != hipSuccess) return nullptr; } while (0)
  PCK(st->markers.alloc(((size_t)shape.cap_markers + 1) * 8, stream));
  PCK(st->dyn.alloc(sizeof(VoxelDyn), stream));
  PCK(st->keys.alloc(n * key_bytes, stream)); PCK(st->keys2.alloc(n * key_bytes, stream));
  PCK(st->idx.alloc(n * 4, stream)); PCK(st->idx2.alloc(n * 4, stream));
  size_t tmp_sort = 0, tmp_scan = 0;
  RadixFirstPass none{nullptr, 0, 0, 0};
  if (key_bytes == 4) PCK(sort_pairs_u32(nullptr, tmp_sort, st->keys.as<uint32_t>(), st->keys2.as<uint32_t>(), st->idx.as<uint32_t>(), st->idx2.as<uint32_t>(), n, end_bit, stream, true, &none));
  else PCK(sort_pairs_u64(nullptr, tmp_sort, st->keys.as<uint64_t>(), st->keys2.as<uint64_t>(), st->idx.as<uint32_t>(), st->idx2.as<uint32_t>(), n, end_bit, stream));
  PCK(st->counts.alloc((tiles + 1) * 4, stream));
  PCK(st->unique.alloc((tiles + 1) * 8, stream));
  PCK(exclusive_sum_u32_u64(nullptr, tmp_scan, st->counts.as<uint32_t>(), st->unique.as<unsigned long long>(), tiles + 1, stream));
  PCK(st->tmp.alloc(std::max(tmp_sort, tmp_scan), stream));
  st->tmp_sort = tmp_sort; st->tmp_scan = tmp_scan;
  PCK(st->starts.alloc((shape.cap_voxels + 1) * 8, stream));
  PCK(st->big_list.alloc(((size_t)(n / kMidVoxel) + 1) * 4, stream));
  PCK(st->big_count.alloc(16, stream));
  if (n > kMidVoxel) PCK(st->hist.alloc((size_t)kBigBlocks * 65536u * 4u, stream));
  (void)voxel_stage_limit();  // (the per-device opt-in happens here, not inside a stream-ordered call)
#undef PCK
  st->planned = true;
  return st.release();
}

// bounds6: the cloud's {min xyz, max xyz} in device memory (calculate_bounds on the same stream).  count_and_status: two device words.
bool voxel_grid_build_async(VoxelGridState* st, const uint8_t* pos_base, uint64_t pos_stride, const double* bounds6, unsigned long long* count_and_status,
                            hipStream_t stream) {
  if (!st || !st->planned) return false;
  const VoxelPlanShape& sh = st->shape;
  double* dm = st->markers.as<double>();
  hipLaunchKernelGGL(voxel_plan_markers_kernel, dim3(1), dim3(64), 0, stream, bounds6, sh.leaf[0], sh.leaf[1], sh.leaf[2], dm, sh.cap_markers, sh.bits[0], sh.bits[1],
                     sh.bits[2], st->dyn.as<VoxelDyn>());
  const uint32_t bx = sh.bits[0], by = sh.bits[1], bz = sh.bits[2];
  const unsigned end_bit = bx + by + bz;
  // (marker counts, origins and the y / z array positions come from st->dyn inside the key kernel)
  AxisGrid gx{dm, 0, by + bz, 0.0, 1.0 / sh.leaf[0]}, gy{dm, 0, bz, 0.0, 1.0 / sh.leaf[1]}, gz{dm, 0, 0, 0.0, 1.0 / sh.leaf[2]};
  const long long r = end_bit <= 32 ? voxel_grid_build_typed<uint32_t>(st, pos_base, pos_stride, sh.n, gx, gy, gz, end_bit, stream, count_and_status)
                                    : voxel_grid_build_typed<uint64_t>(st, pos_base, pos_stride, sh.n, gx, gy, gz, end_bit, stream, count_and_status);
  return r >= 0 && hipGetLastError() == hipSuccess;
}

void voxel_grid_free(VoxelGridState* st) { delete st; }

}  // namespace pstk
