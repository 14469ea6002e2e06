// K1 / K2 fast path — columnar Vec3f64 stream: copy | affine | AABB in ONE pass over HBM (gfx950).
//
// Replaces, for a columnar POSITION_3D (Vec3f64) attribute:
//   * calculate_bounds_from_default_positions           pasture-algorithms/src/bounds.rs:30-54          (24 B/pt read)
//   * convert_columnar_to_columnar same-type arm        buffer_conversion.rs:461-485 (memcpy + in-place
//     transformation sweep = 2 passes in the reference)                                               (24 R + 24 W)
//   * transform_attribute with the LAS affine closure   point_buffer.rs:391-404, raw_readers.rs:42-48   (24 R + 24 W)
// and any combination of them (convert + transform + bounds of the result = 48 B/pt, the AABB rides for free).
//
// Memory mapping.  A Vec3f64 column is a flat f64 stream x0 y0 z0 x1 ...  Each lane moves 16-byte double2 vectors
// (1 KiB per wave instruction, fully coalesced).  A block owns tiles of 6 x 256 vectors; lane t's j-th load of every
// tile is vector  base + j*256 + t  with base = 0 (mod 3 vectors), so the xyz phase of (t, j) is LOOP-INVARIANT:
// it is computed once, the per-phase scale/offset live in registers, and the running min/max are kept per
// (j mod 3, half) — no per-element modulo, no divergence.  The phase is folded back to x/y/z once, after the loop.
// Six independent 16-byte loads per lane are in flight before the first use.
//
// Reduction: wave64 shuffle -> LDS across the 4 waves -> one {min xyz, max xyz} record per block -> a one-block
// finalize kernel.  Seeds are +/-f64::MAX like the reference; fmin/fmax never let a NaN win, exactly like the
// reference's strict `<` / `>` compares.  HBM-bound: no MFMA, no LDS staging needed.
//
// Since round 5 the kernel in THIS file serves the read-only modes (AABB, AABB of the transformed values) as a persistent grid; every mode that
// writes runs the one-tile-per-block body of stream_tile.hpp (rotated accumulators, LDS block fold, few bytes in flight per CU).
#include "device_common.hpp"
#include "kernels.hpp"
#include "stream_tile.hpp"

#include <algorithm>
#include <mutex>
#include <vector>
#include <cstdlib>

using namespace pstd;

namespace pstk {
// Device AABB records the library has been told to leave as {min, -max} (pst_bounds_record_set_form): what ONE ncclAllReduce(ncclMin) reduces.  The last
// fold kernel of whichever path writes such a record negates its three maxima as it stores them, so the exposed exchange of the sharded path is the
// collective alone (round-5 review: two one-block negation launches around a 48-byte all-reduce).  A handful of addresses per process (an exchange ring).
static std::mutex g_form_mu;
static std::vector<const void*> g_negated_records;
void set_bounds_record_form(const void* device_rec6, int form) {
  std::lock_guard<std::mutex> lock(g_form_mu);
  auto it = std::find(g_negated_records.begin(), g_negated_records.end(), device_rec6);
  if (form && it == g_negated_records.end()) g_negated_records.push_back(device_rec6);
  if (!form && it != g_negated_records.end()) g_negated_records.erase(it);
}
bool bounds_record_negates_max(const void* device_rec6) {
  std::lock_guard<std::mutex> lock(g_form_mu);
  return !g_negated_records.empty() && std::find(g_negated_records.begin(), g_negated_records.end(), device_rec6) != g_negated_records.end();
}
}  // namespace pstk
using pstk::bounds_record_negates_max;

namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));

// KLOADS = 16-byte loads in flight per lane (multiple of 3 => a tile is a whole number of xyz periods => phase
// invariance); NTL / NTS = non-temporal loads / stores (streaming data is touched once).
template <bool AFFINE, bool WRITE, bool BOUNDS, int KLOADS, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void vec3f64_stream_kernel(const StreamParams p) {
  static_assert(KLOADS % 3 == 0, "tile must be a multiple of 3 vectors per lane");
  constexpr int kLoads = KLOADS;
  constexpr int kTileVec = kLoads * kBlock;
  const uint32_t t = threadIdx.x;
  const PST_AS_GLOBAL f64x2* __restrict__ src = (const PST_AS_GLOBAL f64x2*)(p.src + p.vec_first);
  PST_AS_GLOBAL f64x2* __restrict__ dst = (PST_AS_GLOBAL f64x2*)(p.dst + p.vec_first);

  // component of half h of lane t's j-th vector: (vec_first + 2*(j*256 + t) + h) mod 3, invariant over tiles
  double sc[3][2], of[3][2];
  uint32_t comp[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint32_t c = (uint32_t)((p.vec_first + 2ull * (uint64_t)(j * kBlock + t) + hh) % 3ull);
      comp[j][hh] = c;
      sc[j][hh] = pick3(c, p.scale[0], p.scale[1], p.scale[2]);
      of[j][hh] = pick3(c, p.offset[0], p.offset[1], p.offset[2]);
    }
  }
  double mn[3][2], mx[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    mn[j][0] = mn[j][1] = kF64Max;
    mx[j][0] = mx[j][1] = -kF64Max;
  }

  auto body = [&](f64x2 v, int j) __attribute__((always_inline)) -> f64x2 {
    const int jj = j % 3;
    double a = v.x, b = v.y;
    if constexpr (AFFINE) {
#pragma clang fp contract(off)
      a = a * sc[jj][0];
      a = a + of[jj][0];
      b = b * sc[jj][1];
      b = b + of[jj][1];
    }
    if constexpr (BOUNDS) {
      mn[jj][0] = __builtin_fmin(mn[jj][0], a);
      mx[jj][0] = __builtin_fmax(mx[jj][0], a);
      mn[jj][1] = __builtin_fmin(mn[jj][1], b);
      mx[jj][1] = __builtin_fmax(mx[jj][1], b);
    }
    f64x2 r;
    r.x = a;
    r.y = b;
    return r;
  };

  const uint64_t n_tiles = (p.n_vec + kTileVec - 1) / kTileVec;
  // XCD-aware tile numbering (one tile per block): consecutive workgroup ids land on different XCDs, so without it every XCD
  // touches every eighth tile of the stream; with it each XCD streams one contiguous eighth of the range.
  uint64_t tile0 = blockIdx.x;
  if (p.xcd_chunk) {
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    if (p.xcd_block) tile0 = ((uint64_t)(k / p.xcd_block) * 8u + xcd) * p.xcd_block + k % p.xcd_block;  // runs of xcd_block tiles, round-robin
    else tile0 = (uint64_t)xcd * p.xcd_chunk + k;
    if (tile0 >= n_tiles) tile0 = n_tiles;  // the surplus blocks still write their (identity) partial record
  }
  for (uint64_t tile = tile0; tile < n_tiles; tile += gridDim.x) {
    const uint64_t base = tile * kTileVec + t;
    if ((tile + 1) * (uint64_t)kTileVec <= p.n_vec) {
      f64x2 v[kLoads];
#pragma unroll
      for (int j = 0; j < kLoads; ++j) {
        if constexpr (NTL) v[j] = __builtin_nontemporal_load(&src[base + (uint64_t)j * kBlock]);
        else v[j] = src[base + (uint64_t)j * kBlock];
      }
#pragma unroll
      for (int j = 0; j < kLoads; ++j) {
        const f64x2 r = body(v[j], j);
        if constexpr (WRITE) {
          if constexpr (NTS) __builtin_nontemporal_store(r, &dst[base + (uint64_t)j * kBlock]);
          else dst[base + (uint64_t)j * kBlock] = r;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < kLoads; ++j) {
        const uint64_t i = base + (uint64_t)j * kBlock;
        if (i < p.n_vec) {
          const f64x2 r = body(src[i], j);
          if constexpr (WRITE) dst[i] = r;
        }
      }
    }
  }

  // fold the (phase, half) accumulators back to x / y / z
  double bmn[3] = {kF64Max, kF64Max, kF64Max}, bmx[3] = {-kF64Max, -kF64Max, -kF64Max};
  if constexpr (BOUNDS) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
          const bool hit = comp[j][hh] == c;
          bmn[c] = __builtin_fmin(bmn[c], hit ? mn[j][hh] : kF64Max);
          bmx[c] = __builtin_fmax(bmx[c], hit ? mx[j][hh] : -kF64Max);
        }
      }
    }
  }

  // ragged doubles outside the 16-byte aligned vector body (at most one in front, two behind): block 0, lanes 0..2
  if (blockIdx.x == 0 && t < 3) {
    const uint64_t tail_first = p.vec_first + 2 * p.n_vec;
    uint64_t idx = ~0ull;
    if (t == 0 && p.vec_first == 1) idx = 0;
    if (t >= 1 && tail_first + (t - 1) < p.n_doubles) idx = tail_first + (t - 1);
    if (idx != ~0ull) {
      const uint32_t c = (uint32_t)(idx % 3ull);
      double a = p.src[idx];
      if constexpr (AFFINE) {
#pragma clang fp contract(off)
        a = a * pick3(c, p.scale[0], p.scale[1], p.scale[2]);
        a = a + pick3(c, p.offset[0], p.offset[1], p.offset[2]);
      }
      if constexpr (WRITE) p.dst[idx] = a;
      if constexpr (BOUNDS) {
#pragma unroll
        for (uint32_t cc = 0; cc < 3; ++cc) {
          bmn[cc] = __builtin_fmin(bmn[cc], cc == c ? a : kF64Max);
          bmx[cc] = __builtin_fmax(bmx[cc], cc == c ? a : -kF64Max);
        }
      }
    }
  }

  if constexpr (BOUNDS) {
    __shared__ double scratch[(kBlock / 64) * 6];
    block_reduce_minmax<double, 3>(bmn, bmx, scratch);
    if (t == 0) {
      double* out = p.partials + (uint64_t)blockIdx.x * 6;
      out[0] = bmn[0]; out[1] = bmn[1]; out[2] = bmn[2];
      out[3] = bmx[0]; out[4] = bmx[1]; out[5] = bmx[2];
    }
  }
}

// Folds per-block records: block b reduces records [b*chunk, (b+1)*chunk) into out[b] = {min[NV], max[NV]}.
// Launched with one block (chunk >= n_records) for the final fold, or as a first level when there are many records.
template <typename T, int NV>
__global__ __launch_bounds__(kBlock) void finalize_minmax_kernel(const T* __restrict__ partials, uint32_t n_records, uint32_t chunk,
                                                                 T* __restrict__ out, T seed_min, T seed_max, uint32_t negate_max = 0) {
  T mn[NV], mx[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { mn[i] = seed_min; mx[i] = seed_max; }
  const uint64_t first = (uint64_t)blockIdx.x * chunk;
  const uint64_t last = first + chunk < n_records ? first + chunk : n_records;
  for (uint64_t r = first + threadIdx.x; r < last; r += kBlock) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      mn[i] = fold_min(mn[i], partials[r * 2 * NV + i]);
      mx[i] = fold_max(mx[i], partials[r * 2 * NV + NV + i]);
    }
  }
  __shared__ T scratch[(kBlock / 64) * 2 * NV];
  block_reduce_minmax<T, NV>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    T* o = out + (uint64_t)blockIdx.x * 2 * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) { o[i] = mn[i]; o[NV + i] = negate_max ? (T)(-mx[i]) : mx[i]; }  // (negate_max: the record leaves as {min, -max}, see pst_set_bounds_record_form)
  }
}

// two-level fold when there are many records (one-tile-per-block launches); `partials` must have room for
// n_records + kFoldBlocks records
constexpr uint32_t kFoldBlocks = 128;
template <typename T, int NV>
void launch_finalize(T* partials, uint32_t n_records, T* out, T seed_min, T seed_max, hipStream_t stream, bool negate_max = false) {
  const uint32_t neg = negate_max ? 1u : 0u;  // (the LAST level only)
  if (n_records > 4096) {
    const uint32_t chunk = (n_records + kFoldBlocks - 1) / kFoldBlocks;
    T* level1 = partials + (uint64_t)n_records * 2 * NV;
    hipLaunchKernelGGL((finalize_minmax_kernel<T, NV>), dim3(kFoldBlocks), dim3(kBlock), 0, stream, (const T*)partials, n_records, chunk, level1,
                       seed_min, seed_max, 0u);
    hipLaunchKernelGGL((finalize_minmax_kernel<T, NV>), dim3(1), dim3(kBlock), 0, stream, (const T*)level1, kFoldBlocks, kFoldBlocks, out,
                       seed_min, seed_max, neg);
  } else {
    hipLaunchKernelGGL((finalize_minmax_kernel<T, NV>), dim3(1), dim3(kBlock), 0, stream, (const T*)partials, n_records, n_records, out, seed_min,
                       seed_max, neg);
  }
}

// Generic strided min/max: element e lives at base + e*stride, NCOMP components of T.
// ACC = double: Rust-`as` each component to f64 and fold with +/-f64::MAX seeds (calculate_bounds_from_custom_positions,
// bounds.rs:56-85).  ACC = T: fold in the attribute's own type (minmax_attribute, minmax.rs:13-51).
template <typename T, typename ACC, int NCOMP>
__global__ __launch_bounds__(kBlock) void strided_minmax_kernel(const ReduceParams p, ACC seed_min, ACC seed_max) {
  ACC mn[NCOMP], mx[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { mn[c] = seed_min; mx[c] = seed_max; }
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  cgptr_t base = (cgptr_t)(uint64_t)p.base;
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < p.n; e += step) {
    cgptr_t q = base + e * p.stride;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const ACC v = rust_as<ACC, T>(load_un<T>(q + c * sizeof(T)));
      mn[c] = fold_min(mn[c], v);
      mx[c] = fold_max(mx[c], v);
    }
  }
  __shared__ ACC scratch[(kBlock / 64) * 2 * NCOMP];
  block_reduce_minmax<ACC, NCOMP>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    ACC* out = (ACC*)p.partials + (uint64_t)blockIdx.x * 2 * NCOMP;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) { out[c] = mn[c]; out[NCOMP + c] = mx[c]; }
  }
}

template <typename T> struct Identity {
  static T min_seed() {
    if constexpr (std::is_floating_point<T>::value) return std::numeric_limits<T>::infinity();
    else return std::numeric_limits<T>::max();
  }
  static T max_seed() {
    if constexpr (std::is_floating_point<T>::value) return -std::numeric_limits<T>::infinity();
    else return std::numeric_limits<T>::lowest();
  }
};

template <typename T, int NCOMP>
void launch_minmax_typed(const ReduceParams& p, bool acc_f64, void* out, unsigned grid, hipStream_t stream) {
  if (acc_f64) {
    hipLaunchKernelGGL((strided_minmax_kernel<T, double, NCOMP>), dim3(grid), dim3(kBlock), 0, stream, p, kF64Max, -kF64Max);
    launch_finalize<double, NCOMP>((double*)p.partials, grid, (double*)out, kF64Max, -kF64Max, stream, NCOMP == 3 && bounds_record_negates_max(out));
  } else {
    hipLaunchKernelGGL((strided_minmax_kernel<T, T, NCOMP>), dim3(grid), dim3(kBlock), 0, stream, p, Identity<T>::min_seed(),
                       Identity<T>::max_seed());
    launch_finalize<T, NCOMP>((T*)p.partials, grid, (T*)out, Identity<T>::min_seed(), Identity<T>::max_seed(), stream);
  }
}


// compute_centroid, normal_estimation.rs:198-237: the sums of BOTH branches in one pass -- over all points (is_dense: no coordinate is NaN)
// and over the finite points (the other branch), with the finite count and a "some coordinate is NaN" flag; the host picks the branch.
// Record per block: {ax, ay, az, fx, fy, fz, finite count, NaN seen}.  A parallel sum is not the reference's left-to-right sum: the
// centroid agrees to a few ulps of sum |x| / n (the north star's 1e-9 relative for f64 results), not bit for bit.
__global__ __launch_bounds__(kBlock) void centroid_kernel(const ReduceParams p) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  cgptr_t base = (cgptr_t)(uint64_t)p.base;
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < p.n; e += step) {
    cgptr_t q = base + e * p.stride;
    const double x = load_un<double>(q), y = load_un<double>(q + 8), z = load_un<double>(q + 16);
    a[0] += x; a[1] += y; a[2] += z;
    const bool fin = __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
    if (fin) { a[3] += x; a[4] += y; a[5] += z; a[6] += 1.0; }
    if (x != x || y != y || z != z) a[7] = 1.0;
  }
  __shared__ double scratch[(kBlock / 64) * 8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    double v = a[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    a[c] = v;
  }
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < 8; ++c) scratch[wave * 8 + c] = a[c];
  __syncthreads();
  if (threadIdx.x < 8) {
    double v = 0;
    for (unsigned w = 0; w < kBlock / 64; ++w) v += scratch[w * 8 + threadIdx.x];
    ((double*)p.partials)[(uint64_t)blockIdx.x * 8 + threadIdx.x] = v;
  }
}

}  // namespace

namespace pstk {

// Launch geometry (measured on MI355X; tools/tune_stream*.hip, profiles/r05_stream_sweeps.txt; 10^8 and 10^9 points, random data):
//  * read-only AABB: the round 1-4 kernel as a persistent grid of 4 blocks per CU, 6 loads in flight per lane -> 7.05-7.1 TB/s; the one-tile-
//    per-block body reaches 7.2 but pays it back in the fold of its 10^5 records.
//  * any mode that writes: the round-5 body (stream_tile.hpp), ONE tile per block, and -- the finding of round 5 -- FEW BYTES IN FLIGHT:
//    the rate peaks at about 48 KiB of loads in flight per CU (24 KiB without the reduction tail) and falls off on both sides; the round 1-4
//    launch kept 8 blocks x 6 loads x 256 lanes = 192 KiB per CU in flight.  Same box, 10^8 points, fused convert + affine + AABB:
//    8 blocks/CU x K=6 x 256 lanes 5.93 TB/s | 4 x K=3 x 256 6.47 | 2 x K=3 x 512 6.73 (10^9 points: 6.55 -> 7.03).  The blocks resident per CU
//    are capped through the dynamic LDS size of the launch.
constexpr int kStreamLoads = 6;  // read-only persistent kernel
constexpr int kStreamTileVec = kStreamLoads * kBlock;
int stream_grid() { return device_cus() * 4; }
int reduce_grid() { return device_cus() * 8; }
size_t minmax_partials_bytes() { return (size_t)(reduce_grid() + kFoldBlocks) * 6 * sizeof(double); }

static bool stream_xcd_aware() {
  static const bool on = [] { const char* v = std::getenv("PST_STREAM_XCD"); return !(v && *v == '0'); }();  // on by default: each XCD streams one contiguous eighth
  return on;
}
// shape of the writing modes: loads per lane, threads per block, blocks resident per CU
struct StreamShape { int loads, block, resident; };
static StreamShape stream_shape(unsigned mode) {
  StreamShape s = (mode & 4u) ? StreamShape{3, 512, 2} : StreamShape{3, 256, 2};
  static const int cap = [] { const char* v = std::getenv("PST_STREAM_RESIDENT"); return v && *v ? std::atoi(v) : 0; }();  // same-box A/Bs
  if (cap > 0) s.resident = cap;
  return s;
}
unsigned lds_per_cu();
uint32_t lds_with_resident_cap(size_t lds_bytes, int resident) {
  static const int forced = [] { const char* v = std::getenv("PST_RESIDENT"); return v && *v ? std::atoi(v) : -1; }();
  if (forced >= 0) resident = forced;
  if (resident <= 0 || resident >= 32) return (uint32_t)lds_bytes;
  const size_t want = lds_per_cu() / (unsigned)(resident + 1) + 64u;
  return (uint32_t)std::max(lds_bytes, std::min<size_t>(want, 64u * 1024u));
}
unsigned lds_per_cu() {
  static const unsigned v = [] {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || bytes <= 0) bytes = 64 * 1024;
    return (unsigned)bytes;
  }();
  return v;
}
// dynamic LDS of one block: the reduction rows (modes with bounds), padded so that at most `resident` blocks fit one CU
static unsigned stream_lds_bytes(const StreamShape& s, bool bounds) {
  const unsigned need = bounds ? 6u * (unsigned)(s.block + 8) * 8u : 0u;
  if (s.resident >= 8) return need;
  return std::min(64u * 1024u, std::max(need, lds_per_cu() / (unsigned)(s.resident + 1) + 64u));
}
static uint64_t stream_launch_grid(uint64_t n_points, unsigned mode) {
  const uint64_t n_vec = (3 * n_points) / 2;
  if (mode & 2u) {
    const StreamShape s = stream_shape(mode);
    const uint64_t tile_vec = (uint64_t)s.loads * s.block;
    return 8 * ((std::max<uint64_t>(1, (n_vec + tile_vec - 1) / tile_vec) + 7) / 8);
  }
  const uint64_t n_tiles = std::max<uint64_t>(1, (n_vec + kStreamTileVec - 1) / kStreamTileVec);
  return std::min<uint64_t>(n_tiles, (uint64_t)stream_grid());
}
size_t stream_partials_bytes(uint64_t n_points, unsigned mode) {
  return (size_t)(stream_launch_grid(n_points, mode) + kFoldBlocks) * 6 * sizeof(double);
}

void launch_vec3f64_stream(const double* src, double* dst, uint64_t n_points, const double scale[3], const double offset[3], unsigned mode,
                           double* partials, double* out6, hipStream_t stream) {
  const bool write = mode & 2u, bounds = mode & 4u;
  const unsigned grid = (unsigned)stream_launch_grid(n_points, mode);
  // 16-byte vectors: src and dst must share their alignment phase — the host only takes this path when
  // (src - dst) % 16 == 0 (converter.cpp), so only the phase of src matters.
  const uint64_t n_doubles = 3 * n_points;
  const uint32_t vec_first = (((uintptr_t)src & 15u) != 0 && n_doubles > 0) ? 1 : 0;
  if (write) {
    Stream2Params p{};
    p.src = src;
    p.dst = dst;
    p.n_doubles = n_doubles;
    p.vec_first = vec_first;
    p.n_vec = (n_doubles - vec_first) / 2;
    for (int c = 0; c < 3; ++c) { p.scale[c] = scale ? scale[c] : 1.0; p.offset[c] = offset ? offset[c] : 0.0; }
    p.partials = partials;
    p.xcd_stride = grid / 8u;
    p.plain = stream_xcd_aware() ? 0u : 1u;
    const StreamShape s = stream_shape(mode);
    const unsigned lds = stream_lds_bytes(s, bounds);
#define PST_STREAM2(A, B, K, BLK) hipLaunchKernelGGL((vec3f64_stream2_kernel<A, true, B, K, BLK>), dim3(grid), dim3(BLK), lds, stream, p)
    switch (mode & 7u) {
      case 2: PST_STREAM2(false, false, 3, 256); break;
      case 3: PST_STREAM2(true, false, 3, 256); break;
      case 6: PST_STREAM2(false, true, 3, 512); break;
      case 7: PST_STREAM2(true, true, 3, 512); break;
    }
#undef PST_STREAM2
  } else if (bounds) {
    StreamParams p{};
    p.src = src;
    p.dst = const_cast<double*>(src);
    p.n_doubles = n_doubles;
    p.vec_first = vec_first;
    p.n_vec = (n_doubles - vec_first) / 2;
    for (int c = 0; c < 3; ++c) { p.scale[c] = scale ? scale[c] : 1.0; p.offset[c] = offset ? offset[c] : 0.0; }
    p.partials = partials;
    p.xcd_chunk = 0u;  // the read-only persistent grid loses 1.5 % with the XCD-aware numbering
    p.xcd_block = 0u;
    if (mode & 1u) hipLaunchKernelGGL((vec3f64_stream_kernel<true, false, true, kStreamLoads, true, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    else hipLaunchKernelGGL((vec3f64_stream_kernel<false, false, true, kStreamLoads, true, true>), dim3(grid), dim3(kBlock), 0, stream, p);
  }
  if (bounds) launch_finalize<double, 3>(partials, grid, out6, kF64Max, -kF64Max, stream, bounds_record_negates_max(out6));
}

size_t centroid_partials_bytes() { return (size_t)reduce_grid() * 8 * sizeof(double); }
unsigned launch_centroid(const uint8_t* base, uint64_t stride, uint64_t n, double* partials, hipStream_t stream) {
  ReduceParams p{base, stride, n, partials};
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)reduce_grid()));
  hipLaunchKernelGGL(centroid_kernel, dim3(grid), dim3(kBlock), 0, stream, p);
  return grid;
}

size_t bounds_partials_bytes(unsigned n_records) { return (size_t)(n_records + kFoldBlocks) * 6 * sizeof(double); }
void launch_finalize_bounds(double* partials, unsigned n_records, double* out6, hipStream_t stream) {
  launch_finalize<double, 3>(partials, n_records, out6, kF64Max, -kF64Max, stream, bounds_record_negates_max(out6));
}

void launch_minmax(const uint8_t* base, uint64_t stride, uint64_t n, uint32_t ct, uint32_t ncomp, bool acc_f64, void* partials, void* out,
                   hipStream_t stream) {
  ReduceParams p{base, stride, n, partials};
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)reduce_grid()));
#define PST_MM(T)                                                             \
  do {                                                                        \
    if (ncomp == 3) launch_minmax_typed<T, 3>(p, acc_f64, out, grid, stream); \
    else launch_minmax_typed<T, 1>(p, acc_f64, out, grid, stream);            \
  } while (0)
  switch (ct) {
    case CT_U8: PST_MM(uint8_t); break;
    case CT_I8: PST_MM(int8_t); break;
    case CT_U16: PST_MM(uint16_t); break;
    case CT_I16: PST_MM(int16_t); break;
    case CT_U32: PST_MM(uint32_t); break;
    case CT_I32: PST_MM(int32_t); break;
    case CT_U64: PST_MM(uint64_t); break;
    case CT_I64: PST_MM(int64_t); break;
    case CT_F32: PST_MM(float); break;
    default: PST_MM(double); break;
  }
#undef PST_MM
}

}  // namespace pstk
