// LAS record encoder (SURVEY 8(f) rank 2 — the writer side of the hot path) for gfx950.
//
// Replaces RawLASWriter::write_points_default_layout (pasture-io/src/las/raw_writers.rs:203-363) with its helpers
// write_position_as_las_position / write_las_bit_attributes (write_helpers.rs:10-55) and the header side effects
// update_bounds_in_las_header (raw_writers.rs:28-48) and the points-by-return histogram (:220-229, :256-258, :50-82):
//
//   typed LAS points (LasPointFormatN::layout(), las_types.rs — interleaved or columnar)
//     -> exact-binary LAS point records (las_layout.rs:70-107), interleaved:
//        X,Y,Z = (((p - offset) / scale) as i64) checked into i32   (truncation toward zero; out of range = panic)
//        flags = rn&7 | (nr&7)<<3 | (sd&1)<<6 | (eof&1)<<7   (formats 0-5)  /  two bytes for the extended formats 6-10
//        every other field copied in the record order of the LAS specification.
//
// One lane per point.  The format is a template parameter, so every source attribute slot and record offset is a compile
// time constant (no interpretation).  Source attributes are read where they live (columnar: coalesced; interleaved: the
// wave's lanes cover a contiguous span of records); the record is assembled in an LDS tile (byte-granular ds_writes) and
// leaves with 16-byte coalesced stores.  The header reductions (6-double AABB with the header's current bounds as seeds,
// 15 return counters, out-of-range count) are folded per wave -> per block -> one fold kernel.  HBM-bound; no MFMA.
#include "device_common.hpp"
#include "kernels.hpp"
#include "tile_io.hpp"

#include <algorithm>

using namespace pstd;

namespace {

constexpr int kMaxAttrs = 24;
constexpr int kReturnSlots = 16;  // slot r counts return number r (1..15); slot 0 = out-of-range positions

struct EncodeArgs {
  uint64_t attr_base[kMaxAttrs];    // address of the typed attribute of point 0 (layout slot order of LasPointFormatN)
  uint32_t attr_stride[kMaxAttrs];  // bytes between consecutive points of that attribute
  uint64_t dst;                     // address of raw record 0 of the target range
  uint64_t n;
  double scale[3], offset[3];
  double seed_min[3], seed_max[3];  // the header's current bounds (raw_writers.rs:136-141: f64::MAX / f64::MIN initially)
  double* partial_bounds;           // [grid][6]
  unsigned long long* partial_counts;  // [grid][kReturnSlots]
  uint32_t max_return;              // 5 (legacy header) or 15 (large_file) — raw_writers.rs:221-225
  uint32_t tile;                    // points per tile (multiple of kBlock)
};

struct Fmt { bool ext, gps, color, nir, wave; };
__host__ __device__ constexpr Fmt fmt_of(int n) {
  return Fmt{n >= 6, n == 1 || n == 3 || n == 4 || n == 5 || n >= 6, n == 2 || n == 3 || n == 5 || n == 7 || n == 8 || n == 10, n == 8 || n == 10,
             n == 4 || n == 5 || n == 9 || n == 10};
}
__host__ __device__ constexpr uint32_t raw_size(Fmt f) {
  return (f.ext ? 30u : 20u) + (f.gps && !f.ext ? 8u : 0u) + (f.color ? 6u : 0u) + (f.nir ? 2u : 0u) + (f.wave ? 29u : 0u);
}

template <typename T>
__device__ __forceinline__ T src_load(const EncodeArgs& a, int slot, uint64_t i) {
  return load_un<T>((cgptr_t)(as_global(a.attr_base[slot]) + i * a.attr_stride[slot]));
}

template <int FORMAT>
__global__ __launch_bounds__(kBlock) void las_encode_kernel(const EncodeArgs a) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t RS = raw_size(F);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  __shared__ unsigned int hist[kReturnSlots];
  if (threadIdx.x < kReturnSlots) hist[threadIdx.x] = 0;
  double mn[3] = {a.seed_min[0], a.seed_min[1], a.seed_min[2]}, mx[3] = {a.seed_max[0], a.seed_max[1], a.seed_max[2]};
  unsigned long long counts_acc = 0;  // lane r < 16 accumulates hist[r] across tiles
  __syncthreads();

  const uint64_t n_tiles = (a.n + a.tile - 1) / a.tile;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * a.tile;
    const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
    const uint64_t ga = a.dst + first * RS;
    const uint32_t mis = (uint32_t)(ga & 15u);
    for (uint32_t lp = threadIdx.x; lp < cnt; lp += kBlock) {
      const uint64_t i = first + lp;
      lptr_t rec = lds + (mis + lp * RS);
      int s = 0;  // typed slot cursor (LasPointFormatN field order, las_types.rs)
      uint32_t o = 0;  // raw record cursor
      // position: write_position_as_las_position, write_helpers.rs:10-23
      {
        cgptr_t pp = (cgptr_t)(as_global(a.attr_base[s]) + i * a.attr_stride[s]);
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double w = load_un<double>(pp + 8 * c);
          const double local = (w - a.offset[c]) / a.scale[c];  // two roundings, like the Rust expression
          // `as i64` saturates and maps NaN to 0; try_into::<i32>() then fails outside [i32::MIN, i32::MAX]
          const long long t = rust_as<long long, double>(local);
          if (t > 2147483647ll || t < -2147483648ll) bad = true;
          store_un<int32_t>(rec + o, (int32_t)t);
          o += 4;
          mn[c] = __builtin_fmin(mn[c], w);  // update_bounds_in_las_header: strict compares, NaN never wins
          mx[c] = __builtin_fmax(mx[c], w);
        }
        if (bad) atomicAdd(&hist[0], 1u);
        s += 1;
      }
      store_un<uint16_t>(rec + o, src_load<uint16_t>(a, s, i)); o += 2; s += 1;  // intensity
      {
        const uint32_t rn = src_load<uint8_t>(a, s, i), nr = src_load<uint8_t>(a, s + 1, i);
        s += 2;
        if (rn >= 1 && rn <= a.max_return) atomicAdd(&hist[rn], 1u);  // points_by_return.get_mut(&return_number)
        if constexpr (F.ext) {
          const uint32_t cf = src_load<uint8_t>(a, s, i), sc = src_load<uint8_t>(a, s + 1, i), sd = src_load<uint8_t>(a, s + 2, i),
                         eof = src_load<uint8_t>(a, s + 3, i);
          s += 4;
          store_un<uint8_t>(rec + o, (uint8_t)((rn & 15u) | ((nr & 15u) << 4)));
          store_un<uint8_t>(rec + o + 1, (uint8_t)((cf & 15u) | ((sc & 3u) << 4) | ((sd & 1u) << 6) | ((eof & 1u) << 7)));
          o += 2;
        } else {
          const uint32_t sd = src_load<uint8_t>(a, s, i), eof = src_load<uint8_t>(a, s + 1, i);
          s += 2;
          store_un<uint8_t>(rec + o, (uint8_t)((rn & 7u) | ((nr & 7u) << 3) | ((sd & 1u) << 6) | ((eof & 1u) << 7)));
          o += 1;
        }
      }
      store_un<uint8_t>(rec + o, src_load<uint8_t>(a, s, i)); o += 1; s += 1;  // classification
      if constexpr (F.ext) {
        store_un<uint8_t>(rec + o, src_load<uint8_t>(a, s, i)); o += 1; s += 1;    // user data
        store_un<int16_t>(rec + o, src_load<int16_t>(a, s, i)); o += 2; s += 1;    // scan angle
      } else {
        store_un<int8_t>(rec + o, src_load<int8_t>(a, s, i)); o += 1; s += 1;      // scan angle rank
        store_un<uint8_t>(rec + o, src_load<uint8_t>(a, s, i)); o += 1; s += 1;    // user data
      }
      store_un<uint16_t>(rec + o, src_load<uint16_t>(a, s, i)); o += 2; s += 1;  // point source id
      if constexpr (F.gps) { store_un<double>(rec + o, src_load<double>(a, s, i)); o += 8; s += 1; }
      if constexpr (F.color) {
        cgptr_t cp = (cgptr_t)(as_global(a.attr_base[s]) + i * a.attr_stride[s]);
#pragma unroll
        for (int c = 0; c < 3; ++c) store_un<uint16_t>(rec + o + 2 * c, load_un<uint16_t>(cp + 2 * c));
        o += 6; s += 1;
      }
      if constexpr (F.nir) { store_un<uint16_t>(rec + o, src_load<uint16_t>(a, s, i)); o += 2; s += 1; }
      if constexpr (F.wave) {
        store_un<uint8_t>(rec + o, src_load<uint8_t>(a, s, i)); o += 1; s += 1;
        store_un<uint64_t>(rec + o, src_load<uint64_t>(a, s, i)); o += 8; s += 1;
        store_un<uint32_t>(rec + o, src_load<uint32_t>(a, s, i)); o += 4; s += 1;
        store_un<float>(rec + o, src_load<float>(a, s, i)); o += 4; s += 1;
        cgptr_t wp = (cgptr_t)(as_global(a.attr_base[s]) + i * a.attr_stride[s]);
#pragma unroll
        for (int c = 0; c < 3; ++c) store_un<float>(rec + o + 4 * c, load_un<float>(wp + 4 * c));
        o += 12; s += 1;
      }
    }
    __syncthreads();
    tile_store<kBlock>(lds, as_global(ga - mis), mis, cnt * RS);
    __syncthreads();
  }

  // block fold: bounds through LDS, counters are already block-wide in LDS
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* o = a.partial_bounds + (uint64_t)blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
  (void)counts_acc;
  if (threadIdx.x < kReturnSlots) a.partial_counts[(uint64_t)blockIdx.x * kReturnSlots + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void las_encode_fold_kernel(const double* __restrict__ partial_bounds,
                                                                 const unsigned long long* __restrict__ partial_counts, uint32_t n_blocks,
                                                                 double* __restrict__ out_bounds, unsigned long long* __restrict__ out_counts,
                                                                 double s0, double s1, double s2, double t0, double t1, double t2) {
  double mn[3] = {s0, s1, s2}, mx[3] = {t0, t1, t2};
  for (uint32_t b = threadIdx.x; b < n_blocks; b += kBlock) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = __builtin_fmin(mn[c], partial_bounds[(uint64_t)b * 6 + c]);
      mx[c] = __builtin_fmax(mx[c], partial_bounds[(uint64_t)b * 6 + 3 + c]);
    }
  }
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    out_bounds[0] = mn[0]; out_bounds[1] = mn[1]; out_bounds[2] = mn[2];
    out_bounds[3] = mx[0]; out_bounds[4] = mx[1]; out_bounds[5] = mx[2];
  }
  __shared__ unsigned long long csum[kReturnSlots];
  if (threadIdx.x < kReturnSlots) csum[threadIdx.x] = 0;
  __syncthreads();
  // 16 lanes per counter row: lane l handles counter (l & 15) of blocks l/16, l/16 + 16, ...
  unsigned long long local = 0;
  for (uint32_t b = threadIdx.x >> 4; b < n_blocks; b += kBlock / 16) local += partial_counts[(uint64_t)b * kReturnSlots + (threadIdx.x & 15u)];
  atomicAdd(&csum[threadIdx.x & 15u], local);
  __syncthreads();
  if (threadIdx.x < kReturnSlots) out_counts[threadIdx.x] = csum[threadIdx.x];
}

}  // namespace

namespace pstk {

uint32_t las_raw_record_size(int format) { return raw_size(fmt_of(format)); }

// attr_base / attr_stride: typed attributes in LasPointFormatN field order.  out_bounds (6 doubles) and out_counts (16 u64)
// are device-accessible.  workspace must hold las_encode_workspace_bytes().
size_t las_encode_workspace_bytes() { return (size_t)16384 * (6 * sizeof(double) + kReturnSlots * sizeof(unsigned long long)); }

bool launch_las_encode(int format, const uint64_t* attr_base, const uint32_t* attr_stride, int n_attrs, uint64_t dst, uint64_t n,
                       const double scale[3], const double offset[3], const double bounds_in[6], uint32_t max_return, uint8_t* workspace,
                       double* out_bounds, unsigned long long* out_counts, hipStream_t stream) {
  EncodeArgs a{};
  for (int i = 0; i < n_attrs && i < kMaxAttrs; ++i) { a.attr_base[i] = attr_base[i]; a.attr_stride[i] = attr_stride[i]; }
  a.dst = dst;
  a.n = n;
  for (int c = 0; c < 3; ++c) { a.scale[c] = scale[c]; a.offset[c] = offset[c]; a.seed_min[c] = bounds_in[c]; a.seed_max[c] = bounds_in[3 + c]; }
  a.max_return = max_return;
  const uint32_t rs = las_raw_record_size(format);
  a.tile = std::max<uint32_t>(kBlock, ((32u * 1024u) / rs) / kBlock * kBlock);  // ~32 KiB of records per tile
  const uint64_t n_tiles = std::max<uint64_t>(1, (n + a.tile - 1) / a.tile);
  const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, 16384);
  a.partial_bounds = (double*)workspace;
  a.partial_counts = (unsigned long long*)(workspace + (size_t)16384 * 6 * sizeof(double));
  const size_t lds_bytes = (size_t)a.tile * rs + 32;
#define PST_ENC(N) case N: hipLaunchKernelGGL((las_encode_kernel<N>), dim3(grid), dim3(kBlock), lds_bytes, stream, a); break;
  switch (format) {
    PST_ENC(0) PST_ENC(1) PST_ENC(2) PST_ENC(3) PST_ENC(4) PST_ENC(5) PST_ENC(6) PST_ENC(7) PST_ENC(8) PST_ENC(9) PST_ENC(10)
    default: return false;
  }
#undef PST_ENC
  hipLaunchKernelGGL(las_encode_fold_kernel, dim3(1), dim3(kBlock), 0, stream, (const double*)a.partial_bounds,
                     (const unsigned long long*)a.partial_counts, grid, out_bounds, out_counts, bounds_in[0], bounds_in[1], bounds_in[2], bounds_in[3],
                     bounds_in[4], bounds_in[5]);
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
