// Storage transposition of typed LAS points (LasPointFormatN::layout(), las_types.rs) between a HashMapBuffer (columns) and a
// VectorBuffer (packed records): the identity plan of BufferLayoutConverter::for_layouts(layout, layout) for two buffers of the
// same typed LAS layout but different storage (buffer_conversion.rs:489-604).  BASELINE.json configs[2] is the records -> columns
// direction for format 0.  Same building blocks as las_encode.hip / las_decode.hip with the point format as a template
// parameter: four consecutive points per lane for every narrow attribute, the Vec3f64 position column in its lane-contiguous
// 16-byte chunk layout, records staged in an LDS tile (LDS-DMA in, 16-byte stores out).  Pure byte moves: bit-exact.
#include <type_traits>

#include "device_common.hpp"
#include "kernels.hpp"
#include "las_device.hpp"
#include "tile_io.hpp"

#include <algorithm>

using namespace pstd;
using namespace pstlas;

namespace {

constexpr uint32_t kQuadTile = 4 * kBlock;
constexpr int kResidentTranspose = 0;  // workgroups resident per CU (kernels.hpp lds_with_resident_cap; 0 = whatever fits)

struct TransposeArgs {
  uint64_t aos;                // address of typed record 0 of the range
  uint64_t n;
  uint64_t col[kMaxAttrs];     // typed attribute columns (slot order): address of point 0 of the range
  double* partial_bounds;      // null, or [gridDim.x][6] records {min xyz, max xyz} of the positions moved (fused AABB)
};

__device__ __forceinline__ uint32_t round_up16(uint32_t v) { return (v + 15u) & ~15u; }

// sizes of the typed slots after Position3D, as compile-time lists per format
template <int FORMAT, typename F>
__device__ __forceinline__ void for_each_tail_slot(F&& f) {
  constexpr Fmt M = fmt_of(FORMAT);
  // f(slot, offset_in_record, size) with compile-time arguments through integral_constant-like template lambdas is not
  // available in C++17; the lists are short enough to spell out
  int s = 1;
  uint32_t o = 24;
  auto emit = [&](uint32_t size) __attribute__((always_inline)) { f(s, o, size); s += 1; o += size; };
  emit(2);                      // intensity
  emit(1); emit(1);             // return number, number of returns
  if constexpr (M.ext) { emit(1); emit(1); }  // classification flags, scanner channel
  emit(1); emit(1); emit(1);    // scan direction flag, edge of flight line, classification
  if constexpr (M.ext) { emit(1); emit(2); } else { emit(1); emit(1); }  // (user data, scan angle) / (scan angle rank, user data)
  emit(2);                      // point source id
  if constexpr (M.gps) emit(8);
  if constexpr (M.color) emit(6);
  if constexpr (M.nir) emit(2);
  if constexpr (M.wave) { emit(1); emit(8); emit(4); emit(4); emit(12); }
}

// AABB of the positions a lane moves in the chunk layout: element (j, e) of a lane has component (c0 + (512 j + e) % 3) % 3, so
// three accumulator pairs indexed by the compile-time r = (512 j + e) % 3 are kept "rotated by c0" and un-rotated once.
struct RotBounds {
  double rmn[3] = {kF64Max, kF64Max, kF64Max}, rmx[3] = {-kF64Max, -kF64Max, -kF64Max};
  __device__ __forceinline__ void fold(uint32_t r, uint64_t bits) {
    const double w = __builtin_bit_cast(double, bits);
    rmn[r] = __builtin_fmin(rmn[r], w);
    rmx[r] = __builtin_fmax(rmx[r], w);
  }
  __device__ __forceinline__ void unrotate(uint32_t c0, double (&mn)[3], double (&mx)[3]) const {
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
      const uint32_t r = c >= c0 ? c - c0 : c + 3u - c0;
      mn[c] = __builtin_fmin(mn[c], pick3(r, rmn[0], rmn[1], rmn[2]));
      mx[c] = __builtin_fmax(mx[c], pick3(r, rmx[0], rmx[1], rmx[2]));
    }
  }
};
__device__ __forceinline__ void write_partial_bounds(double* partials, double (&mn)[3], double (&mx)[3]) {
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* o = partials + (uint64_t)blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
}

// ---- columns -> records -------------------------------------------------------------------------------------------------
// -DPST_LAS_ALIGNED_LDS (A/B builds): the record pieces leave for LDS as naturally aligned b8 / b16 / b32 stores (RecordImage::store_aligned)
// instead of one store per piece at whatever alignment the odd record size gives it.  Measured and left OFF: SQ_LDS_UNALIGNED_STALL is 90 % of this
// kernel's LDS-active cycles (profiles/r05_record_side_pmc.txt), yet five aligned stores per double cost more than one stalled one -- typed LAS-0
// columns -> records, 8 interleaved pairs: 1.186 ms as is, 1.263 ms aligned (-6.1 %, 8 of 8, profiles/r05_abab.txt).
#ifdef PST_LAS_ALIGNED_LDS
constexpr bool kAlignedLdsStores = true;
#else
constexpr bool kAlignedLdsStores = false;
#endif
template <int FORMAT>
__global__ __launch_bounds__(kBlock) void las_columns_to_records_kernel(const TransposeArgs a) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t TS = typed_size(F);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  const uint32_t tid = threadIdx.x;
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t n_tiles = (a.n + kQuadTile - 1) / kQuadTile;
  for (uint64_t tile = xcd_block_id(); tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * kQuadTile;
    const uint32_t cnt = (uint32_t)((a.n - first) < kQuadTile ? (a.n - first) : kQuadTile);
    const uint64_t ga = a.aos + first * TS;
    const uint32_t mis = (uint32_t)(ga & 15u);
    if (cnt == kQuadTile) {
      // positions: 16-byte chunks of the Vec3f64 column, lane-contiguous; double d = 2*tid + 512*j + e -> point d/3, component d%3
      u32x4 pc[6];
      cgptr_t pb = (cgptr_t)(as_global(a.col[0]) + first * 24u);
#pragma unroll
      for (int j = 0; j < 6; ++j)
        pc[j] = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x4>::type*>(pb + 16u * (tid + (uint32_t)kBlock * j)));
      // every other column: this lane's four consecutive points
      const uint64_t p0 = first + 4u * tid;
      QuadCol<2> c2[6];
      QuadCol<1> c1[12];
      QuadCol<8> c8[2];
      QuadCol<6> c6;
      QuadCol<4> c4[2];
      QuadCol<12> c12;
      int n1 = 0, n2 = 0, n4 = 0, n8 = 0;
      for_each_tail_slot<FORMAT>([&](int slot, uint32_t, uint32_t size) __attribute__((always_inline)) {
        cgptr_t p = (cgptr_t)(as_global(a.col[slot]) + p0 * size);
        if (size == 1) c1[n1++].load(p);
        else if (size == 2) c2[n2++].load(p);
        else if (size == 4) c4[n4++].load(p);
        else if (size == 8) c8[n8++].load(p);
        else if (size == 6) c6.load(p);
        else c12.load(p);
      });
      const uint32_t d0 = 2u * tid, q0 = d0 / 3u, c0 = d0 - 3u * q0;
      RotBounds rb;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t k = 512u * j + e, A = k / 3u, r = k % 3u;  // compile-time
          const bool wrap = c0 + r >= 3u;
          const uint32_t c = wrap ? c0 + r - 3u : c0 + r, q = q0 + A + (wrap ? 1u : 0u);
          const uint64_t bits = (uint64_t)(e ? pc[j].z : pc[j].x) | ((uint64_t)(e ? pc[j].w : pc[j].y) << 32);
          if constexpr (kAlignedLdsStores) {
            RecordImage<8> img;
            img.w[0] = (uint32_t)bits; img.w[1] = (uint32_t)(bits >> 32);
            img.store_aligned(lds + (mis + q * TS + 8u * c));
          } else {
            store_un<uint64_t>(lds + (mis + q * TS + 8u * c), bits);
          }
          if (a.partial_bounds) rb.fold(r, bits);
        }
      }
      if (a.partial_bounds) rb.unrotate(c0, mn, mx);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        typename std::conditional<kAlignedLdsStores, RecordImage<TS - 24>, RecTail<TS - 24>>::type rec;
        int i1 = 0, i2 = 0, i4 = 0, i8 = 0;
        for_each_tail_slot<FORMAT>([&](int, uint32_t off, uint32_t size) __attribute__((always_inline)) {
          const int o = (int)off - 24;
          if (size == 1) rec.put(o, 1, c1[i1++].value(t));
          else if (size == 2) rec.put(o, 2, c2[i2++].value(t));
          else if (size == 4) rec.put(o, 4, c4[i4++].value(t));
          else if (size == 8) rec.put(o, 8, c8[i8++].value(t));
          else if (size == 6) rec.put(o, 6, c6.value(t));
          else { rec.put(o, 8, c12.bytes_at(12 * t)); rec.put(o + 8, 4, c12.bytes_at(12 * t + 8) & 0xFFFFFFFFull); }
        });
        if constexpr (kAlignedLdsStores) rec.store_aligned(lds + (mis + (4u * tid + t) * TS + 24u));
        else rec.store(lds + (mis + (4u * tid + t) * TS + 24u));
      }
    } else {
      for (uint32_t lp = tid; lp < cnt; lp += kBlock) {  // ragged last tile: one lane per point
        const uint64_t i = first + lp;
        lptr_t rec = lds + (mis + lp * TS);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint64_t bits = load_un<uint64_t>((cgptr_t)as_global(a.col[0]) + i * 24u + 8u * c);
          store_un<uint64_t>(rec + 8 * c, bits);
          const double w = __builtin_bit_cast(double, bits);
          mn[c] = __builtin_fmin(mn[c], w);
          mx[c] = __builtin_fmax(mx[c], w);
        }
        for_each_tail_slot<FORMAT>([&](int slot, uint32_t off, uint32_t size) __attribute__((always_inline)) {
          cgptr_t p = (cgptr_t)as_global(a.col[slot]) + i * size;
          for (uint32_t b = 0; b < size; ++b) rec[off + b] = p[b];
        });
      }
    }
    __syncthreads();
    tile_store<kBlock>(lds, as_global(ga - mis), mis, cnt * TS);
    __syncthreads();
  }
  if (a.partial_bounds) write_partial_bounds(a.partial_bounds, mn, mx);
}

// ---- records -> columns -------------------------------------------------------------------------------------------------
template <int FORMAT>
__global__ __launch_bounds__(kBlock) void las_records_to_columns_kernel(const TransposeArgs a) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t TS = typed_size(F);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  const uint32_t tid = threadIdx.x;
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t n_tiles = (a.n + kQuadTile - 1) / kQuadTile;
  for (uint64_t tile = xcd_block_id(); tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * kQuadTile;
    const uint32_t cnt = (uint32_t)((a.n - first) < kQuadTile ? (a.n - first) : kQuadTile);
    const uint64_t sa = a.aos + first * TS;
    const uint32_t smis = (uint32_t)(sa & 15u);
    tile_load<kBlock>(lds, as_global(sa - smis), round_up16(smis + cnt * TS));
    wait_tile_loads();
    __syncthreads();
    if (cnt == kQuadTile) {
      gptr_t pcol = as_global(a.col[0]) + first * 24u;
      const uint32_t d0 = 2u * tid, q0 = d0 / 3u, c0 = d0 - 3u * q0;
      uint64_t pv[12];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t k = 512u * j + e, A = k / 3u, r = k % 3u;
          const bool wrap = c0 + r >= 3u;
          const uint32_t c = wrap ? c0 + r - 3u : c0 + r, q = q0 + A + (wrap ? 1u : 0u);
          pv[2 * j + e] = lds_load<uint64_t>(lds + (smis + q * TS + 8u * c));
        }
      }
      if (a.partial_bounds) {
        RotBounds rb;
#pragma unroll
        for (int j = 0; j < 6; ++j) { rb.fold((512u * j) % 3u, pv[2 * j]); rb.fold((512u * j + 1u) % 3u, pv[2 * j + 1]); }
        rb.unrotate(c0, mn, mx);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        u32x4 v;
        v.x = (uint32_t)pv[2 * j]; v.y = (uint32_t)(pv[2 * j] >> 32); v.z = (uint32_t)pv[2 * j + 1]; v.w = (uint32_t)(pv[2 * j + 1] >> 32);
        __builtin_nontemporal_store(v, reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x4>::type*>(pcol + 16u * (tid + (uint32_t)kBlock * j)));
      }
      Pack4<2> c2[6];
      Pack4<1> c1[12];
      Pack4<8> c8[2];
      Pack4<6> c6;
      Pack4<4> c4[2];
      Pack4<12> c12;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const LdsBytes<TS - 24> rec(lds + (smis + (4u * tid + t) * TS + 24u));
        int i1 = 0, i2 = 0, i4 = 0, i8 = 0;
        for_each_tail_slot<FORMAT>([&](int, uint32_t off, uint32_t size) __attribute__((always_inline)) {
          const int o = (int)off - 24;
          if (size == 1) c1[i1++].put(t, rec.at(o));
          else if (size == 2) c2[i2++].put(t, rec.at(o));
          else if (size == 4) c4[i4++].put(t, rec.at(o));
          else if (size == 8) c8[i8++].put(t, rec.at(o));
          else if (size == 6) c6.put(t, rec.at(o));
          else { c12.put_at(12 * t, 8, rec.at(o)); c12.put_at(12 * t + 8, 4, rec.at(o + 8) & 0xFFFFFFFFull); }
        });
      }
      const uint64_t p0 = first + 4u * tid;
      int n1 = 0, n2 = 0, n4 = 0, n8 = 0;
      for_each_tail_slot<FORMAT>([&](int slot, uint32_t, uint32_t size) __attribute__((always_inline)) {
        gptr_t p = as_global(a.col[slot]) + p0 * size;
        if (size == 1) c1[n1++].store(p);
        else if (size == 2) c2[n2++].store(p);
        else if (size == 4) c4[n4++].store(p);
        else if (size == 8) c8[n8++].store(p);
        else if (size == 6) c6.store(p);
        else c12.store(p);
      });
    } else {
      for (uint32_t lp = tid; lp < cnt; lp += kBlock) {
        const uint64_t i = first + lp;
        clptr_t rec = lds + (smis + lp * TS);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint64_t bits = lds_load<uint64_t>(rec + 8 * c);
          store_un<uint64_t>(as_global(a.col[0]) + i * 24u + 8u * c, bits);
          const double w = __builtin_bit_cast(double, bits);
          mn[c] = __builtin_fmin(mn[c], w);
          mx[c] = __builtin_fmax(mx[c], w);
        }
        for_each_tail_slot<FORMAT>([&](int slot, uint32_t off, uint32_t size) __attribute__((always_inline)) {
          gptr_t p = as_global(a.col[slot]) + i * size;
          for (uint32_t b = 0; b < size; ++b) p[b] = rec[off + b];
        });
      }
    }
    __syncthreads();  // the next tile's DMA overwrites the records
  }
  if (a.partial_bounds) write_partial_bounds(a.partial_bounds, mn, mx);
}

}  // namespace

namespace pstk {

// to_records = true: columns -> packed typed records; false: records -> columns.  cols in LasPointFormatN slot order.
unsigned las_transpose_grid(uint64_t n) {
  const uint64_t n_tiles = std::max<uint64_t>(1, (n + kQuadTile - 1) / kQuadTile);
  return (unsigned)((std::min<uint64_t>(n_tiles, 1u << 22) + 7) / 8 * 8);  // one tile per block; multiple of 8 for xcd_block_id()
}

// partials: null, or las_transpose_grid(n) records of 6 doubles (fused AABB of the positions; finish with launch_finalize_bounds)
bool launch_las_transpose(int format, bool to_records, uint64_t aos, const uint64_t* cols, int n_cols, uint64_t n, double* partials,
                          hipStream_t stream) {
  TransposeArgs a{};
  a.partial_bounds = partials;
  a.aos = aos;
  a.n = n;
  for (int i = 0; i < n_cols && i < kMaxAttrs; ++i) a.col[i] = cols[i];
  const unsigned grid = las_transpose_grid(n);
  const size_t lds_bytes = lds_with_resident_cap((size_t)kQuadTile * typed_size(fmt_of(format)) + 64, kResidentTranspose);
#define PST_TR(N)                                                                                                                                   \
  case N: {                                                                                                                                         \
    if (to_records) {                                                                                                                               \
      static const hipError_t at1 = hipFuncSetAttribute((const void*)las_columns_to_records_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
      (void)at1;                                                                                                                                    \
      hipLaunchKernelGGL((las_columns_to_records_kernel<N>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);                                      \
    } else {                                                                                                                                        \
      static const hipError_t at2 = hipFuncSetAttribute((const void*)las_records_to_columns_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
      (void)at2;                                                                                                                                    \
      hipLaunchKernelGGL((las_records_to_columns_kernel<N>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);                                      \
    }                                                                                                                                               \
    break;                                                                                                                                          \
  }
  switch (format) {
    PST_TR(0) PST_TR(1) PST_TR(2) PST_TR(3) PST_TR(4) PST_TR(5) PST_TR(6) PST_TR(7) PST_TR(8) PST_TR(9) PST_TR(10)
    default: return false;
  }
#undef PST_TR
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
