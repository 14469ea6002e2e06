// LAS record decoder for gfx950: raw LAS point records (exact-binary layout, las_layout.rs:70-107) -> the format's typed default
// layout (LasPointFormatN::layout(), las_types.rs) in COLUMNAR storage, i.e. the plan that get_default_las_converter
// (pasture-io/src/las/raw_readers.rs:31-167) builds for RawLASReader::read_into_custom_layout (:299-352):
//   Position3D  = (LASLocalPosition as f64 * scale) + offset         (two roundings, :42-48)
//   ReturnNumber / NumberOfReturns / ScanDirectionFlag / EdgeOfFlightLine [/ ClassificationFlags / ScannerChannel]
//               = (flags >> shift) & mask                              (:61-164)
//   everything else copied.
// The generic converter interprets that plan entry by entry (convert.hip, 5.0-5.3 TB/s); this kernel is the same plan with the
// point format as a template parameter — the mirror image of las_encode.hip:  the record tile is staged with LDS-DMA; positions
// are produced in the lane-contiguous 16-byte chunk layout of the Vec3f64 column (component phase fixed per lane, accumulators
// rotated, one coalesced dwordx4 store per chunk); every other column receives the values of four consecutive points per
// lane in one vector store (u8 columns: one dword, u16: 8 bytes, f64: 32 bytes ...).  The AABB of the result can be folded in.
// converter.cpp routes a conversion here only when layouts AND mappings match the plan exactly; everything else stays generic.
#include "device_common.hpp"
#include "kernels.hpp"
#include "las_device.hpp"
#include "tile_io.hpp"

#include <algorithm>

using namespace pstd;
using namespace pstlas;

namespace {

constexpr uint32_t kQuadTile = 4 * kBlock;
// workgroups resident per CU (kernels.hpp lds_with_resident_cap; 0 = whatever fits)
constexpr int kResidentDecode = 0, kResidentDecodeAos = 0;

struct DecodeArgs {
  uint64_t src;                  // address of raw record 0 of the source range
  uint64_t n;
  uint64_t attr_dst[kMaxAttrs];  // typed attribute columns (LasPointFormatN slot order): address of target point 0
  double scale[3], offset[3];
  double* partial_bounds;        // [grid][6] or null
};

__device__ __forceinline__ uint32_t round_up16(uint32_t v) { return (v + 15u) & ~15u; }

// one full tile: records staged at lds + smis, typed point index of the tile's first point = first
template <int FORMAT>
__device__ __forceinline__ void decode_quad_tile(const DecodeArgs& a, uint64_t first, clptr_t lds, uint32_t smis, double (&mn)[3], double (&mx)[3]) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t RS = raw_size(F);
  const uint32_t tid = threadIdx.x;
  // ---- positions: double d = 2*tid + 512*j + e of the tile's Vec3f64 stream belongs to point d/3, component d%3 ----
  {
    const uint32_t d0 = 2u * tid, q0 = d0 / 3u, c0 = d0 - 3u * q0;
    double sc_r[3], of_r[3], rmn[3], rmx[3];
#pragma unroll
    for (uint32_t r = 0; r < 3; ++r) {
      const uint32_t c = c0 + r >= 3u ? c0 + r - 3u : c0 + r;
      sc_r[r] = pick3(c, a.scale[0], a.scale[1], a.scale[2]);
      of_r[r] = pick3(c, a.offset[0], a.offset[1], a.offset[2]);
      rmn[r] = pick3(c, mn[0], mn[1], mn[2]);
      rmx[r] = pick3(c, mx[0], mx[1], mx[2]);
    }
    gptr_t pcol = as_global(a.attr_dst[0]) + first * 24u;
    int32_t raw[12];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t k = 512u * j + e, A = k / 3u, r = k % 3u;  // compile-time
        const bool wrap = c0 + r >= 3u;
        const uint32_t c = wrap ? c0 + r - 3u : c0 + r, q = q0 + A + (wrap ? 1u : 0u);
        raw[2 * j + e] = lds_load<int32_t>(lds + (smis + q * RS + 4u * c));
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double w[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t r = (512u * j + e) % 3u;
        const double m = (double)raw[2 * j + e] * sc_r[r];  // (pos as f64 * scale) + offset, raw_readers.rs:42-48
        w[e] = m + of_r[r];
        rmn[r] = __builtin_fmin(rmn[r], w[e]);
        rmx[r] = __builtin_fmax(rmx[r], w[e]);
      }
      u32x4 v;
      __builtin_memcpy(&v, w, 16);
      __builtin_nontemporal_store(v, reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x4>::type*>(pcol + 16u * (tid + (uint32_t)kBlock * j)));
    }
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
      const uint32_t r = c >= c0 ? c - c0 : c + 3u - c0;
      mn[c] = pick3(r, rmn[0], rmn[1], rmn[2]);
      mx[c] = pick3(r, rmx[0], rmx[1], rmx[2]);
    }
  }
  // ---- every other attribute: this lane's four consecutive points ----
  Pack4<2> intensity, sa, psid, nir;
  Pack4<1> f0, f1, cls, sar, ud, widx;
  Pack4<8> gps, woff;
  Pack4<6> color;
  Pack4<4> wsize, wloc;
  Pack4<12> wpar;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const LdsBytes<RS - 12> rec(lds + (smis + (4u * tid + t) * RS + 12u));
    int o = 0;
    intensity.put(t, rec.at(o)); o += 2;
    f0.put(t, rec.at(o)); o += 1;
    if constexpr (F.ext) { f1.put(t, rec.at(o)); o += 1; }
    cls.put(t, rec.at(o)); o += 1;
    if constexpr (F.ext) { ud.put(t, rec.at(o)); sa.put(t, rec.at(o + 1)); o += 3; }
    else { sar.put(t, rec.at(o)); ud.put(t, rec.at(o + 1)); o += 2; }
    psid.put(t, rec.at(o)); o += 2;
    if constexpr (F.gps) { gps.put(t, rec.at(o)); o += 8; }
    if constexpr (F.color) { color.put(t, rec.at(o)); o += 6; }
    if constexpr (F.nir) { nir.put(t, rec.at(o)); o += 2; }
    if constexpr (F.wave) {
      widx.put(t, rec.at(o)); woff.put(t, rec.at(o + 1)); wsize.put(t, rec.at(o + 9)); wloc.put(t, rec.at(o + 13));
      wpar.put_at(12 * t, 8, rec.at(o + 17)); wpar.put_at(12 * t + 8, 4, rec.at(o + 25) & 0xFFFFFFFFull);
      o += 29;
    }
  }
  const uint64_t p0 = first + 4u * tid;
  int s = 1;
  auto col = [&](int slot, uint32_t bytes) -> gptr_t { return as_global(a.attr_dst[slot]) + p0 * bytes; };
  intensity.store(col(s, 2)); s += 1;
  // bit fields of four points at once ((flags >> shift) & mask, raw_readers.rs:61-164)
  Pack4<1> rn, nr, cf, sc, sd, eof;
  if constexpr (F.ext) {
    rn.w[0] = f0.w[0] & 0x0F0F0F0Fu; nr.w[0] = (f0.w[0] >> 4) & 0x0F0F0F0Fu;
    cf.w[0] = f1.w[0] & 0x0F0F0F0Fu; sc.w[0] = (f1.w[0] >> 4) & 0x03030303u;
    sd.w[0] = (f1.w[0] >> 6) & 0x01010101u; eof.w[0] = (f1.w[0] >> 7) & 0x01010101u;
  } else {
    rn.w[0] = f0.w[0] & 0x07070707u; nr.w[0] = (f0.w[0] >> 3) & 0x07070707u;
    sd.w[0] = (f0.w[0] >> 6) & 0x01010101u; eof.w[0] = (f0.w[0] >> 7) & 0x01010101u;
  }
  rn.store(col(s, 1)); nr.store(col(s + 1, 1)); s += 2;
  if constexpr (F.ext) { cf.store(col(s, 1)); sc.store(col(s + 1, 1)); s += 2; }
  sd.store(col(s, 1)); eof.store(col(s + 1, 1)); cls.store(col(s + 2, 1)); s += 3;
  if constexpr (F.ext) { ud.store(col(s, 1)); sa.store(col(s + 1, 2)); s += 2; }
  else { sar.store(col(s, 1)); ud.store(col(s + 1, 1)); s += 2; }
  psid.store(col(s, 2)); s += 1;
  if constexpr (F.gps) { gps.store(col(s, 8)); s += 1; }
  if constexpr (F.color) { color.store(col(s, 6)); s += 1; }
  if constexpr (F.nir) { nir.store(col(s, 2)); s += 1; }
  if constexpr (F.wave) {
    widx.store(col(s, 1)); woff.store(col(s + 1, 8)); wsize.store(col(s + 2, 4)); wloc.store(col(s + 3, 4)); wpar.store(col(s + 4, 12));
    s += 5;
  }
}

// one point (ragged last tile): record at `rec` in LDS, typed point index i
template <int FORMAT>
__device__ __forceinline__ void decode_point(const DecodeArgs& a, uint64_t i, clptr_t rec, double (&mn)[3], double (&mx)[3]) {
  constexpr Fmt F = fmt_of(FORMAT);
  auto dst = [&](int slot, uint32_t bytes) -> gptr_t { return as_global(a.attr_dst[slot]) + i * bytes; };
  int s = 0;
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double m = (double)lds_load<int32_t>(rec + 4 * c) * a.scale[c];
    const double w = m + a.offset[c];
    store_un<double>(dst(0, 24) + 8 * c, w);
    mn[c] = __builtin_fmin(mn[c], w);
    mx[c] = __builtin_fmax(mx[c], w);
  }
  o = 12; s = 1;
  store_un<uint16_t>(dst(s, 2), lds_load<uint16_t>(rec + o)); o += 2; s += 1;
  if constexpr (F.ext) {
    const uint32_t g0 = rec[o], g1 = rec[o + 1];
    o += 2;
    store_un<uint8_t>(dst(s, 1), (uint8_t)(g0 & 15u)); store_un<uint8_t>(dst(s + 1, 1), (uint8_t)(g0 >> 4));
    store_un<uint8_t>(dst(s + 2, 1), (uint8_t)(g1 & 15u)); store_un<uint8_t>(dst(s + 3, 1), (uint8_t)((g1 >> 4) & 3u));
    store_un<uint8_t>(dst(s + 4, 1), (uint8_t)((g1 >> 6) & 1u)); store_un<uint8_t>(dst(s + 5, 1), (uint8_t)(g1 >> 7));
    s += 6;
  } else {
    const uint32_t g = rec[o];
    o += 1;
    store_un<uint8_t>(dst(s, 1), (uint8_t)(g & 7u)); store_un<uint8_t>(dst(s + 1, 1), (uint8_t)((g >> 3) & 7u));
    store_un<uint8_t>(dst(s + 2, 1), (uint8_t)((g >> 6) & 1u)); store_un<uint8_t>(dst(s + 3, 1), (uint8_t)(g >> 7));
    s += 4;
  }
  store_un<uint8_t>(dst(s, 1), rec[o]); o += 1; s += 1;  // classification
  if constexpr (F.ext) {
    store_un<uint8_t>(dst(s, 1), rec[o]); o += 1; s += 1;                              // user data
    store_un<uint16_t>(dst(s, 2), lds_load<uint16_t>(rec + o)); o += 2; s += 1;        // scan angle
  } else {
    store_un<uint8_t>(dst(s, 1), rec[o]); o += 1; s += 1;                              // scan angle rank
    store_un<uint8_t>(dst(s, 1), rec[o]); o += 1; s += 1;                              // user data
  }
  store_un<uint16_t>(dst(s, 2), lds_load<uint16_t>(rec + o)); o += 2; s += 1;          // point source id
  if constexpr (F.gps) { store_un<uint64_t>(dst(s, 8), lds_load<uint64_t>(rec + o)); o += 8; s += 1; }
  if constexpr (F.color) {
#pragma unroll
    for (int c = 0; c < 3; ++c) store_un<uint16_t>(dst(s, 6) + 2 * c, lds_load<uint16_t>(rec + o + 2 * c));
    o += 6; s += 1;
  }
  if constexpr (F.nir) { store_un<uint16_t>(dst(s, 2), lds_load<uint16_t>(rec + o)); o += 2; s += 1; }
  if constexpr (F.wave) {
    store_un<uint8_t>(dst(s, 1), rec[o]); o += 1; s += 1;
    store_un<uint64_t>(dst(s, 8), lds_load<uint64_t>(rec + o)); o += 8; s += 1;
    store_un<uint32_t>(dst(s, 4), lds_load<uint32_t>(rec + o)); o += 4; s += 1;
    store_un<uint32_t>(dst(s, 4), lds_load<uint32_t>(rec + o)); o += 4; s += 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) store_un<uint32_t>(dst(s, 12) + 4 * c, lds_load<uint32_t>(rec + o + 4 * c));
    o += 12; s += 1;
  }
}

// ---- interleaved typed target (VectorBuffer of LasPointFormatN): one lane per point -------------------------------------
// The typed record (packed, las_types.rs) is assembled in registers at compile-time offsets and written to an LDS record tile
// with dword stores; the tile leaves with 16-byte stores.  Source records come from the LDS-DMA staged tile as aligned dwords.
struct DecodeAosArgs {
  uint64_t src, dst;  // raw record 0 of the source range / typed record 0 of the target range
  uint64_t n;
  double scale[3], offset[3];
  double* partial_bounds;
  uint32_t tile;
};

template <int FORMAT>
__global__ __launch_bounds__(kBlock) void las_decode_aos_kernel(const DecodeAosArgs a) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t RS = raw_size(F), TS = typed_size(F);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds_src = (lptr_t)lds_raw;
  lptr_t lds_dst = lds_src + round_up16(a.tile * RS + 32u);
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t n_tiles = (a.n + a.tile - 1) / a.tile;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * a.tile;
    const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
    const uint64_t sa = a.src + first * RS, da = a.dst + first * TS;
    const uint32_t smis = (uint32_t)(sa & 15u), dmis = (uint32_t)(da & 15u);
    tile_load<kBlock>(lds_src, as_global(sa - smis), round_up16(smis + cnt * RS));
    wait_tile_loads();
    __syncthreads();
    for (uint32_t lp = threadIdx.x; lp < cnt; lp += kBlock) {
      const LdsBytes<RS> rec(lds_src + (smis + lp * RS));
      RecordImage<TS> out;
      int o = 0, t = 0;  // raw cursor, typed cursor
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int32_t raw = (int32_t)(uint32_t)rec.at(4 * c);
        const double m = (double)raw * a.scale[c];  // (pos as f64 * scale) + offset, raw_readers.rs:42-48
        const double w = m + a.offset[c];
        out.put(8 * c, 8, __builtin_bit_cast(uint64_t, w));
        mn[c] = __builtin_fmin(mn[c], w);
        mx[c] = __builtin_fmax(mx[c], w);
      }
      o = 12; t = 24;
      out.put(t, 2, rec.at(o) & 0xFFFFull); o += 2; t += 2;  // intensity
      if constexpr (F.ext) {
        const uint32_t g0 = (uint32_t)rec.at(o) & 255u, g1 = (uint32_t)rec.at(o + 1) & 255u;
        o += 2;
        out.put(t, 1, g0 & 15u); out.put(t + 1, 1, g0 >> 4); out.put(t + 2, 1, g1 & 15u); out.put(t + 3, 1, (g1 >> 4) & 3u);
        out.put(t + 4, 1, (g1 >> 6) & 1u); out.put(t + 5, 1, g1 >> 7);
        t += 6;
      } else {
        const uint32_t g = (uint32_t)rec.at(o) & 255u;
        o += 1;
        out.put(t, 1, g & 7u); out.put(t + 1, 1, (g >> 3) & 7u); out.put(t + 2, 1, (g >> 6) & 1u); out.put(t + 3, 1, g >> 7);
        t += 4;
      }
      // classification, (user data, scan angle) / (scan angle rank, user data), point source id: same order and sizes on both sides
      constexpr int kMid = F.ext ? 6 : 5;
      out.put(t, kMid, rec.at(o) & ((1ull << (8 * kMid)) - 1ull)); o += kMid; t += kMid;
      // GPS time, colour, NIR, waveform: identical bytes
      constexpr int kTail = (F.gps ? 8 : 0) + (F.color ? 6 : 0) + (F.nir ? 2 : 0) + (F.wave ? 29 : 0);
#pragma unroll
      for (int b = 0; b < kTail; b += 8) {
        const int nb = kTail - b < 8 ? kTail - b : 8;
        const uint64_t v = rec.at(o + b);
        out.put(t + b, nb, nb == 8 ? v : (v & ((1ull << (8 * nb)) - 1ull)));
      }
      out.store(lds_dst + (dmis + lp * TS));
    }
    __syncthreads();
    tile_store<kBlock>(lds_dst, as_global(da - dmis), dmis, cnt * TS);
    __syncthreads();
  }
  if (a.partial_bounds) {
    __shared__ double scratch[(kBlock / 64) * 6];
    block_reduce_minmax<double, 3>(mn, mx, scratch);
    if (threadIdx.x == 0) {
      double* o6 = a.partial_bounds + (uint64_t)blockIdx.x * 6;
      o6[0] = mn[0]; o6[1] = mn[1]; o6[2] = mn[2]; o6[3] = mx[0]; o6[4] = mx[1]; o6[5] = mx[2];
    }
  }
}

template <int FORMAT>
__global__ __launch_bounds__(kBlock) void las_decode_kernel(const DecodeArgs a) {
  constexpr uint32_t RS = raw_size(fmt_of(FORMAT));
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t n_tiles = (a.n + kQuadTile - 1) / kQuadTile;
  for (uint64_t tile = xcd_block_id(); tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * kQuadTile;
    const uint32_t cnt = (uint32_t)((a.n - first) < kQuadTile ? (a.n - first) : kQuadTile);
    const uint64_t sa = a.src + first * RS;
    const uint32_t smis = (uint32_t)(sa & 15u);
    tile_load<kBlock>(lds, as_global(sa - smis), round_up16(smis + cnt * RS));
    wait_tile_loads();
    __syncthreads();
    if (cnt == kQuadTile) decode_quad_tile<FORMAT>(a, first, lds, smis, mn, mx);
    else
      for (uint32_t lp = threadIdx.x; lp < cnt; lp += kBlock) decode_point<FORMAT>(a, first + lp, lds + (smis + lp * RS), mn, mx);
    __syncthreads();  // the next tile's DMA overwrites the records
  }
  if (a.partial_bounds) {
    __shared__ double scratch[(kBlock / 64) * 6];
    block_reduce_minmax<double, 3>(mn, mx, scratch);
    if (threadIdx.x == 0) {
      double* o = a.partial_bounds + (uint64_t)blockIdx.x * 6;
      o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
    }
  }
}

}  // namespace

namespace pstk {

unsigned las_decode_grid(uint64_t n) {
  const uint64_t n_tiles = std::max<uint64_t>(1, (n + kQuadTile - 1) / kQuadTile);
  return (unsigned)((std::min<uint64_t>(n_tiles, 1u << 22) + 7) / 8 * 8);  // one tile per block: +5 % over a persistent grid of 16384 (same-box A/B)
}

// dst_cols: typed attribute columns in LasPointFormatN slot order (address of the first target point).  partials: null, or
// las_decode_grid(n) records of 6 doubles for the fused AABB (finish with launch_finalize_bounds).
bool launch_las_decode(int format, uint64_t src, uint64_t n, const uint64_t* dst_cols, int n_cols, const double scale[3], const double offset[3],
                       double* partials, hipStream_t stream) {
  DecodeArgs a{};
  a.src = src;
  a.n = n;
  for (int i = 0; i < n_cols && i < kMaxAttrs; ++i) a.attr_dst[i] = dst_cols[i];
  for (int c = 0; c < 3; ++c) { a.scale[c] = scale[c]; a.offset[c] = offset[c]; }
  a.partial_bounds = partials;
  const unsigned grid = las_decode_grid(n);
  const size_t lds_bytes = lds_with_resident_cap((size_t)kQuadTile * raw_size(fmt_of(format)) + 64, kResidentDecode);
#define PST_DEC(N)                                                                                                                          \
  case N: {                                                                                                                                 \
    static const hipError_t attr = hipFuncSetAttribute((const void*)las_decode_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    (void)attr;                                                                                                                             \
    hipLaunchKernelGGL((las_decode_kernel<N>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);                                            \
    break;                                                                                                                                  \
  }
  switch (format) {
    PST_DEC(0) PST_DEC(1) PST_DEC(2) PST_DEC(3) PST_DEC(4) PST_DEC(5) PST_DEC(6) PST_DEC(7) PST_DEC(8) PST_DEC(9) PST_DEC(10)
    default: return false;
  }
#undef PST_DEC
  return hipGetLastError() == hipSuccess;
}


static uint32_t las_decode_aos_tile(int format) {
  const uint32_t per_point = raw_size(fmt_of(format)) + typed_size(fmt_of(format));
  return std::max<uint32_t>(kBlock, ((48u * 1024u) / per_point) / kBlock * kBlock);
}
unsigned las_decode_aos_grid(int format, uint64_t n) {
  const uint32_t tile = las_decode_aos_tile(format);
  return (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, (n + tile - 1) / tile), 16384);  // persistent grid: one tile per block loses 4 % here
}
// raw records -> interleaved typed records (VectorBuffer of LasPointFormatN)
bool launch_las_decode_aos(int format, uint64_t src, uint64_t dst, uint64_t n, const double scale[3], const double offset[3], double* partials,
                           hipStream_t stream) {
  DecodeAosArgs a{};
  a.src = src;
  a.dst = dst;
  a.n = n;
  for (int c = 0; c < 3; ++c) { a.scale[c] = scale[c]; a.offset[c] = offset[c]; }
  a.partial_bounds = partials;
  a.tile = las_decode_aos_tile(format);
  const unsigned grid = las_decode_aos_grid(format, n);
  const Fmt f = fmt_of(format);
  const size_t lds_bytes = lds_with_resident_cap((((size_t)a.tile * raw_size(f) + 32 + 15) & ~(size_t)15) + (size_t)a.tile * typed_size(f) + 64, kResidentDecodeAos);
#define PST_DEC(N)                                                                                                                              \
  case N: {                                                                                                                                     \
    static const hipError_t attr = hipFuncSetAttribute((const void*)las_decode_aos_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    (void)attr;                                                                                                                                 \
    hipLaunchKernelGGL((las_decode_aos_kernel<N>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);                                            \
    break;                                                                                                                                      \
  }
  switch (format) {
    PST_DEC(0) PST_DEC(1) PST_DEC(2) PST_DEC(3) PST_DEC(4) PST_DEC(5) PST_DEC(6) PST_DEC(7) PST_DEC(8) PST_DEC(9) PST_DEC(10)
    default: return false;
  }
#undef PST_DEC
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
