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
#include "las_device.hpp"
#include "tile_io.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace pstd;
using namespace pstlas;

namespace {

constexpr int kReturnSlots = 16;  // slot r counts return number r (1..15); slot 0 = out-of-range positions

struct EncodeArgs {
  uint64_t attr_base[kMaxAttrs];    // address of the typed attribute of point 0 (layout slot order of LasPointFormatN)
  uint32_t attr_stride[kMaxAttrs];  // bytes between consecutive points of that attribute
  uint64_t dst;                     // address of raw record 0 of the target range
  uint64_t n;
  double scale[3], offset[3];
  double rscale[3];                 // RN(1 / scale[c]) when fast_div (host: launch_las_encode)
  uint32_t fast_div;                // 1: every scale passes quotient_by_reciprocal's preconditions; 0: the IEEE division instruction sequence
  double seed_min[3], seed_max[3];  // the header's current bounds (raw_writers.rs:136-141: f64::MAX / f64::MIN initially)
  double* partial_bounds;           // [grid][6]
  unsigned long long* partial_counts;  // [grid][kReturnSlots]
  uint32_t max_return;              // 5 (legacy header) or 15 (large_file) — raw_writers.rs:221-225
  uint32_t tile;                    // points per tile (multiple of kBlock)
};

// (An alternative, OFF by default -- see las_encode_fast_div_allowed: built on the round-4 review's suggestion, proven and tested, and measured
// to change nothing.)  a / b for a divisor that is the same for every point, WITHOUT the division sequence (v_div_scale, the quarter-rate
// v_rcp_f64, six fused multiply-adds, v_div_fmas, v_div_fixup: a third of this kernel's vector instructions) and still the CORRECTLY ROUNDED
// quotient, which the record needs bit for bit: LAS positions are multiples of the scale, so (p - offset) / scale sits within an ulp of an integer
// for nearly every point and the truncation of write_helpers.rs:15-17 sees the last bit.
//   y = RN(1 / b) (host, one IEEE division);  q0 = RN(a y)  -- relative error <= 2^-52 (1 + 2^-53), up to two ulps;
//   r0 = RN(a - b q0), q1 = RN(q0 + r0 y)  -- q0 + r0 y = a/b + (a/b - q0)(b y - 1) (+ the rounding of r0, itself <= 2^-53 |r0|): within
//   2^-104 |a/b| of the quotient before the rounding, so q1 is a FAITHFUL rounding of a/b (one of the two doubles next to it);
//   r1 = a - b q1 is then exact (the residual of a faithful quotient is representable) and q2 = RN(q1 + r1 y) is RN(a / b): Markstein's division theorem
//   (P. Markstein, "Computation of elementary functions on the IBM RISC System/6000 processor", IBM J. Res. Dev. 34, 1990 -- the correction step
//   every fused-multiply-add division ends with), which needs y = RN(1 / b); divisors whose significand is all ones (where 1 / b rounds worst) are
//   left to the division sequence as well.
// Preconditions checked on the host (fast_div): b finite, normal, 2^-400 <= |b| <= 2^400, significand not all ones.  Per value: |q0| < 2^62 and
// |a| >= 2^-500 or the plain quotient is taken (overflow, infinities and NaN propagate through a y exactly as through a / b as far as the checked
// narrowing can tell: +-inf and NaN stay what they are, and a quotient that large is out of the i32 range either way; a tiny |a| gives |a / b| < 1/2
// in both forms = the integer 0, but its residuals could underflow, so it does not take the correction steps).  Five full-rate instructions.
// tests/test_gpu_las_encode.py::test_reciprocal_division_is_the_ieee_quotient compares both forms of this kernel over adversarial and random inputs.
__device__ __forceinline__ double quotient_by_reciprocal(double a, double b, double y) {
  const double q0 = a * y;
  const double r0 = __builtin_fma(-q0, b, a);
  const double q1 = __builtin_fma(r0, y, q0);
  const double r1 = __builtin_fma(-q1, b, a);
  const double q2 = __builtin_fma(r1, y, q1);
  return (__builtin_fabs(q0) < 0x1p62 && __builtin_fabs(a) >= 0x1p-500) ? q2 : q0;
}

// Where one point's typed attributes are read from: HBM at any per-attribute stride, or a record staged in LDS.
struct GlobalSrc {
  const EncodeArgs& a;
  uint64_t i;
  template <typename T>
  __device__ __forceinline__ T get(int slot, uint32_t off = 0) const {
    return load_un<T>((cgptr_t)(as_global(a.attr_base[slot]) + i * a.attr_stride[slot] + off));
  }
};
// One interleaved typed record staged in LDS at ANY byte alignment: read as aligned dwords and re-aligned in registers
// with v_alignbyte (unaligned ds accesses stall the LDS pipe: records of odd size make 3 of 4 lanes unaligned).
template <int FORMAT>
struct RecordSrc {
  static constexpr uint32_t TS = typed_size(fmt_of(FORMAT)), NW = (TS + 3) / 4;
  uint32_t r[NW + 2];
  __device__ __forceinline__ explicit RecordSrc(clptr_t rec) {
    const uint32_t addr = (uint32_t)(uintptr_t)rec, m = addr & 3u;
    const PST_AS_LDS uint32_t* p = (const PST_AS_LDS uint32_t*)(rec - m);
    uint32_t d[NW + 1];
#pragma unroll
    for (uint32_t k = 0; k <= NW; ++k) d[k] = p[k];
#pragma unroll
    for (uint32_t k = 0; k < NW; ++k) r[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], m);
    r[NW] = 0; r[NW + 1] = 0;
  }
  template <typename T>
  __device__ __forceinline__ T get(int slot, uint32_t off = 0) const {
    const uint32_t b = typed_slot_offset(fmt_of(FORMAT), slot) + off, wi = b >> 2, sh = (b & 3u) * 8u;
    const uint64_t lo = r[wi], mid = r[wi + 1], hi = r[wi + 2];
    uint64_t v = lo | (mid << 32);
    if (sh != 0) v = (v >> sh) | (hi << (64 - sh));
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, v);
    else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (uint32_t)v);
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (uint16_t)v);
    else return __builtin_bit_cast(T, (uint8_t)v);
  }
};

// One point, one lane: typed attributes read where they live (any stride), record assembled at `rec` in LDS.
template <int FORMAT, typename Src>
__device__ __forceinline__ void encode_point(const EncodeArgs& a, const Src& src, lptr_t rec, double (&mn)[3], double (&mx)[3], unsigned int* hist) {
  constexpr Fmt F = fmt_of(FORMAT);
  int s = 0;       // typed slot cursor (LasPointFormatN field order, las_types.rs)
  uint32_t o = 0;  // raw record cursor
  // position: write_position_as_las_position, write_helpers.rs:10-23
  {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w = src.template get<double>(s, 8 * c);
      const double num = w - a.offset[c];  // two roundings, like the Rust expression
      const double local = a.fast_div ? quotient_by_reciprocal(num, a.scale[c], a.rscale[c]) : num / a.scale[c];
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
  store_un<uint16_t>(rec + o, src.template get<uint16_t>(s)); o += 2; s += 1;  // intensity
  {
    const uint32_t rn = src.template get<uint8_t>(s), nr = src.template get<uint8_t>(s + 1);
    s += 2;
    if (rn >= 1 && rn <= a.max_return) atomicAdd(&hist[rn], 1u);  // points_by_return.get_mut(&return_number)
    if constexpr (F.ext) {
      const uint32_t cf = src.template get<uint8_t>(s), sc = src.template get<uint8_t>(s + 1), sd = src.template get<uint8_t>(s + 2),
                     eof = src.template get<uint8_t>(s + 3);
      s += 4;
      store_un<uint8_t>(rec + o, (uint8_t)((rn & 15u) | ((nr & 15u) << 4)));
      store_un<uint8_t>(rec + o + 1, (uint8_t)((cf & 15u) | ((sc & 3u) << 4) | ((sd & 1u) << 6) | ((eof & 1u) << 7)));
      o += 2;
    } else {
      const uint32_t sd = src.template get<uint8_t>(s), eof = src.template get<uint8_t>(s + 1);
      s += 2;
      store_un<uint8_t>(rec + o, (uint8_t)((rn & 7u) | ((nr & 7u) << 3) | ((sd & 1u) << 6) | ((eof & 1u) << 7)));
      o += 1;
    }
  }
  store_un<uint8_t>(rec + o, src.template get<uint8_t>(s)); o += 1; s += 1;  // classification
  if constexpr (F.ext) {
    store_un<uint8_t>(rec + o, src.template get<uint8_t>(s)); o += 1; s += 1;    // user data
    store_un<int16_t>(rec + o, src.template get<int16_t>(s)); o += 2; s += 1;    // scan angle
  } else {
    store_un<int8_t>(rec + o, src.template get<int8_t>(s)); o += 1; s += 1;      // scan angle rank
    store_un<uint8_t>(rec + o, src.template get<uint8_t>(s)); o += 1; s += 1;    // user data
  }
  store_un<uint16_t>(rec + o, src.template get<uint16_t>(s)); o += 2; s += 1;  // point source id
  if constexpr (F.gps) { store_un<double>(rec + o, src.template get<double>(s)); o += 8; s += 1; }
  if constexpr (F.color) {
#pragma unroll
    for (int c = 0; c < 3; ++c) store_un<uint16_t>(rec + o + 2 * c, src.template get<uint16_t>(s, 2 * c));
    o += 6; s += 1;
  }
  if constexpr (F.nir) { store_un<uint16_t>(rec + o, src.template get<uint16_t>(s)); o += 2; s += 1; }
  if constexpr (F.wave) {
    store_un<uint8_t>(rec + o, src.template get<uint8_t>(s)); o += 1; s += 1;
    store_un<uint64_t>(rec + o, src.template get<uint64_t>(s)); o += 8; s += 1;
    store_un<uint32_t>(rec + o, src.template get<uint32_t>(s)); o += 4; s += 1;
    store_un<float>(rec + o, src.template get<float>(s)); o += 4; s += 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) store_un<float>(rec + o + 4 * c, src.template get<float>(s, 4 * c));
    o += 12; s += 1;
  }
}

constexpr uint32_t kQuadTile = 4 * kBlock;  // points per tile of the columnar path
constexpr int kResidentEncode = 0;  // workgroups resident per CU (kernels.hpp lds_with_resident_cap; 0 = whatever fits)

// One full tile of kQuadTile points starting at `first`; records staged at lds + mis.
template <int FORMAT>
__device__ __forceinline__ void encode_quad_tile(const EncodeArgs& a, uint64_t first, lptr_t lds, uint32_t mis, double (&mn)[3], double (&mx)[3],
                                                 unsigned int* hist) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t RS = raw_size(F);
  const uint32_t tid = threadIdx.x;
  const uint64_t p0 = first + 4u * tid;  // this lane's four points (byte / short / tail columns)
  int s = 0;
  // ---- issue every load of the tile before the first use ----
  // positions: the tile's 3 * kQuadTile doubles as 16-byte chunks, lane-contiguous (chunk = tid + kBlock * j)
  u32x4 pc[6];
  {
    cgptr_t pb = (cgptr_t)(as_global(a.attr_base[0]) + first * 24u);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      pc[j] = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x4>::type*>(pb + 16u * (tid + (uint32_t)kBlock * j)));
    s = 1;
  }
  auto col = [&](int slot, uint32_t bytes) -> cgptr_t { return (cgptr_t)(as_global(a.attr_base[slot]) + p0 * bytes); };
  QuadCol<2> intensity; intensity.load(col(s, 2)); s += 1;
  QuadCol<1> rn, nr, cf, sc, sd, eof, cls, sar, ud;
  QuadCol<2> sa, psid, nir;
  QuadCol<8> gps, woff;
  QuadCol<6> color;
  QuadCol<1> widx;
  QuadCol<4> wsize, wloc;
  QuadCol<12> wpar;
  rn.load(col(s, 1)); nr.load(col(s + 1, 1)); s += 2;
  if constexpr (F.ext) { cf.load(col(s, 1)); sc.load(col(s + 1, 1)); s += 2; }
  sd.load(col(s, 1)); eof.load(col(s + 1, 1)); cls.load(col(s + 2, 1)); s += 3;
  if constexpr (F.ext) { ud.load(col(s, 1)); sa.load(col(s + 1, 2)); s += 2; }
  else { sar.load(col(s, 1)); ud.load(col(s + 1, 1)); s += 2; }
  psid.load(col(s, 2)); s += 1;
  if constexpr (F.gps) { gps.load(col(s, 8)); s += 1; }
  if constexpr (F.color) { color.load(col(s, 6)); s += 1; }
  if constexpr (F.nir) { nir.load(col(s, 2)); s += 1; }
  if constexpr (F.wave) {
    widx.load(col(s, 1)); woff.load(col(s + 1, 8)); wsize.load(col(s + 2, 4)); wloc.load(col(s + 3, 4)); wpar.load(col(s + 4, 12));
    s += 5;
  }

  // ---- positions: double d = 2*tid + 512*j + e of the tile belongs to point d/3, component d%3 ----
  // c0 = (2*tid) % 3 is fixed per lane, so the component of (j, e) is (c0 + (512*j + e) % 3) % 3: accumulators and
  // scale/offset are kept "rotated by c0" and indexed with compile-time r; the rotation is undone once at the end.
  {
    const uint32_t d0 = 2u * tid, q0 = d0 / 3u, c0 = d0 - 3u * q0;
    double sc_r[3], rc_r[3], of_r[3], rmn[3], rmx[3];
#pragma unroll
    for (uint32_t r = 0; r < 3; ++r) {
      const uint32_t c = c0 + r >= 3u ? c0 + r - 3u : c0 + r;
      sc_r[r] = pick3(c, a.scale[0], a.scale[1], a.scale[2]);
      rc_r[r] = pick3(c, a.rscale[0], a.rscale[1], a.rscale[2]);
      of_r[r] = pick3(c, a.offset[0], a.offset[1], a.offset[2]);
      rmn[r] = pick3(c, mn[0], mn[1], mn[2]);
      rmx[r] = pick3(c, mx[0], mx[1], mx[2]);
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t k = 512u * j + e, A = k / 3u, r = k % 3u;  // compile-time
        const uint64_t bits = (uint64_t)(e ? pc[j].z : pc[j].x) | ((uint64_t)(e ? pc[j].w : pc[j].y) << 32);
        const double w = __builtin_bit_cast(double, bits);
        const double num = w - of_r[r];
        const double local = a.fast_div ? quotient_by_reciprocal(num, sc_r[r], rc_r[r]) : num / sc_r[r];
        if (local >= 2147483648.0 || local <= -2147483649.0) bad = true;  // == (local as i64) does not fit an i32 (NaN -> 0 fits)
        const int32_t v = rust_as<int32_t, double>(local);
        const bool wrap = c0 + r >= 3u;
        const uint32_t c = wrap ? c0 + r - 3u : c0 + r, q = q0 + A + (wrap ? 1u : 0u);
        store_un<int32_t>(lds + (mis + q * RS + 4u * c), v);
        rmn[r] = __builtin_fmin(rmn[r], w);
        rmx[r] = __builtin_fmax(rmx[r], w);
      }
    }
    if (bad) atomicAdd(&hist[0], 1u);
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
      const uint32_t r = c >= c0 ? c - c0 : c + 3u - c0;
      mn[c] = pick3(r, rmn[0], rmn[1], rmn[2]);
      mx[c] = pick3(r, rmx[0], rmx[1], rmx[2]);
    }
  }

  // ---- flags (write_las_bit_attributes, write_helpers.rs:32-49) for four points at once, then the record tails ----
  QuadCol<1> f0, f1;
  if constexpr (F.ext) {
    f0.w[1] = f0.w[2] = f1.w[1] = f1.w[2] = 0;
    f0.w[0] = (rn.w[0] & 0x0F0F0F0Fu) | ((nr.w[0] & 0x0F0F0F0Fu) << 4);
    f1.w[0] = (cf.w[0] & 0x0F0F0F0Fu) | ((sc.w[0] & 0x03030303u) << 4) | ((sd.w[0] & 0x01010101u) << 6) | ((eof.w[0] & 0x01010101u) << 7);
  } else {
    f0.w[1] = f0.w[2] = 0;
    f0.w[0] = (rn.w[0] & 0x07070707u) | ((nr.w[0] & 0x07070707u) << 3) | ((sd.w[0] & 0x01010101u) << 6) | ((eof.w[0] & 0x01010101u) << 7);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t r = (rn.w[0] >> (8 * t)) & 255u;
    if (r >= 1 && r <= a.max_return) atomicAdd(&hist[r], 1u);  // points_by_return.get_mut(&return_number)
    RecTail<RS - 12> rec;
    int o = 0;
    rec.put(o, 2, intensity.value(t)); o += 2;
    rec.put(o, 1, f0.value(t)); o += 1;
    if constexpr (F.ext) { rec.put(o, 1, f1.value(t)); o += 1; }
    rec.put(o, 1, cls.value(t)); o += 1;
    if constexpr (F.ext) { rec.put(o, 1, ud.value(t)); rec.put(o + 1, 2, sa.value(t)); o += 3; }
    else { rec.put(o, 1, sar.value(t)); rec.put(o + 1, 1, ud.value(t)); o += 2; }
    rec.put(o, 2, psid.value(t)); o += 2;
    if constexpr (F.gps) { rec.put(o, 8, gps.value(t)); o += 8; }
    if constexpr (F.color) { rec.put(o, 6, color.value(t)); o += 6; }
    if constexpr (F.nir) { rec.put(o, 2, nir.value(t)); o += 2; }
    if constexpr (F.wave) {
      rec.put(o, 1, widx.value(t)); rec.put(o + 1, 8, woff.value(t)); rec.put(o + 9, 4, wsize.value(t)); rec.put(o + 13, 4, wloc.value(t));
      rec.put(o + 17, 8, wpar.bytes_at(12 * t)); rec.put(o + 25, 4, wpar.bytes_at(12 * t + 8) & 0xFFFFFFFFull);
      o += 29;
    }
    rec.store(lds + (mis + (4u * tid + t) * RS + 12u));
  }
}

// MODE_QUAD: every typed attribute is a dense column (stride == element size): full tiles take the four-points-per-lane path.
// MODE_STAGED: interleaved typed records: the tile's source bytes are staged in LDS with LDS-DMA, then read per point.
// MODE_STRIDED: any per-attribute base / stride, read from HBM per point.
enum { MODE_STRIDED = 0, MODE_QUAD = 1, MODE_STAGED = 2 };

__device__ __forceinline__ uint32_t round_up16(uint32_t v) { return (v + 15u) & ~15u; }

template <int FORMAT, int MODE>
__global__ __launch_bounds__(kBlock) void las_encode_kernel(const EncodeArgs a) {
  constexpr Fmt F = fmt_of(FORMAT);
  constexpr uint32_t RS = raw_size(F);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  __shared__ unsigned int hist[kReturnSlots];
  if (threadIdx.x < kReturnSlots) hist[threadIdx.x] = 0;
  double mn[3] = {a.seed_min[0], a.seed_min[1], a.seed_min[2]}, mx[3] = {a.seed_max[0], a.seed_max[1], a.seed_max[2]};
  __syncthreads();

  const uint32_t TS = a.attr_stride[0];                     // MODE_STAGED: typed record size
  lptr_t lds_src = lds + round_up16(a.tile * RS + 16u);     // MODE_STAGED: staged source records
  const uint64_t n_tiles = (a.n + a.tile - 1) / a.tile;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * a.tile;
    const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
    const uint64_t ga = a.dst + first * RS;
    const uint32_t mis = (uint32_t)(ga & 15u);
    if (MODE == MODE_QUAD && cnt == kQuadTile) {
      encode_quad_tile<FORMAT>(a, first, lds, mis, mn, mx, hist);
    } else if (MODE == MODE_STAGED) {
      const uint64_t sa = a.attr_base[0] + first * TS;
      const uint32_t smis = (uint32_t)(sa & 15u);
      tile_load<kBlock>(lds_src, as_global(sa - smis), round_up16(smis + cnt * TS));
      wait_tile_loads();
      __syncthreads();
      for (uint32_t lp = threadIdx.x; lp < cnt; lp += kBlock)
        encode_point<FORMAT>(a, RecordSrc<FORMAT>(lds_src + (smis + lp * TS)), lds + (mis + lp * RS), mn, mx, hist);
    } else {
      for (uint32_t lp = threadIdx.x; lp < cnt; lp += kBlock)
        encode_point<FORMAT>(a, GlobalSrc{a, first + lp}, lds + (mis + lp * RS), mn, mx, hist);
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
  if (threadIdx.x < kReturnSlots) a.partial_counts[(uint64_t)blockIdx.x * kReturnSlots + threadIdx.x] = hist[threadIdx.x];
}

// Folds partials [0, n_in) into one record per block (block b takes b, b + gridDim.x, ...).
__global__ __launch_bounds__(kBlock) void las_encode_fold_kernel(const double* __restrict__ partial_bounds,
                                                                 const unsigned long long* __restrict__ partial_counts, uint32_t n_in,
                                                                 double* __restrict__ out_bounds, unsigned long long* __restrict__ out_counts,
                                                                 double s0, double s1, double s2, double t0, double t1, double t2) {
  double mn[3] = {s0, s1, s2}, mx[3] = {t0, t1, t2};
  for (uint32_t b = blockIdx.x + gridDim.x * threadIdx.x; b < n_in; b += gridDim.x * kBlock) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = __builtin_fmin(mn[c], partial_bounds[(uint64_t)b * 6 + c]);
      mx[c] = __builtin_fmax(mx[c], partial_bounds[(uint64_t)b * 6 + 3 + c]);
    }
  }
  __shared__ double scratch[(kBlock / 64) * 6];
  block_reduce_minmax<double, 3>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* o = out_bounds + (uint64_t)blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
  __shared__ unsigned long long csum[kReturnSlots];
  if (threadIdx.x < kReturnSlots) csum[threadIdx.x] = 0;
  __syncthreads();
  // 16 lanes per counter row: lane l handles counter (l & 15) of this block's rows l/16, l/16 + 16, ...
  unsigned long long local = 0;
  for (uint32_t b = blockIdx.x + gridDim.x * (threadIdx.x >> 4); b < n_in; b += gridDim.x * (kBlock / 16))
    local += partial_counts[(uint64_t)b * kReturnSlots + (threadIdx.x & 15u)];
  atomicAdd(&csum[threadIdx.x & 15u], local);
  __syncthreads();
  if (threadIdx.x < kReturnSlots) out_counts[(uint64_t)blockIdx.x * kReturnSlots + threadIdx.x] = csum[threadIdx.x];
}

}  // namespace

namespace pstk {

uint32_t las_raw_record_size(int format) { return raw_size(fmt_of(format)); }

// attr_base / attr_stride: typed attributes in LasPointFormatN field order.  out_bounds (6 doubles) and out_counts (16 u64)
// are device-accessible.  workspace must hold las_encode_workspace_bytes().
constexpr uint32_t kMaxGrid = 16384, kFoldGrid = 64;
// PST_LAS_RECIPROCAL_DIV=1 selects quotient_by_reciprocal (read per launch: the parity test runs both forms in one process).  OFF by default:
// measured on the quad path (10^8 LAS-0 points, 8 interleaved pairs, profiles/r05_abab.txt) the two forms are indistinguishable -- 0.941 ms with
// the division sequence, 0.967 ms with the reciprocal, IQRs overlapping --: the encoder is bound by LDS / vector-memory issue (r04_sq_cycles.txt),
// not by its vector arithmetic, so the form that needs no proof stays the default.
static bool las_encode_fast_div_allowed() {
  const char* e = std::getenv("PST_LAS_RECIPROCAL_DIV");
  return e && e[0] == '1';
}
size_t las_encode_workspace_bytes() { return (size_t)(kMaxGrid + kFoldGrid) * (6 * sizeof(double) + kReturnSlots * sizeof(unsigned long long)); }

bool launch_las_encode(int format, const uint64_t* attr_base, const uint32_t* attr_stride, const uint32_t* attr_size, int n_attrs, bool interleaved,
                       uint64_t dst, uint64_t n, const double scale[3], const double offset[3], const double bounds_in[6], uint32_t max_return, uint8_t* workspace,
                       double* out_bounds, unsigned long long* out_counts, hipStream_t stream) {
  EncodeArgs a{};
  bool dense = true;
  for (int i = 0; i < n_attrs && i < kMaxAttrs; ++i) {
    a.attr_base[i] = attr_base[i];
    a.attr_stride[i] = attr_stride[i];
    dense = dense && attr_stride[i] == attr_size[i];
  }
  a.dst = dst;
  a.n = n;
  for (int c = 0; c < 3; ++c) { a.scale[c] = scale[c]; a.offset[c] = offset[c]; a.seed_min[c] = bounds_in[c]; a.seed_max[c] = bounds_in[3 + c]; }
  a.max_return = max_return;
  a.fast_div = las_encode_fast_div_allowed() ? 1u : 0u;
  for (int c = 0; c < 3; ++c) {
    uint64_t bits; std::memcpy(&bits, &scale[c], 8);
    const double m = std::fabs(scale[c]);
    if (!(std::isfinite(m) && m >= 0x1p-400 && m <= 0x1p400) || (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) a.fast_div = 0;
    a.rscale[c] = 1.0 / scale[c];
  }
  const uint32_t rs = las_raw_record_size(format);
  const uint32_t ts = attr_stride[0];
  // columnar sources: tiles of kQuadTile points (four per lane); interleaved: source + records in <= 48 KiB; otherwise ~32 KiB of records
  const int mode = dense ? MODE_QUAD : (interleaved ? MODE_STAGED : MODE_STRIDED);
  if (mode == MODE_QUAD) a.tile = kQuadTile;
  else if (mode == MODE_STAGED) a.tile = std::max<uint32_t>(kBlock, ((48u * 1024u) / (rs + ts)) / kBlock * kBlock);
  else a.tile = std::max<uint32_t>(kBlock, ((32u * 1024u) / rs) / kBlock * kBlock);
  const uint64_t n_tiles = std::max<uint64_t>(1, (n + a.tile - 1) / a.tile);
  const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, kMaxGrid);
  double* pb = (double*)workspace;
  unsigned long long* pc = (unsigned long long*)(workspace + (size_t)(kMaxGrid + kFoldGrid) * 6 * sizeof(double));
  a.partial_bounds = pb;
  a.partial_counts = pc;
  const size_t lds_bytes = lds_with_resident_cap((((size_t)a.tile * rs + 16 + 15) & ~(size_t)15) + (mode == MODE_STAGED ? (size_t)a.tile * ts + 48 : 16), kResidentEncode);
#define PST_ENC_MODE(N, M)                                                                                                         \
  {                                                                                                                                 \
    static const hipError_t attr = hipFuncSetAttribute((const void*)las_encode_kernel<N, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    (void)attr;                                                                                                                     \
    hipLaunchKernelGGL((las_encode_kernel<N, M>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);                                 \
  }
#define PST_ENC(N)                                                                                                                  \
  case N:                                                                                                                           \
    if (mode == MODE_QUAD) PST_ENC_MODE(N, MODE_QUAD)                                                                               \
    else if (mode == MODE_STAGED) PST_ENC_MODE(N, MODE_STAGED)                                                                      \
    else PST_ENC_MODE(N, MODE_STRIDED)                                                                                              \
    break;
  switch (format) {
    PST_ENC(0) PST_ENC(1) PST_ENC(2) PST_ENC(3) PST_ENC(4) PST_ENC(5) PST_ENC(6) PST_ENC(7) PST_ENC(8) PST_ENC(9) PST_ENC(10)
    default: return false;
  }
#undef PST_ENC
#undef PST_ENC_MODE
  uint32_t n_in = grid;
  const double* in_b = pb;
  const unsigned long long* in_c = pc;
  if (n_in > 4 * kFoldGrid) {  // two-level fold: 64 blocks first
    hipLaunchKernelGGL(las_encode_fold_kernel, dim3(kFoldGrid), dim3(kBlock), 0, stream, in_b, in_c, n_in, pb + (size_t)kMaxGrid * 6,
                       pc + (size_t)kMaxGrid * kReturnSlots, bounds_in[0], bounds_in[1], bounds_in[2], bounds_in[3], bounds_in[4], bounds_in[5]);
    in_b = pb + (size_t)kMaxGrid * 6;
    in_c = pc + (size_t)kMaxGrid * kReturnSlots;
    n_in = kFoldGrid;
  }
  hipLaunchKernelGGL(las_encode_fold_kernel, dim3(1), dim3(kBlock), 0, stream, in_b, in_c, n_in, out_bounds, out_counts, bounds_in[0], bounds_in[1],
                     bounds_in[2], bounds_in[3], bounds_in[4], bounds_in[5]);
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
