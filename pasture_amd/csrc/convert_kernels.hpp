// Device code of the generic conversion kernels (see convert.hip for the description).  Kept in a header so that the three tile
// kernels and the direct kernels compile as separate translation units (convert.hip, convert_tile_{ft,tf,tt}.hip): each
// instantiates ~100 (source type, target type) bodies, and one TU took six minutes.
#pragma once
#include "device_common.hpp"
#include "kernels.hpp"
#include "tile_io.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <mutex>

using namespace pstd;

namespace {

__device__ __forceinline__ uint32_t round_up16(uint32_t v) { return (v + 15u) & ~15u; }

struct XfRegs {  // transformation parameters held in (scalar) registers for the duration of one mapping
  uint32_t kind, pre, shift;
  uint64_t mask;
  double s0, s1, s2, o0, o1, o2;
};
__device__ __forceinline__ XfRegs load_xf(const PlanEntry& e) {
  XfRegs x;
  x.kind = e.xf_kind; x.pre = e.xf_on_source; x.shift = e.shift; x.mask = e.mask;
  x.s0 = e.scale[0]; x.s1 = e.scale[1]; x.s2 = e.scale[2];
  x.o0 = e.offset[0]; x.o1 = e.offset[1]; x.o2 = e.offset[2];
  return x;
}

// per-thread AABB accumulators of the launch (only touched by `.bounds` entries)
struct BoundsAcc {
  double mn0, mn1, mn2, mx0, mx1, mx2;
  __device__ __forceinline__ void init() { mn0 = mn1 = mn2 = kF64Max; mx0 = mx1 = mx2 = -kF64Max; }
  __device__ __forceinline__ void fold2(uint32_t c, double lo, double hi) {
    mn0 = __builtin_fmin(mn0, c == 0 ? lo : kF64Max);  mx0 = __builtin_fmax(mx0, c == 0 ? hi : -kF64Max);
    mn1 = __builtin_fmin(mn1, c == 1 ? lo : kF64Max);  mx1 = __builtin_fmax(mx1, c == 1 ? hi : -kF64Max);
    mn2 = __builtin_fmin(mn2, c == 2 ? lo : kF64Max);  mx2 = __builtin_fmax(mx2, c == 2 ? hi : -kF64Max);
  }
  __device__ __forceinline__ void fold(uint32_t c, double v) {
    mn0 = __builtin_fmin(mn0, c == 0 ? v : kF64Max);  mx0 = __builtin_fmax(mx0, c == 0 ? v : -kF64Max);
    mn1 = __builtin_fmin(mn1, c == 1 ? v : kF64Max);  mx1 = __builtin_fmax(mx1, c == 1 ? v : -kF64Max);
    mn2 = __builtin_fmin(mn2, c == 2 ? v : kF64Max);  mx2 = __builtin_fmax(mx2, c == 2 ? v : -kF64Max);
  }
};

// One value: optional pre-transform, `as` D, optional post-transform (buffer_conversion.rs:446-456)
template <typename S, typename D>
__device__ __forceinline__ D convert_value_sc(S v, const XfRegs& x, double sc, double of) {
  if (x.kind == 0) return rust_as<D, S>(v);  // wave-uniform: the common untransformed mapping pays nothing
  if (x.pre != 0) v = apply_xf<S>(v, x.kind, sc, of, x.shift, x.mask);
  D w = rust_as<D, S>(v);
  if (x.pre == 0) w = apply_xf<D>(w, x.kind, sc, of, x.shift, x.mask);
  return w;
}
template <typename S, typename D>
__device__ __forceinline__ D convert_value(S v, const XfRegs& x, uint32_t c) {
  if (x.kind == 0) return rust_as<D, S>(v);
  return convert_value_sc<S, D>(v, x, pick3(c, x.s0, x.s1, x.s2), pick3(c, x.o0, x.o1, x.o2));
}

// Split a flat component index into (point, component).
__device__ __forceinline__ void split_comp(uint64_t k, uint32_t ncomp, uint64_t& p, uint32_t& c) {
  if (ncomp == 1) { p = k; c = 0; }
  else if (ncomp == 3) { p = k / 3; c = (uint32_t)(k - 3 * p); }
  else { p = k / ncomp; c = (uint32_t)(k - p * ncomp); }
}
__device__ __forceinline__ void split_comp32(uint32_t k, uint32_t ncomp, uint32_t& p, uint32_t& c) {
  if (ncomp == 1) { p = k; c = 0; }
  else if (ncomp == 3) { p = k / 3; c = k - 3 * p; }
  else { p = k / ncomp; c = k - p * ncomp; }
}

// wave-uniform fetch of one plan entry through the constant address space (scalar loads)
__device__ __forceinline__ PlanEntry fetch_entry(const PlanEntry* entries, uint32_t m) {
  static_assert(sizeof(PlanEntry) % 4 == 0, "PlanEntry must be a whole number of dwords");
  const PST_AS_CONST uint32_t* w = (const PST_AS_CONST uint32_t*)(entries + m);
  PlanEntry e;
  uint32_t* d = reinterpret_cast<uint32_t*>(&e);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(PlanEntry) / 4; ++i) d[i] = w[i];
  return e;
}

template <typename T> struct ChunkOf { static constexpr uint32_t value = sizeof(T) >= 4 ? 1u : 4u / (uint32_t)sizeof(T); };

// block-level write-out of the fused AABB record
template <int BLK>
__device__ __forceinline__ void flush_bounds(BoundsAcc& acc, uint64_t partials_addr) {
  if (partials_addr == 0) return;
  __shared__ double scratch[(BLK / 64) * 6];
  double mn[3] = {acc.mn0, acc.mn1, acc.mn2}, mx[3] = {acc.mx0, acc.mx1, acc.mx2};
  block_reduce_minmax<double, 3, BLK>(mn, mx, scratch);
  if (threadIdx.x == 0) {
    double* out = (double*)partials_addr + (uint64_t)blockIdx.x * 6;
    out[0] = mn[0]; out[1] = mn[1]; out[2] = mn[2];
    out[3] = mx[0]; out[4] = mx[1]; out[5] = mx[2];
  }
}

// ------------------------------------------------------------------------------------------------------
// direct kernel
// ------------------------------------------------------------------------------------------------------
template <bool SRC_AOS, bool DST_AOS, typename S, typename D>
__device__ __forceinline__ void run_direct(const ConvertHeader& h, const PlanEntry& e, BoundsAcc& acc) {
  const uint64_t total = h.n * e.ncomp;
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  const XfRegs x = load_xf(e);
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < total; k += step) {
    uint64_t p;
    uint32_t c;
    split_comp(k, e.ncomp, p, c);
    cgptr_t sp;
    gptr_t dp;
    if constexpr (SRC_AOS) sp = as_global(h.src_aos) + p * h.src_stride + e.src_off + c * sizeof(S);
    else sp = as_global(e.src_col) + k * sizeof(S);
    if constexpr (DST_AOS) dp = as_global(h.dst_aos) + p * h.dst_stride + e.dst_off + c * sizeof(D);
    else dp = as_global(e.dst_col) + k * sizeof(D);
    const D w = convert_value<S, D>(load_un<S>(sp), x, c);
    store_un<D>(dp, w);
    if constexpr (std::is_same<D, double>::value) {
      if (e.bounds) acc.fold(c, w);
    }
  }
}

template <bool SRC_AOS, bool DST_AOS>
__global__ __launch_bounds__(kBlock) void convert_direct_kernel(const ConvertHeader h, const PlanEntry* __restrict__ entries) {
  BoundsAcc acc;
  acc.init();
  for (uint32_t m = 0; m < h.n_entries; ++m) {
    const PlanEntry e = fetch_entry(entries, m);
    dispatch_ct(e.src_ct, [&](auto s) __attribute__((always_inline)) {
      using S = decltype(s);
      if (!e.convert) {
        run_direct<SRC_AOS, DST_AOS, S, S>(h, e, acc);
      } else {
        dispatch_ct(e.dst_ct, [&](auto d) __attribute__((always_inline)) { run_direct<SRC_AOS, DST_AOS, S, decltype(d)>(h, e, acc); });
      }
    });
  }
  flush_bounds<kBlock>(acc, h.bounds_partials);
}

// ------------------------------------------------------------------------------------------------------
// LDS-tiled kernel
// ------------------------------------------------------------------------------------------------------
// Which lanes work on an entry: all BLK lanes of the block, or (small entries) the 64 lanes of the one wave that owns it.
struct LaneSpan { uint32_t first, step; };

// NE values of D as ONE vector store of 4 / 8 / 16 bytes (unaligned column starts are fine: range offsets are arbitrary)
template <typename D, uint32_t NE>
__device__ __forceinline__ void store_vec(gptr_t dst, const D (&v)[NE]) {
  constexpr uint32_t B = (uint32_t)sizeof(D) * NE;
  static_assert(B == 4 || B == 8 || B == 16, "one dword / dwordx2 / dwordx4 store");
  if constexpr (B == 16) { u32x4 w; __builtin_memcpy(&w, v, 16); store_un<u32x4>(dst, w); }
  else if constexpr (B == 8) { uint64_t w; __builtin_memcpy(&w, v, 8); store_un<uint64_t>(dst, w); }
  else { uint32_t w; __builtin_memcpy(&w, v, 4); store_un<uint32_t>(dst, w); }
}

// interleaved (LDS) -> columnar (global): each lane produces 16 bytes of the column per iteration (one dwordx4 store =
// 1 KiB per wave instruction) from NE = 16 / sizeof(D) strided LDS reads
template <typename S, typename D>
__device__ __forceinline__ void run_tile_to_column(const ConvertHeader& h, const PlanEntry& e, clptr_t lds_src, uint64_t first, uint32_t cnt,
                                                   LaneSpan span, BoundsAcc& acc) {
  constexpr uint32_t E = ChunkOf<D>::value;
  constexpr uint32_t NS = 16u / (uint32_t)sizeof(D);  // scalar attributes: values per lane and iteration (16 bytes)
  constexpr uint32_t NE = NS < 4u ? NS : 4u;          // Vec3 attributes: at most four values (bounded register arrays)
  const uint32_t total = cnt * e.ncomp;
  const XfRegs x = load_xf(e);
  gptr_t col = as_global(e.dst_col) + first * e.ncomp * sizeof(D);
  if (e.ncomp == 3) {
    // Vec3 values: lanes - lanes%3 lanes work, so lane*NE and the per-iteration advance are multiples of... the advance
    // (lanes*NE values) is a multiple of 3: the component of a lane's j-th value, (c0 + j) % 3 with c0 = (lane*NE) % 3, NEVER
    // changes — no division in the loop, scale / offset / AABB slot per (lane, j) are loop constants, the LDS and column
    // addresses advance by constants.  (255 of 256 lanes work on a shared entry, 63 of 64 on a wave-owned one.)
    const uint32_t lanes = span.step - span.step % 3u;
    if (span.first < lanes) {
      const uint32_t k_lane = span.first * NE, p_lane = k_lane / 3u, c0 = k_lane - 3u * p_lane;
      const uint32_t k_step = lanes * NE, p_step = k_step / 3u;
      uint32_t la[NE];   // LDS byte offset of value j of this lane in iteration 0
      double sc[NE], of[NE];
#pragma unroll
      for (uint32_t j = 0; j < NE; ++j) {
        const uint32_t cj = (c0 + j) % 3u, pj = p_lane + (c0 + j) / 3u;
        la[j] = pj * h.src_stride + e.src_off + cj * (uint32_t)sizeof(S);
        sc[j] = pick3(cj, x.s0, x.s1, x.s2);
        of[j] = pick3(cj, x.o0, x.o1, x.o2);
      }
      const uint32_t la_step = p_step * h.src_stride;
      double lo[NE], hi[NE];
#pragma unroll
      for (uint32_t j = 0; j < NE; ++j) { lo[j] = kF64Max; hi[j] = -kF64Max; }
      uint32_t k = k_lane, lofs = 0;
      for (; k + NE <= total; k += k_step, lofs += la_step) {
        S v[NE];
#pragma unroll
        for (uint32_t j = 0; j < NE; ++j) v[j] = lds_load<S>(lds_src + (la[j] + lofs));
        D w[NE];
#pragma unroll
        for (uint32_t j = 0; j < NE; ++j) {
          w[j] = convert_value_sc<S, D>(v[j], x, sc[j], of[j]);
          if constexpr (std::is_same<D, double>::value) {
            lo[j] = __builtin_fmin(lo[j], w[j]);
            hi[j] = __builtin_fmax(hi[j], w[j]);
          }
        }
        store_vec<D, NE>(col + (uint64_t)k * sizeof(D), w);
      }
      if (k < total) {  // ragged end of the tile: fewer than NE values left for this lane
#pragma unroll
        for (uint32_t j = 0; j < NE; ++j) {
          if (k + j < total) {
            const D w = convert_value_sc<S, D>(lds_load<S>(lds_src + (la[j] + lofs)), x, sc[j], of[j]);
            store_un<D>(col + (uint64_t)(k + j) * sizeof(D), w);
            if constexpr (std::is_same<D, double>::value) {
              lo[j] = __builtin_fmin(lo[j], w);
              hi[j] = __builtin_fmax(hi[j], w);
            }
          }
        }
      }
      if constexpr (std::is_same<D, double>::value) {
        if (e.bounds) {
#pragma unroll
          for (uint32_t j = 0; j < NE; ++j) acc.fold2((c0 + j) % 3u, lo[j], hi[j]);
        }
      }
    }
    return;
  }
  if (e.ncomp == 1) {
    // scalar attributes: a lane produces NS consecutive points (16 bytes of the column)
    const uint32_t la_step = span.step * NS * h.src_stride;
    uint32_t la = span.first * NS * h.src_stride + e.src_off;
    uint32_t k0 = span.first * NS;
    for (; k0 + NS <= cnt; k0 += span.step * NS, la += la_step) {
      S v[NS];
#pragma unroll
      for (uint32_t j = 0; j < NS; ++j) v[j] = lds_load<S>(lds_src + (la + j * h.src_stride));
      D w[NS];
#pragma unroll
      for (uint32_t j = 0; j < NS; ++j) w[j] = convert_value_sc<S, D>(v[j], x, x.s0, x.o0);
      store_vec<D, NS>(col + (uint64_t)k0 * sizeof(D), w);
    }
    if (k0 < cnt) {
      for (uint32_t j = 0; k0 + j < cnt; ++j)
        store_un<D>(col + (uint64_t)(k0 + j) * sizeof(D), convert_value_sc<S, D>(lds_load<S>(lds_src + (la + j * h.src_stride)), x, x.s0, x.o0));
    }
    return;
  }
  for (uint32_t q = span.first; q * E < total; q += span.step) {
    const uint32_t k0 = q * E;
    D vals[E];
#pragma unroll
    for (uint32_t i = 0; i < E; ++i) {
      const uint32_t k = k0 + i < total ? k0 + i : total - 1;  // clamp: a ragged tail recomputes the last value, never stores it
      uint32_t p, c;
      split_comp32(k, e.ncomp, p, c);
      vals[i] = convert_value<S, D>(lds_load<S>(lds_src + (p * h.src_stride + e.src_off + c * (uint32_t)sizeof(S))), x, c);
      if constexpr (std::is_same<D, double>::value) {
        if (e.bounds) acc.fold(c, vals[i]);
      }
    }
    if constexpr (E == 1) {
      store_un<D>(col + (uint64_t)k0 * sizeof(D), vals[0]);
    } else {
      if (k0 + E <= total) {
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) {
          typename std::make_unsigned<D>::type u;
          __builtin_memcpy(&u, &vals[i], sizeof(D));
          packed |= (uint32_t)u << (8u * (uint32_t)sizeof(D) * i);
        }
        store_un<uint32_t>(col + (uint64_t)k0 * sizeof(D), packed);
      } else {
        for (uint32_t i = 0; k0 + i < total; ++i) store_un<D>(col + (uint64_t)(k0 + i) * sizeof(D), vals[i]);
      }
    }
  }
}

// columnar (global) -> interleaved (LDS): each lane consumes one >= 4-byte chunk of the column
template <typename S, typename D>
__device__ __forceinline__ void run_tile_from_column(const ConvertHeader& h, const PlanEntry& e, lptr_t lds_dst, uint64_t first, uint32_t cnt,
                                                     LaneSpan span, BoundsAcc& acc) {
  constexpr uint32_t E = ChunkOf<S>::value;
  const uint32_t total = cnt * e.ncomp;
  const XfRegs x = load_xf(e);
  cgptr_t col = as_global(e.src_col) + first * e.ncomp * sizeof(S);
  if constexpr (E == 1) {
    if (e.ncomp == 3) {  // lane-fixed component index, see run_tile_to_column
      const uint32_t lanes = span.step - span.step % 3u;
      if (span.first < lanes) {
        const uint32_t c = span.first % 3u, p0 = span.first / 3u, ppi = lanes / 3u;
        const double sc = pick3(c, x.s0, x.s1, x.s2), of = pick3(c, x.o0, x.o1, x.o2);
        uint32_t la = p0 * h.dst_stride + e.dst_off + c * (uint32_t)sizeof(D);
        uint32_t ga = span.first * (uint32_t)sizeof(S);
        const uint32_t la_step = ppi * h.dst_stride, ga_step = lanes * (uint32_t)sizeof(S);
        double lo = kF64Max, hi = -kF64Max;
        // batches of kBatch global loads in flight per lane (a dependent load -> LDS-store chain per iteration would expose
        // the full HBM latency every time)
        constexpr uint32_t kBatch = 4;
        for (uint32_t pp = p0; pp < cnt; pp += kBatch * ppi, la += kBatch * la_step, ga += kBatch * ga_step) {
          S v[kBatch];
#pragma unroll
          for (uint32_t u = 0; u < kBatch; ++u) v[u] = pp + u * ppi < cnt ? load_un<S>(col + (ga + u * ga_step)) : S{};
#pragma unroll
          for (uint32_t u = 0; u < kBatch; ++u) {
            if (pp + u * ppi < cnt) {
              const D w = convert_value_sc<S, D>(v[u], x, sc, of);
              store_un<D>(lds_dst + (la + u * la_step), w);
              if constexpr (std::is_same<D, double>::value) {
                lo = __builtin_fmin(lo, w);
                hi = __builtin_fmax(hi, w);
              }
            }
          }
        }
        if constexpr (std::is_same<D, double>::value) {
          if (e.bounds) acc.fold2(c, lo, hi);
        }
      }
      return;
    }
  }
  if (e.ncomp == 1) {
    const uint32_t la_step = span.step * E * h.dst_stride, k_step = span.step * E;
    uint32_t la = span.first * E * h.dst_stride + e.dst_off;
    constexpr uint32_t kBatch = 4;
    typedef typename std::conditional<(E > 1), uint32_t, S>::type chunk_t;  // E narrow values travel as one dword
    for (uint32_t k0 = span.first * E; k0 < cnt; k0 += kBatch * k_step, la += kBatch * la_step) {
      chunk_t chunk[kBatch];
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t k = k0 + u * k_step;
        chunk[u] = chunk_t{};
        if (k + E <= cnt) chunk[u] = load_un<chunk_t>(col + (uint64_t)k * sizeof(S));
      }
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t k = k0 + u * k_step, lau = la + u * la_step;
        if (k + E <= cnt) {
          if constexpr (E == 1) {
            store_un<D>(lds_dst + lau, convert_value_sc<S, D>(chunk[u], x, x.s0, x.o0));
          } else {
#pragma unroll
            for (uint32_t i = 0; i < E; ++i) {
              typename std::make_unsigned<S>::type bits = (typename std::make_unsigned<S>::type)(chunk[u] >> (8u * (uint32_t)sizeof(S) * i));
              S v;
              __builtin_memcpy(&v, &bits, sizeof(S));
              store_un<D>(lds_dst + (lau + i * h.dst_stride), convert_value_sc<S, D>(v, x, x.s0, x.o0));
            }
          }
        } else if (k < cnt) {  // ragged tail of the tile
          for (uint32_t i = 0; k + i < cnt; ++i)
            store_un<D>(lds_dst + (lau + i * h.dst_stride), convert_value_sc<S, D>(load_un<S>(col + (uint64_t)(k + i) * sizeof(S)), x, x.s0, x.o0));
        }
      }
    }
    return;
  }
  for (uint32_t q = span.first; q * E < total; q += span.step) {
    const uint32_t k0 = q * E;
    S vals[E];
    if constexpr (E == 1) {
      vals[0] = load_un<S>(col + (uint64_t)k0 * sizeof(S));
    } else {
      if (k0 + E <= total) {
        const uint32_t packed = load_un<uint32_t>(col + (uint64_t)k0 * sizeof(S));
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) {
          typename std::make_unsigned<S>::type u = (typename std::make_unsigned<S>::type)(packed >> (8u * (uint32_t)sizeof(S) * i));
          __builtin_memcpy(&vals[i], &u, sizeof(S));
        }
      } else {
        for (uint32_t i = 0; i < E; ++i) vals[i] = k0 + i < total ? load_un<S>(col + (uint64_t)(k0 + i) * sizeof(S)) : S{};
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < E; ++i) {
      if (k0 + i < total) {
        uint32_t p, c;
        split_comp32(k0 + i, e.ncomp, p, c);
        const D w = convert_value<S, D>(vals[i], x, c);
        store_un<D>(lds_dst + (p * h.dst_stride + e.dst_off + c * (uint32_t)sizeof(D)), w);
        if constexpr (std::is_same<D, double>::value) {
          if (e.bounds) acc.fold(c, w);
        }
      }
    }
  }
}

// ---- columnar (global) -> interleaved (LDS), four consecutive points per lane ------------------------------------
// A lane that owns the points 4q .. 4q+3 reads 4*NC*sizeof(S) contiguous column bytes with vector loads, and the LDS address of
// its i-th record is (tile base + 4q*stride) + i*stride + offset: the alignment class (address & 3) of every store depends only
// on i, so it is wave-uniform and each value is written as head bytes + aligned dwords (re-cut with v_alignbyte) + tail bytes
// behind a SCALAR branch -- no unaligned ds_write (SQ_LDS_UNALIGNED_STALL), no divergence.
template <uint32_t ND>
__device__ __forceinline__ void load_dwords(cgptr_t p, uint32_t (&r)[ND]) {
#pragma unroll
  for (uint32_t i = 0; i + 4 <= ND; i += 4) {
    const u32x4 v = load_un<u32x4>(p + 4u * i);
    r[i] = v.x; r[i + 1] = v.y; r[i + 2] = v.z; r[i + 3] = v.w;
  }
  if constexpr (ND % 4 >= 2) {
    const uint64_t v = load_un<uint64_t>(p + 4u * (ND & ~3u));
    r[ND & ~3u] = (uint32_t)v; r[(ND & ~3u) + 1] = (uint32_t)(v >> 32);
  }
  if constexpr (ND % 2 == 1) r[ND - 1] = load_un<uint32_t>(p + 4u * (ND - 1));
}

template <typename S, uint32_t ND>
__device__ __forceinline__ S quad_elem(const uint32_t (&r)[ND], uint32_t el) {
  if constexpr (sizeof(S) == 8) {
    return __builtin_bit_cast(S, (uint64_t)r[2 * el] | ((uint64_t)r[2 * el + 1] << 32));
  } else if constexpr (sizeof(S) == 4) {
    return __builtin_bit_cast(S, r[el]);
  } else if constexpr (sizeof(S) == 2) {
    return __builtin_bit_cast(S, (uint16_t)(r[el >> 1] >> (16u * (el & 1u))));
  } else {
    return __builtin_bit_cast(S, (uint8_t)(r[el >> 2] >> (8u * (el & 3u))));
  }
}

// L contiguous bytes (little-endian in w[], one spare dword behind them) stored at an LDS address of alignment class CLS
template <uint32_t L, uint32_t CLS>
__device__ __forceinline__ void lds_store_string(lptr_t p, const uint32_t (&w)[(L + 3) / 4 + 1]) {
  typedef PST_AS_LDS uint8_t* p8;
  typedef PST_AS_LDS uint16_t* p16;
  typedef PST_AS_LDS uint32_t* p32;
  constexpr uint32_t H = CLS == 0 ? 0u : ((4u - CLS) < L ? (4u - CLS) : L);  // bytes in front of the first dword boundary
  if constexpr (H == 1) {
    *(p8)p = (uint8_t)w[0];
  } else if constexpr (H == 2) {
    if constexpr (CLS == 2) *(p16)p = (uint16_t)w[0];
    else { *(p8)p = (uint8_t)w[0]; *(p8)(p + 1) = (uint8_t)(w[0] >> 8); }
  } else if constexpr (H == 3) {
    *(p8)p = (uint8_t)w[0];
    *(p16)(p + 1) = (uint16_t)(w[0] >> 8);
  }
  auto cut = [&](uint32_t o) -> uint32_t {  // the dword that starts at byte o of the string
    const uint32_t j = o >> 2, sh = o & 3u;
    return sh == 0 ? w[j] : __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
  };
  constexpr uint32_t NB = (L - H) / 4u, R = (L - H) % 4u;
#pragma unroll
  for (uint32_t k = 0; k < NB; ++k) *(p32)(p + H + 4u * k) = cut(H + 4u * k);
  if constexpr (R != 0) {
    const uint32_t d = cut(H + 4u * NB);
    lptr_t t = p + H + 4u * NB;
    if constexpr (R == 1) *(p8)t = (uint8_t)d;
    else if constexpr (R == 2) *(p16)t = (uint16_t)d;
    else { *(p16)t = (uint16_t)d; *(p8)(t + 2) = (uint8_t)(d >> 16); }
  }
}

template <typename S, typename D, uint32_t NC>
__device__ __forceinline__ void run_tile_from_column_quad(const ConvertHeader& h, const PlanEntry& e, lptr_t lds_dst, uint64_t first, uint32_t cnt,
                                                          LaneSpan span, BoundsAcc& acc) {
  constexpr uint32_t ND = NC * (uint32_t)sizeof(S);  // dwords of the column per quad of points
  constexpr uint32_t L = NC * (uint32_t)sizeof(D);   // bytes of the attribute in a record
  constexpr uint32_t LW = (L + 3) / 4;
  constexpr uint32_t KB = ND <= 2 ? 4u : ND <= 6 ? 2u : 1u;  // quads in flight per lane
  constexpr bool kBounds = std::is_same<D, double>::value && NC == 3;
  const XfRegs x = load_xf(e);
  cgptr_t col = as_global(e.src_col) + first * (NC * sizeof(S));
  const uint32_t quads = cnt >> 2, st = h.dst_stride;
  const uint32_t a0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)lds_dst + e.dst_off));
  const uint32_t cls[4] = {a0 & 3u, (a0 + st) & 3u, (a0 + 2u * st) & 3u, (a0 + 3u * st) & 3u};
  double lo[3] = {kF64Max, kF64Max, kF64Max}, hi[3] = {-kF64Max, -kF64Max, -kF64Max};
  for (uint32_t q0 = span.first; q0 < quads; q0 += KB * span.step) {
    uint32_t r[KB][ND];
#pragma unroll
    for (uint32_t u = 0; u < KB; ++u) {
      const uint32_t q = q0 + u * span.step;
      if (q < quads) load_dwords<ND>(col + (uint64_t)q * (ND * 4u), r[u]);
    }
#pragma unroll
    for (uint32_t u = 0; u < KB; ++u) {
      const uint32_t q = q0 + u * span.step;
      if (q < quads) {
        lptr_t base = lds_dst + (q * 4u * st + e.dst_off);
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
          uint32_t w[LW + 1];
#pragma unroll
          for (uint32_t j = 0; j <= LW; ++j) w[j] = 0;
#pragma unroll
          for (uint32_t c = 0; c < NC; ++c) {
            const S v = quad_elem<S, ND>(r[u], i * NC + c);
            D d;
            if constexpr (NC == 3) d = convert_value_sc<S, D>(v, x, c == 0 ? x.s0 : c == 1 ? x.s1 : x.s2, c == 0 ? x.o0 : c == 1 ? x.o1 : x.o2);
            else if constexpr (NC == 1) d = convert_value_sc<S, D>(v, x, x.s0, x.o0);
            else d = convert_value<S, D>(v, x, c);
            if constexpr (kBounds) {
              if (e.bounds) {  // wave-uniform
                lo[c] = __builtin_fmin(lo[c], d);
                hi[c] = __builtin_fmax(hi[c], d);
              }
            }
            if constexpr (sizeof(D) == 8) {
              const uint64_t b = __builtin_bit_cast(uint64_t, d);
              w[2 * c] = (uint32_t)b; w[2 * c + 1] = (uint32_t)(b >> 32);
            } else if constexpr (sizeof(D) == 4) {
              w[c] = __builtin_bit_cast(uint32_t, d);
            } else if constexpr (sizeof(D) == 2) {
              w[c >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, d) << (16u * (c & 1u));
            } else {
              w[c >> 2] |= (uint32_t)__builtin_bit_cast(uint8_t, d) << (8u * (c & 3u));
            }
          }
          lptr_t p = base + i * st;
          if constexpr (L == 1) {
            lds_store_string<1, 0>(p, w);  // a byte store is aligned wherever it lands
          } else if constexpr (L == 2) {
            if ((cls[i] & 1u) == 0) lds_store_string<2, 0>(p, w);
            else lds_store_string<2, 1>(p, w);
          } else {
            switch (cls[i]) {  // wave-uniform
              case 0: lds_store_string<L, 0>(p, w); break;
              case 1: lds_store_string<L, 1>(p, w); break;
              case 2: lds_store_string<L, 2>(p, w); break;
              default: lds_store_string<L, 3>(p, w); break;
            }
          }
        }
      }
    }
  }
  // the last tile of a range may end inside a quad
  const uint32_t done = quads * 4u, rem = (cnt - done) * NC;
  if (span.first < rem) {
    const uint32_t k = done * NC + span.first, pt = k / NC, c = k - pt * NC;
    const D d = convert_value<S, D>(load_un<S>(col + (uint64_t)k * sizeof(S)), x, c);
    store_un<D>(lds_dst + (pt * st + e.dst_off + c * (uint32_t)sizeof(D)), d);
    if constexpr (kBounds) {
      if (e.bounds) acc.fold(c, d);
    }
  }
  if constexpr (kBounds) {
    if (e.bounds) {
      acc.mn0 = __builtin_fmin(acc.mn0, lo[0]); acc.mx0 = __builtin_fmax(acc.mx0, hi[0]);
      acc.mn1 = __builtin_fmin(acc.mn1, lo[1]); acc.mx1 = __builtin_fmax(acc.mx1, hi[1]);
      acc.mn2 = __builtin_fmin(acc.mn2, lo[2]); acc.mx2 = __builtin_fmax(acc.mx2, hi[2]);
    }
  }
}

template <typename T> struct Vec3Able {
  static constexpr bool value = std::is_same<T, uint8_t>::value || std::is_same<T, uint16_t>::value || std::is_same<T, int32_t>::value ||
                                std::is_same<T, float>::value || std::is_same<T, double>::value;
};

// interleaved (LDS) -> interleaved (LDS)
template <typename S, typename D>
__device__ __forceinline__ void run_tile_lds_to_lds(const ConvertHeader& h, const PlanEntry& e, clptr_t lds_src, lptr_t lds_dst, uint32_t cnt,
                                                    LaneSpan span, BoundsAcc& acc) {
  const uint32_t total = cnt * e.ncomp;
  const XfRegs x = load_xf(e);
  for (uint32_t k = span.first; k < total; k += span.step) {
    uint32_t p, c;
    split_comp32(k, e.ncomp, p, c);
    const D w = convert_value<S, D>(lds_load<S>(lds_src + (p * h.src_stride + e.src_off + c * (uint32_t)sizeof(S))), x, c);
    lds_store<D>(lds_dst + (p * h.dst_stride + e.dst_off + c * (uint32_t)sizeof(D)), w);
    if constexpr (std::is_same<D, double>::value) {
      if (e.bounds) acc.fold(c, w);
    }
  }
}

// L contiguous bytes read from an LDS address of (wave-uniform) alignment class CLS: aligned dwords, re-cut with v_alignbyte
template <uint32_t L, uint32_t CLS>
__device__ __forceinline__ void lds_load_string(clptr_t p, uint32_t (&w)[(L + 3) / 4 + 1]) {
  typedef const PST_AS_LDS uint8_t* p8;
  typedef const PST_AS_LDS uint16_t* p16;
  typedef const PST_AS_LDS uint32_t* p32;
  constexpr uint32_t LW = (L + 3) / 4;
  w[LW] = 0;
  if constexpr (L == 1) {
    w[0] = *(p8)p;
  } else if constexpr (L == 2 && (CLS & 1u) == 0) {
    w[0] = *(p16)p;
  } else {
    constexpr uint32_t NR = (CLS + L + 3) / 4;  // aligned dwords that cover the string
    uint32_t d[NR + 1];
    clptr_t b = p - CLS;
#pragma unroll
    for (uint32_t k = 0; k < NR; ++k) d[k] = *(p32)(b + 4u * k);
    d[NR] = 0;
#pragma unroll
    for (uint32_t j = 0; j < LW; ++j) w[j] = CLS == 0 ? d[j] : __builtin_amdgcn_alignbyte(d[j + 1 < NR ? j + 1 : NR], d[j], CLS);
  }
}

// interleaved (LDS) -> interleaved (LDS), four consecutive records per lane: both alignment classes are wave-uniform
template <typename S, typename D, uint32_t NC>
__device__ __forceinline__ void run_tile_lds_to_lds_quad(const ConvertHeader& h, const PlanEntry& e, clptr_t lds_src, lptr_t lds_dst, uint32_t cnt,
                                                         LaneSpan span, BoundsAcc& acc) {
  constexpr uint32_t LS = NC * (uint32_t)sizeof(S), LD = NC * (uint32_t)sizeof(D);
  constexpr uint32_t LWS = (LS + 3) / 4, LWD = (LD + 3) / 4;
  constexpr bool kBounds = std::is_same<D, double>::value && NC == 3;
  const XfRegs x = load_xf(e);
  const uint32_t quads = cnt >> 2, ss = h.src_stride, ds = h.dst_stride;
  const uint32_t as0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)lds_src + e.src_off));
  const uint32_t ad0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)lds_dst + e.dst_off));
  const uint32_t cs[4] = {as0 & 3u, (as0 + ss) & 3u, (as0 + 2u * ss) & 3u, (as0 + 3u * ss) & 3u};
  const uint32_t cd[4] = {ad0 & 3u, (ad0 + ds) & 3u, (ad0 + 2u * ds) & 3u, (ad0 + 3u * ds) & 3u};
  double lo[3] = {kF64Max, kF64Max, kF64Max}, hi[3] = {-kF64Max, -kF64Max, -kF64Max};
  for (uint32_t q = span.first; q < quads; q += span.step) {
    clptr_t sb = lds_src + (q * 4u * ss + e.src_off);
    lptr_t db = lds_dst + (q * 4u * ds + e.dst_off);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      uint32_t ws[LWS + 1];
      clptr_t sp = sb + i * ss;
      if constexpr (LS == 1) {
        lds_load_string<1, 0>(sp, ws);
      } else if constexpr (LS == 2) {
        if ((cs[i] & 1u) == 0) lds_load_string<2, 0>(sp, ws);
        else if (cs[i] == 1u) lds_load_string<2, 1>(sp, ws);
        else lds_load_string<2, 3>(sp, ws);
      } else {
        switch (cs[i]) {  // wave-uniform
          case 0: lds_load_string<LS, 0>(sp, ws); break;
          case 1: lds_load_string<LS, 1>(sp, ws); break;
          case 2: lds_load_string<LS, 2>(sp, ws); break;
          default: lds_load_string<LS, 3>(sp, ws); break;
        }
      }
      uint32_t w[LWD + 1];
#pragma unroll
      for (uint32_t j = 0; j <= LWD; ++j) w[j] = 0;
#pragma unroll
      for (uint32_t c = 0; c < NC; ++c) {
        const S v = quad_elem<S, LWS + 1>(ws, c);
        D d;
        if constexpr (NC == 3) d = convert_value_sc<S, D>(v, x, c == 0 ? x.s0 : c == 1 ? x.s1 : x.s2, c == 0 ? x.o0 : c == 1 ? x.o1 : x.o2);
        else if constexpr (NC == 1) d = convert_value_sc<S, D>(v, x, x.s0, x.o0);
        else d = convert_value<S, D>(v, x, c);
        if constexpr (kBounds) {
          if (e.bounds) {
            lo[c] = __builtin_fmin(lo[c], d);
            hi[c] = __builtin_fmax(hi[c], d);
          }
        }
        if constexpr (sizeof(D) == 8) {
          const uint64_t b = __builtin_bit_cast(uint64_t, d);
          w[2 * c] = (uint32_t)b; w[2 * c + 1] = (uint32_t)(b >> 32);
        } else if constexpr (sizeof(D) == 4) {
          w[c] = __builtin_bit_cast(uint32_t, d);
        } else if constexpr (sizeof(D) == 2) {
          w[c >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, d) << (16u * (c & 1u));
        } else {
          w[c >> 2] |= (uint32_t)__builtin_bit_cast(uint8_t, d) << (8u * (c & 3u));
        }
      }
      lptr_t dp = db + i * ds;
      if constexpr (LD == 1) {
        lds_store_string<1, 0>(dp, w);
      } else if constexpr (LD == 2) {
        if ((cd[i] & 1u) == 0) lds_store_string<2, 0>(dp, w);
        else lds_store_string<2, 1>(dp, w);
      } else {
        switch (cd[i]) {
          case 0: lds_store_string<LD, 0>(dp, w); break;
          case 1: lds_store_string<LD, 1>(dp, w); break;
          case 2: lds_store_string<LD, 2>(dp, w); break;
          default: lds_store_string<LD, 3>(dp, w); break;
        }
      }
    }
  }
  const uint32_t done = quads * 4u, rem = (cnt - done) * NC;
  if (span.first < rem) {
    const uint32_t k = done * NC + span.first, pt = k / NC, c = k - pt * NC;
    const D d = convert_value<S, D>(lds_load<S>(lds_src + (pt * ss + e.src_off + c * (uint32_t)sizeof(S))), x, c);
    lds_store<D>(lds_dst + (pt * ds + e.dst_off + c * (uint32_t)sizeof(D)), d);
    if constexpr (kBounds) {
      if (e.bounds) acc.fold(c, d);
    }
  }
  if constexpr (kBounds) {
    if (e.bounds) {
      acc.mn0 = __builtin_fmin(acc.mn0, lo[0]); acc.mx0 = __builtin_fmax(acc.mx0, hi[0]);
      acc.mn1 = __builtin_fmin(acc.mn1, lo[1]); acc.mx1 = __builtin_fmax(acc.mx1, hi[1]);
      acc.mn2 = __builtin_fmin(acc.mn2, lo[2]); acc.mx2 = __builtin_fmax(acc.mx2, hi[2]);
    }
  }
}

template <bool SRC_AOS, bool DST_AOS, typename S, typename D>
__device__ __forceinline__ void run_tile(const ConvertHeader& h, const PlanEntry& e, clptr_t lds_src, lptr_t lds_dst, uint64_t first,
                                         uint32_t cnt, LaneSpan span, BoundsAcc& acc) {
  if constexpr (SRC_AOS && DST_AOS) {
    if (h.quad) {
      if (e.ncomp == 1) { run_tile_lds_to_lds_quad<S, D, 1>(h, e, lds_src, lds_dst, cnt, span, acc); return; }
      if constexpr (Vec3Able<S>::value && Vec3Able<D>::value) {
        if (e.ncomp == 3) { run_tile_lds_to_lds_quad<S, D, 3>(h, e, lds_src, lds_dst, cnt, span, acc); return; }
      }
      if constexpr (std::is_same<S, uint8_t>::value && std::is_same<D, uint8_t>::value) {
        if (e.ncomp == 4) { run_tile_lds_to_lds_quad<S, D, 4>(h, e, lds_src, lds_dst, cnt, span, acc); return; }
      }
    }
    run_tile_lds_to_lds<S, D>(h, e, lds_src, lds_dst, cnt, span, acc);
  }
  else if constexpr (SRC_AOS) run_tile_to_column<S, D>(h, e, lds_src, first, cnt, span, acc);
  else {
    if (h.quad) {
      if (e.ncomp == 1) { run_tile_from_column_quad<S, D, 1>(h, e, lds_dst, first, cnt, span, acc); return; }
      if constexpr (Vec3Able<S>::value && Vec3Able<D>::value) {
        if (e.ncomp == 3) { run_tile_from_column_quad<S, D, 3>(h, e, lds_dst, first, cnt, span, acc); return; }
      }
      if constexpr (std::is_same<S, uint8_t>::value && std::is_same<D, uint8_t>::value) {
        if (e.ncomp == 4) { run_tile_from_column_quad<S, D, 4>(h, e, lds_dst, first, cnt, span, acc); return; }
      }
    }
    run_tile_from_column<S, D>(h, e, lds_dst, first, cnt, span, acc);
  }
}

// Entry scheduling inside a block.  Interpreting an entry (scalar fetch, type dispatch, loop set-up) costs every wave
// that touches it a few hundred issue slots, whatever the entry's size.  Wide attributes are processed by all waves of
// the block; narrow ones (<= 4 bytes per point on the columnar side) are each OWNED by one wave, which sweeps the whole
// tile for that attribute.  The host stores, after the entries, masks[0] = entries handled by every wave and
// masks[1 + w] = entries owned by wave w (w < 16).
template <int BLK, bool SRC_AOS, bool DST_AOS>
__global__ __launch_bounds__(BLK) void convert_tile_kernel(const ConvertHeader h, const PlanEntry* __restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  const uint32_t T = h.tile;
  const uint32_t src_cap = (SRC_AOS && !(DST_AOS && h.in_place)) ? round_up16(T * h.src_stride + 32u) : 0u;
  lptr_t lds_s = lds;
  lptr_t lds_d = lds + src_cap;
  // readfirstlane: the wave index is wave-uniform by construction, but the compiler only knows it derives from threadIdx;
  // without it every entry field lands in VGPRs and the type dispatch becomes exec-mask control flow.
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  const PST_AS_CONST uint32_t* masks = (const PST_AS_CONST uint32_t*)(entries + PST_PLAN_MAX_ENTRIES);
  const uint32_t mask_all = masks[0];
  const uint32_t mask_own = masks[1 + (wave & 15u)];
  BoundsAcc acc;
  acc.init();
  const uint64_t n_tiles = (h.n + T - 1) / T;
  // XCD-aware numbering pays for interleaved -> columnar (+4 %, same-box A/B) and costs 2-4 % on the other pairings
  for (uint64_t tile = (SRC_AOS && !DST_AOS) ? xcd_block_id() : blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * T;
    const uint32_t cnt = (uint32_t)((h.n - first) < T ? (h.n - first) : T);
    uint32_t s_mis = 0, d_mis = 0;
    gptr_t g_dst = nullptr;
    const bool in_place = SRC_AOS && DST_AOS && h.in_place;  // same records on both sides: stage them once
    if constexpr (SRC_AOS) {
      if (!in_place) {
        const uint64_t ga = h.src_aos + first * h.src_stride;
        s_mis = (uint32_t)(ga & 15u);
        tile_load<BLK>(lds_s, as_global(ga - s_mis), round_up16(s_mis + cnt * h.src_stride));
      }
    }
    if constexpr (DST_AOS) {
      const uint64_t ga = h.dst_aos + first * h.dst_stride;
      d_mis = (uint32_t)(ga & 15u);
      g_dst = as_global(ga - d_mis);
      // record bytes no mapping writes (unmapped attributes, padding) must survive: read-modify-write the tile
      if (!h.dst_fully_covered || in_place) tile_load<BLK>(lds_d, g_dst, round_up16(d_mis + cnt * h.dst_stride));
    }
    clptr_t tile_src = in_place ? (clptr_t)(lds_d + d_mis) : (clptr_t)(lds_s + s_mis);
    wait_tile_loads();
    __syncthreads();
    for (uint32_t bits = mask_all | mask_own; bits != 0; bits &= bits - 1) {
      const uint32_t m = (uint32_t)__builtin_ctz(bits);
      const bool shared_entry = (mask_all >> m) & 1u;
      const LaneSpan span = shared_entry ? LaneSpan{threadIdx.x, (uint32_t)BLK} : LaneSpan{lane, 64u};
      const PlanEntry e = fetch_entry(entries, m);
      dispatch_ct(e.src_ct, [&](auto s) __attribute__((always_inline)) {
        using S = decltype(s);
        if (!e.convert) {
          run_tile<SRC_AOS, DST_AOS, S, S>(h, e, tile_src, lds_d + d_mis, first, cnt, span, acc);
        } else {
          dispatch_ct(e.dst_ct, [&](auto d) __attribute__((always_inline)) {
            run_tile<SRC_AOS, DST_AOS, S, decltype(d)>(h, e, tile_src, lds_d + d_mis, first, cnt, span, acc);
          });
        }
      });
    }
    __syncthreads();
    if constexpr (DST_AOS) {
      tile_store<BLK>(lds_d, g_dst, d_mis, cnt * h.dst_stride);
      __syncthreads();
    }
  }
  flush_bounds<BLK>(acc, h.bounds_partials);
}


}  // namespace
