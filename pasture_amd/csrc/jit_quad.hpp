// Point-major conversion kernels with the WHOLE plan as a compile-time constant (gfx950).
//
// The tile kernels of convert_kernels.hpp interpret a mapping list (buffer_conversion.rs:41-55, 98-102) at run time; the LAS kernels
// (las_decode.hip, las_encode.hip, las_transpose.hip) showed what a layout is worth once the compiler sees every offset.  This header
// is that form for ARBITRARY layouts: `quad_convert_body<P>` is instantiated over a plan type P whose entries are constexpr.  P comes
// from two places: jit.cpp writes it as source text for the plan of a BufferLayoutConverter and compiles it with hipRTC at run time
// (cached per plan signature), and convert_static.hip instantiates it in-tree for the layouts of the reference's own benches (a warm
// cache, and the build-time check that this header compiles).
//
// Shape (interleaved = "records", columnar = "columns"; buffer_conversion.rs:489-662):
//   * a workgroup of BLK lanes owns a tile of T = 4 BLK consecutive points, a lane the four consecutive points 4 tid .. 4 tid + 3;
//   * four records of `stride` bytes are 4 stride bytes = `stride` DWORDS: a lane's piece of a record tile is a dword-aligned "quad
//     image" whatever the record size, so every LDS access of this kernel is an aligned dword (or wider) access -- packed(1) records
//     no longer cost unaligned ds_read / ds_write (SQ_LDS_UNALIGNED_STALL);
//   * a column of B-byte values is the same thing with stride B: the lane's four values are B contiguous dwords, loaded / stored with
//     16-byte vector accesses (wave = 256 consecutive points);
//   * record tiles travel between HBM and LDS flat and coalesced (LDS-DMA in, 16-byte stores out);
//   * values move between the source and the target image with compile-time byte offsets: plain copies as byte strings, converted
//     ones (Rust `as`, attribute_conversion.rs:310-343) and transformed ones (raw_readers.rs:42-164) component by component.
// The kernel covers FULL tiles of a range whose interleaved sides start 16-byte aligned; the host sends the ragged tail (< T points)
// and unaligned ranges through the interpreted kernels.
#pragma once
#include "device_common.hpp"
#include "tile_io.hpp"

namespace pstq {

using namespace pstd;

template <uint32_t CT> struct CtType;
template <> struct CtType<CT_U8> { typedef uint8_t type; };
template <> struct CtType<CT_I8> { typedef int8_t type; };
template <> struct CtType<CT_U16> { typedef uint16_t type; };
template <> struct CtType<CT_I16> { typedef int16_t type; };
template <> struct CtType<CT_U32> { typedef uint32_t type; };
template <> struct CtType<CT_I32> { typedef int32_t type; };
template <> struct CtType<CT_U64> { typedef uint64_t type; };
template <> struct CtType<CT_I64> { typedef int64_t type; };
template <> struct CtType<CT_F32> { typedef float type; };
template <> struct CtType<CT_F64> { typedef double type; };

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename T> struct BitsOf;
template <> struct BitsOf<uint8_t> { typedef uint8_t type; };
template <> struct BitsOf<int8_t> { typedef uint8_t type; };
template <> struct BitsOf<uint16_t> { typedef uint16_t type; };
template <> struct BitsOf<int16_t> { typedef uint16_t type; };
template <> struct BitsOf<uint32_t> { typedef uint32_t type; };
template <> struct BitsOf<int32_t> { typedef uint32_t type; };
template <> struct BitsOf<float> { typedef uint32_t type; };
template <> struct BitsOf<uint64_t> { typedef uint64_t type; };
template <> struct BitsOf<int64_t> { typedef uint64_t type; };
template <> struct BitsOf<double> { typedef uint64_t type; };

template <typename T>
__device__ __forceinline__ T from_bits(uint64_t v) { return __builtin_bit_cast(T, (typename BitsOf<T>::type)v); }
template <typename T>
__device__ __forceinline__ uint64_t to_bits(T v) { return (uint64_t)__builtin_bit_cast(typename BitsOf<T>::type, v); }

// ---- quad images: little-endian byte strings held in dword registers -------------------------------------------------------------------
// Every offset is a TEMPLATE constant: the arrays are indexed by literals from the first optimisation pass on and become registers at once
// (offsets that only fold after loop unrolling let the optimiser merge neighbouring words into 64-bit loads of the array first, and the
// array then stays in scratch memory).
// up to 8 bytes starting at byte OFF, zero-extended
template <uint32_t OFF, uint32_t NB, int NW>
__device__ __forceinline__ uint64_t img_get(const uint32_t (&w)[NW]) {
  constexpr uint32_t wi = OFF >> 2, sh = (OFF & 3u) * 8u;
  static_assert(NB >= 1 && NB <= 8 && wi < (uint32_t)NW, "byte string inside the image");
  uint32_t lo = w[wi], hi = 0;
  if constexpr (sh == 0) {
    if constexpr (NB > 4) hi = w[wi + 1];
  } else {
    if constexpr (sh + 8u * NB > 32u) {
      const uint32_t mid = w[wi + 1];
      lo = __builtin_amdgcn_alignbit(mid, lo, sh);
      if constexpr (sh + 8u * NB > 64u) hi = __builtin_amdgcn_alignbit(w[wi + 2], mid, sh);
      else if constexpr (NB > 4) hi = mid >> sh;
    } else {
      lo >>= sh;
    }
  }
  if constexpr (NB < 4) lo &= (1u << (8u * NB)) - 1u;
  if constexpr (NB > 4 && NB < 8) hi &= (1u << (8u * (NB - 4u))) - 1u;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v; v.x = lo; v.y = hi;
  return __builtin_bit_cast(uint64_t, v);
}
// v (zero-extended) into the bytes [OFF, OFF + NB); CLEAR: the bytes may hold old data (read-modify-written records)
template <bool CLEAR, uint32_t OFF, uint32_t NB, int NW>
__device__ __forceinline__ void img_put(uint32_t (&w)[NW], uint64_t v) {
  constexpr uint32_t wi = OFF >> 2, sh = (OFF & 3u) * 8u;
  static_assert(NB >= 1 && NB <= 8 && wi < (uint32_t)NW, "byte string inside the image");
  constexpr uint64_t ones = NB >= 8u ? ~0ull : ((1ull << (8u * (NB & 7u))) - 1ull);
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if constexpr (CLEAR) {
    w[wi] &= ~(uint32_t)(ones << sh);
    if constexpr (sh + 8u * NB > 32u) w[wi + 1] &= ~(uint32_t)(sh ? (ones >> (32u - sh)) : (ones >> 32));
    if constexpr (sh + 8u * NB > 64u) w[wi + 2] &= ~(uint32_t)(ones >> (64u - sh));
  }
  w[wi] |= lo << sh;
  if constexpr (sh + 8u * NB > 32u) {
    if constexpr (sh == 0) w[wi + 1] |= hi;
    else w[wi + 1] |= __builtin_amdgcn_alignbit(hi, lo, 32u - sh);
  }
  if constexpr (sh + 8u * NB > 64u) w[wi + 2] |= hi >> (32u - sh);
}

// ND dwords between a (possibly unaligned: range starts are arbitrary) column address and w[base ..): 16 / 8 / 4-byte accesses
template <uint32_t ND, bool NT, int NW>
__device__ __forceinline__ void load_words(cgptr_t p, uint32_t (&w)[NW], uint32_t base) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr uint32_t K4 = ND / 4u * 4u, K2 = K4 + ((ND - K4) >= 2u ? 2u : 0u);
#pragma unroll
  for (uint32_t k = 0; k < K4; k += 4) {
    const PST_AS_GLOBAL Unaligned<u32x4>::type* q = reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x4>::type*>(p + 4u * k);
    const u32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
    w[base + k] = v.x; w[base + k + 1] = v.y; w[base + k + 2] = v.z; w[base + k + 3] = v.w;
  }
  if constexpr (K2 > K4) {
    const PST_AS_GLOBAL Unaligned<u32x2>::type* q = reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x2>::type*>(p + 4u * K4);
    const u32x2 v = NT ? __builtin_nontemporal_load(q) : *q;
    w[base + K4] = v.x; w[base + K4 + 1] = v.y;
  }
  if constexpr (ND > K2) {
    const PST_AS_GLOBAL Unaligned<uint32_t>::type* q = reinterpret_cast<const PST_AS_GLOBAL Unaligned<uint32_t>::type*>(p + 4u * K2);
    w[base + K2] = NT ? __builtin_nontemporal_load(q) : *q;
  }
}
template <uint32_t ND, bool NT>
__device__ __forceinline__ void store_words(gptr_t p, const uint32_t (&w)[ND]) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr uint32_t K4 = ND / 4u * 4u, K2 = K4 + ((ND - K4) >= 2u ? 2u : 0u);
#pragma unroll
  for (uint32_t k = 0; k < K4; k += 4) {
    u32x4 v; v.x = w[k]; v.y = w[k + 1]; v.z = w[k + 2]; v.w = w[k + 3];
    PST_AS_GLOBAL Unaligned<u32x4>::type* q = reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x4>::type*>(p + 4u * k);
    if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v;
  }
  if constexpr (K2 > K4) {
    u32x2 v; v.x = w[K4]; v.y = w[K4 + 1];
    PST_AS_GLOBAL Unaligned<u32x2>::type* q = reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x2>::type*>(p + 4u * K4);
    if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v;
  }
  if constexpr (ND > K2) {
    PST_AS_GLOBAL Unaligned<uint32_t>::type* q = reinterpret_cast<PST_AS_GLOBAL Unaligned<uint32_t>::type*>(p + 4u * K2);
    if constexpr (NT) __builtin_nontemporal_store(w[K2], q); else *q = w[K2];
  }
}

// A lane's quad image in an LDS record tile: NW dwords at a dword-aligned address, as the widest accesses its alignment allows
// (lanes are NW dwords apart: odd NW is conflict-free with b32, NW = 2 mod 4 with b64, NW = 4 mod 8 with b128).
template <int NW>
__device__ __forceinline__ void lds_read_image(clptr_t p, uint32_t (&w)[NW]) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NW % 4 == 0) {
#pragma unroll
    for (int k = 0; k < NW; k += 4) {
      const u32x4 v = *reinterpret_cast<cl4ptr_t>(p + 4 * k);
      w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
  } else if constexpr (NW % 2 == 0) {
#pragma unroll
    for (int k = 0; k < NW; k += 2) {
      const u32x2 v = *reinterpret_cast<const PST_AS_LDS u32x2*>(p + 4 * k);
      w[k] = v.x; w[k + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = *reinterpret_cast<const PST_AS_LDS uint32_t*>(p + 4 * k);
  }
}
template <int NW>
__device__ __forceinline__ void lds_write_image(lptr_t p, const uint32_t (&w)[NW]) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NW % 4 == 0) {
#pragma unroll
    for (int k = 0; k < NW; k += 4) {
      u32x4 v; v.x = w[k]; v.y = w[k + 1]; v.z = w[k + 2]; v.w = w[k + 3];
      *reinterpret_cast<l4ptr_t>(p + 4 * k) = v;
    }
  } else if constexpr (NW % 2 == 0) {
#pragma unroll
    for (int k = 0; k < NW; k += 2) {
      u32x2 v; v.x = w[k]; v.y = w[k + 1];
      *reinterpret_cast<PST_AS_LDS u32x2*>(p + 4 * k) = v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NW; ++k) *reinterpret_cast<PST_AS_LDS uint32_t*>(p + 4 * k) = w[k];
  }
}

// LDS -> global copy of a whole tile of NBYTES (a multiple of 16, both sides 16-byte aligned): the trip count is a constant, so the
// copy is straight-line code -- every LDS read of a batch issued before its stores, no per-chunk bounds test.
template <int BLK, uint32_t NBYTES, bool NT>
__device__ __forceinline__ void tile_store_const(clptr_t lds, gptr_t gbase) {
  static_assert(NBYTES % 16u == 0, "whole 16-byte chunks");
  constexpr uint32_t NV = NBYTES / 16u, FULL = NV / (uint32_t)BLK, REM = NV % (uint32_t)BLK;
  constexpr uint32_t kBatch = 8;
  cl4ptr_t l = reinterpret_cast<cl4ptr_t>(lds) + threadIdx.x;
  g4ptr_t g = reinterpret_cast<g4ptr_t>(gbase) + threadIdx.x;
#pragma unroll
  for (uint32_t i0 = 0; i0 < FULL; i0 += kBatch) {
    u32x4 v[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) if (i0 + u < FULL) v[u] = l[(i0 + u) * BLK];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) {
      if (i0 + u < FULL) {
        if constexpr (NT) __builtin_nontemporal_store(v[u], &g[(i0 + u) * BLK]);
        else g[(i0 + u) * BLK] = v[u];
      }
    }
  }
  if constexpr (REM != 0) {
    if (threadIdx.x < REM) {
      if constexpr (NT) __builtin_nontemporal_store(l[FULL * BLK], &g[FULL * BLK]);
      else g[FULL * BLK] = l[FULL * BLK];
    }
  }
}

// tile_load (tile_io.hpp) with the cache policy as a constant: AUX = 2 requests the tile non-temporally (read once, by this workgroup)
template <int BLK, int AUX>
__device__ __forceinline__ void tile_load_const(lptr_t lds, cgptr_t gbase, uint32_t nbytes16) {
  const uint32_t nvec = nbytes16 >> 4;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i = threadIdx.x; i < nvec; i += BLK)
    __builtin_amdgcn_global_load_lds((const PST_AS_GLOBAL void*)(gbase + (uint64_t)i * 16), (PST_AS_LDS void*)(lds + (i - lane) * 16u), 16, 0, AUX);
}

// One value: optional pre-transform, `as` D, optional post-transform (buffer_conversion.rs:446-456), the kind a constant
template <typename S, typename D, uint32_t KIND, uint32_t PRE>
__device__ __forceinline__ D quad_convert_value(S v, double sc, double of, uint32_t shift, uint64_t mask) {
  if constexpr (KIND == 0) {
    return rust_as<D, S>(v);
  } else {
    if constexpr (PRE != 0) v = apply_xf<S>(v, KIND, sc, of, shift, mask);
    D w = rust_as<D, S>(v);
    if constexpr (PRE == 0) w = apply_xf<D>(w, KIND, sc, of, shift, mask);
    return w;
  }
}

// run-time part of a transformation, read through the constant address space (wave-uniform scalar loads)
struct QXf {
  double sc[3], of[3];
  uint64_t mask;
  uint32_t shift;
};
__device__ __forceinline__ QXf load_qxf(const PlanEntry* entries, int m) {
  const PST_AS_CONST PlanEntry* e = (const PST_AS_CONST PlanEntry*)(entries + m);
  QXf x;
  x.sc[0] = e->scale[0]; x.sc[1] = e->scale[1]; x.sc[2] = e->scale[2];
  x.of[0] = e->offset[0]; x.of[1] = e->offset[1]; x.of[2] = e->offset[2];
  x.mask = e->mask; x.shift = e->shift;
  return x;
}

// Point T of mapping M of plan P: from the lane's source words into the target image (the record image of an interleaved target, the
// attribute's own quad image of a columnar one).
// (pidx: the point's index in the source buffer -- the `i` of a fused expression; unused, and folded away, in every other plan)
template <typename P, int M, uint32_t T, int SW, int TW>
__device__ __forceinline__ void move_point(const uint32_t (&sw)[SW], uint32_t (&tw)[TW], const QXf& x, double (&lo)[3], double (&hi)[3], const uint64_t pidx,
                                           const uint64_t (&params)[4]) {
  constexpr QEntry e = P::entry(M);
  typedef typename CtType<e.src_ct>::type S;
  typedef typename CtType<e.dst_ct>::type D;
  constexpr bool plain = e.convert == 0 && e.xf_kind == 0 && e.bounds == 0;  // (an expression entry never is: its branch comes first)
  constexpr bool clear = P::dst_aos && !P::covered;
  constexpr uint32_t so = (P::src_aos ? e.src_off : 4u * e.src_img) + T * (P::src_aos ? P::src_stride : e.src_size);
  constexpr uint32_t dofs = (P::dst_aos ? e.dst_off : 0u) + T * (P::dst_aos ? P::dst_stride : e.dst_size);
  if constexpr (e.xf_kind == PST_XF_EXPR) {
    // A user-written transformation (expr.cpp), part of the plan's translation unit as P::expr<M, C>: the arithmetic of expr.cpp's own
    // strided kernel (map_source) -- TI = the source type when the closure sees the source value, else the target type; every component
    // of the value is at hand as x, y, z; the result goes through Rust `as` into TI and then into the target type -- inside the conversion pass.
    constexpr bool pre = e.xf_pre != 0;
    typedef typename std::conditional<pre, S, D>::type TI;
    TI in[3];
    static_for<0, (int)e.ncomp>([&](auto C) __attribute__((always_inline)) {
      constexpr uint32_t c = (uint32_t) decltype(C)::value;
      const S v = from_bits<S>(img_get<so + c * (uint32_t)sizeof(S), (uint32_t)sizeof(S)>(sw));
      if constexpr (pre) in[c] = (TI)v; else in[c] = (TI)rust_as<D, S>(v);
    });
    if constexpr (e.ncomp < 3) { in[1] = in[0]; in[2] = in[0]; }
    static_for<0, (int)e.ncomp>([&](auto C) __attribute__((always_inline)) {
      constexpr uint32_t c = (uint32_t) decltype(C)::value;
      const TI r = P::template expr<M, (int)c, TI>(in[c], in[0], in[1], in[2], pidx, params);
      const D d = rust_as<D, TI>(r);
      img_put<clear, dofs + c * (uint32_t)sizeof(D), (uint32_t)sizeof(D)>(tw, to_bits<D>(d));
    });
  } else if constexpr (plain) {  // same datatype, no transformation: a byte string
    static_for<0, (int)((e.src_size + 7u) / 8u)>([&](auto B) __attribute__((always_inline)) {
      constexpr QEntry f = P::entry(M);
      constexpr uint32_t b = 8u * (uint32_t) decltype(B)::value;
      constexpr uint32_t nb = f.src_size - b < 8u ? f.src_size - b : 8u;
      img_put<clear, dofs + b, nb>(tw, img_get<so + b, nb>(sw));
    });
  } else {
    static_for<0, (int)e.ncomp>([&](auto C) __attribute__((always_inline)) {
      constexpr QEntry f = P::entry(M);
      constexpr uint32_t c = (uint32_t) decltype(C)::value;
      const S v = from_bits<S>(img_get<so + c * (uint32_t)sizeof(S), (uint32_t)sizeof(S)>(sw));
      D d;
      if constexpr (f.xf_kind != 0) {
        constexpr uint32_t cc = f.ncomp == 3 ? c : 0u;
        d = quad_convert_value<S, D, f.xf_kind, f.xf_pre>(v, x.sc[cc], x.of[cc], x.shift, x.mask);
      } else {
        d = rust_as<D, S>(v);
      }
      if constexpr (f.bounds != 0 && std::is_same<D, double>::value) {
        lo[c % 3u] = __builtin_fmin(lo[c % 3u], d);
        hi[c % 3u] = __builtin_fmax(hi[c % 3u], d);
      }
      img_put<clear, dofs + c * (uint32_t)sizeof(D), (uint32_t)sizeof(D)>(tw, to_bits<D>(d));
    });
  }
}

// words [base, base + ND) of w from / to ND dwords at a dword-aligned LDS address (a lane's quad of a staged column: lanes ND dwords apart)
template <uint32_t ND, int NW>
__device__ __forceinline__ void lds_read_words(clptr_t p, uint32_t (&w)[NW], uint32_t base) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (ND % 4u == 0) {
#pragma unroll
    for (uint32_t k = 0; k < ND; k += 4) {
      const u32x4 v = *reinterpret_cast<cl4ptr_t>(p + 4u * k);
      w[base + k] = v.x; w[base + k + 1] = v.y; w[base + k + 2] = v.z; w[base + k + 3] = v.w;
    }
  } else if constexpr (ND % 2u == 0) {
#pragma unroll
    for (uint32_t k = 0; k < ND; k += 2) {
      const u32x2 v = *reinterpret_cast<const PST_AS_LDS u32x2*>(p + 4u * k);
      w[base + k] = v.x; w[base + k + 1] = v.y;
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < ND; ++k) w[base + k] = *reinterpret_cast<const PST_AS_LDS uint32_t*>(p + 4u * k);
  }
}

// A plan type P provides:
//   static constexpr int n;                                  // mappings
//   static constexpr bool src_aos, dst_aos;                  // which sides are interleaved (at least one is)
//   static constexpr uint32_t src_stride, dst_stride;        // record sizes of the interleaved sides (0 on a columnar side)
//   static constexpr uint32_t covered;                       // interleaved target: every byte of a record is written by some mapping
//   static constexpr int blk;                                // lanes per workgroup (tile = 4 blk points)
//   static constexpr uint32_t xcd, nt;                       // XCD-aware tile numbering; non-temporal accesses: bit 0 narrow column loads, 1 narrow column stores, 2 tile stores, 3 tile loads (LDS-DMA)
//   static constexpr uint32_t src_words;                     // columnar source: dwords of the lane's source images (sum of the loaded attributes' sizes)
//   static constexpr uint32_t lds_per_point;                 // LDS bytes per point of the tile: record tiles + staged wide columns
//   static constexpr uint32_t dst_tile_off, alias;           // target record tile at LDS byte T * dst_tile_off; alias: the outgoing regions overlay the incoming ones
//   static constexpr QEntry entry(int m);
//   template <int M, int C, typename TI> static TI expr(TI v, TI x, TI y, TI z, uint64_t i, const uint64_t (&params)[4]);   // plans with PST_XF_EXPR entries only (jit.cpp writes it)
// LDS regions (all flat, 16-byte aligned, at T * {0, src_stride, entry.src_stage / dst_stage}): the source record tile, the target record tile,
// and one region per WIDE columnar attribute (>= 8 bytes per value).  A wide column's tile crosses HBM lane-contiguously -- a lane's own four
// values are 32 ... 96 bytes apart from its neighbour's, which wastes most of every memory transaction --, LDS-DMA in / 16-byte stores out like
// a record tile, and lanes pick / put their quads in LDS; narrow columns (4 values = 4 ... 28 bytes per lane) go straight between registers and HBM.
template <typename P>
__device__ __forceinline__ void quad_convert_body(const ConvertHeader& h, const PlanEntry* __restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pstq_lds[];
  constexpr int BLK = P::blk;
  constexpr uint32_t T = 4u * (uint32_t)BLK;
  constexpr uint32_t SS = P::src_stride, DS = P::dst_stride;
  constexpr int SW = P::src_aos ? (int)SS : (int)P::src_words;
  constexpr int DW = P::dst_aos ? (int)DS : 1;
  constexpr int kDmaAux = (P::nt & 8u) != 0 ? 2 : 0;
  lptr_t lds = (lptr_t)pstq_lds;
  lptr_t lds_s = lds;
  lptr_t lds_d = lds + T * P::dst_tile_off;
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = h.n / T;  // full tiles only (the host peels the rest)
  const uint64_t tile = P::xcd ? (uint64_t)xcd_block_id() : (uint64_t)blockIdx.x;
  double lo[3] = {kF64Max, kF64Max, kF64Max}, hi[3] = {-kF64Max, -kF64Max, -kF64Max};
  if (tile < n_tiles) {
    const uint64_t first = tile * T;
    const uint64_t p0 = first + 4u * tid;  // this lane's first point
    // ---- phase 1: everything that comes from HBM is requested at once ----------------------------------------------------------------
    if constexpr (P::src_aos) tile_load_const<BLK, kDmaAux>(lds_s, as_global(h.src_aos + first * SS), T * SS);
    if constexpr (P::dst_aos && !P::covered) tile_load_const<BLK, 0>(lds_d, as_global(h.dst_aos + first * DS), T * DS);
    uint32_t sw[SW];
    bool staged_in = P::src_aos || (P::dst_aos && !P::covered);
    if constexpr (!P::src_aos) {
      static_for<0, P::n>([&](auto I) __attribute__((always_inline)) {
        constexpr QEntry e = P::entry(decltype(I)::value);
        if constexpr (e.src_load != 0) {
          const uint64_t col = ((const PST_AS_CONST PlanEntry*)(entries + decltype(I)::value))->src_col;
          if constexpr (e.src_wide != 0) {
            tile_load_const<BLK, kDmaAux>(lds + T * e.src_stage, (cgptr_t)as_global(col) + first * e.src_size, T * e.src_size);
            staged_in = true;
          } else {
            load_words<e.src_size, (P::nt & 1u) != 0>((cgptr_t)as_global(col) + p0 * e.src_size, sw, e.src_img);
          }
        }
      });
    }
    if (staged_in) {  // a constant after inlining
      wait_tile_loads();
      __syncthreads();
    }
    // ---- phase 2: the lane's four points, in registers ---------------------------------------------------------------------------------
    if constexpr (P::src_aos) lds_read_image<SW>(lds_s + tid * (4u * SS), sw);
    if constexpr (!P::src_aos) {
      static_for<0, P::n>([&](auto I) __attribute__((always_inline)) {
        constexpr QEntry e = P::entry(decltype(I)::value);
        if constexpr (e.src_load != 0 && e.src_wide != 0) lds_read_words<e.src_size>(lds + T * e.src_stage + tid * (4u * e.src_size), sw, e.src_img);
      });
    }
    // aliased regions: what leaves through LDS is written over what came in, once every lane holds its inputs in registers
    if constexpr (P::alias != 0) __syncthreads();
    uint32_t dw[DW];
    if constexpr (P::dst_aos) {
      if constexpr (P::covered) {
#pragma unroll
        for (int k = 0; k < DW; ++k) dw[k] = 0;
      } else {
        lds_read_image<DW>(lds_d + tid * (4u * DS), dw);  // bytes no mapping writes (unmapped attributes, padding) survive
      }
    }
    bool staged_out = P::dst_aos;
    static_for<0, P::n>([&](auto I) __attribute__((always_inline)) {
      constexpr int M = decltype(I)::value;
      constexpr QEntry e = P::entry(M);
      constexpr int CW = P::dst_aos ? 1 : (int)e.dst_size;
      uint32_t cw[CW];  // columnar target: this attribute's quad image
      if constexpr (!P::dst_aos) {
#pragma unroll
        for (int k = 0; k < CW; ++k) cw[k] = 0;
      }
      QXf x;
      if constexpr (e.xf_kind != 0 && e.xf_kind != PST_XF_EXPR) x = load_qxf(entries, M);
      static_for<0, 4>([&](auto TT) __attribute__((always_inline)) {
        if constexpr (P::dst_aos) move_point<P, M, (uint32_t) decltype(TT)::value>(sw, dw, x, lo, hi, h.first_index + p0 + (uint32_t) decltype(TT)::value, h.expr_params);
        else move_point<P, M, (uint32_t) decltype(TT)::value>(sw, cw, x, lo, hi, h.first_index + p0 + (uint32_t) decltype(TT)::value, h.expr_params);
      });
      if constexpr (!P::dst_aos) {
        if constexpr (e.dst_wide != 0) {
          lds_write_image<CW>(lds + T * e.dst_stage + tid * (4u * e.dst_size), cw);
          staged_out = true;
        } else {
          const uint64_t col = ((const PST_AS_CONST PlanEntry*)(entries + M))->dst_col;
          store_words<e.dst_size, (P::nt & 2u) != 0>(as_global(col) + p0 * e.dst_size, cw);
        }
      }
    });
    // ---- phase 3: tiles leave flat -----------------------------------------------------------------------------------------------------
    if constexpr (P::dst_aos) lds_write_image<DW>(lds_d + tid * (4u * DS), dw);
    if (staged_out) __syncthreads();
    if constexpr (P::dst_aos) tile_store_const<BLK, T * DS, (P::nt & 4u) != 0>(lds_d, as_global(h.dst_aos + first * DS));
    if constexpr (!P::dst_aos) {
      static_for<0, P::n>([&](auto I) __attribute__((always_inline)) {
        constexpr QEntry e = P::entry(decltype(I)::value);
        if constexpr (e.dst_wide != 0) {
          const uint64_t col = ((const PST_AS_CONST PlanEntry*)(entries + decltype(I)::value))->dst_col;
          tile_store_const<BLK, T * e.dst_size, (P::nt & 4u) != 0>(lds + T * e.dst_stage, as_global(col) + first * e.dst_size);
        }
      });
    }
  }
  if (h.bounds_partials != 0) {  // wave-uniform; every workgroup of the grid writes its record (idle ones the seeds)
    __syncthreads();             // the tile is no longer needed: its first bytes serve as the reduction's scratch
    block_reduce_minmax<double, 3, BLK>(lo, hi, (double*)pstq_lds);
    if (threadIdx.x == 0) {
      double* out = (double*)h.bounds_partials + (uint64_t)blockIdx.x * 6;
      out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2];
      out[3] = hi[0]; out[4] = hi[1]; out[5] = hi[2];
    }
  }
}

// LDS bytes a launch of plan P needs (host side mirrors this in jit.cpp for run-time plans)
template <typename P>
constexpr uint32_t quad_lds_bytes() {
  constexpr uint32_t b = 4u * (uint32_t)P::blk * P::lds_per_point;
  return b < 256u ? 256u : b;  // the AABB reduction's scratch
}

}  // namespace pstq
