// Device-side helpers shared by the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#ifdef __HIPCC_RTC__
#include "rtc_prelude.hpp"  // compiled at run time by hipRTC (jit.cpp): no standard library headers there
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits>
#include <type_traits>
#endif

#include "plan.h"

namespace pstd {

constexpr int kBlock = 256;  // 4 waves of 64
constexpr double kF64Max = 1.7976931348623157e308;

// component type codes (must match pst::CompType in core.hpp)
enum : uint32_t { CT_U8 = 0, CT_I8, CT_U16, CT_I16, CT_U32, CT_I32, CT_U64, CT_I64, CT_F32, CT_F64 };

template <typename F>
__device__ __forceinline__ void dispatch_ct(uint32_t ct, F&& f) {
  switch (ct) {
    case CT_U8: f(uint8_t{}); break;
    case CT_I8: f(int8_t{}); break;
    case CT_U16: f(uint16_t{}); break;
    case CT_I16: f(int16_t{}); break;
    case CT_U32: f(uint32_t{}); break;
    case CT_I32: f(int32_t{}); break;
    case CT_U64: f(uint64_t{}); break;
    case CT_I64: f(int64_t{}); break;
    case CT_F32: f(float{}); break;
    default: f(double{}); break;
  }
}

// Address-space qualified byte pointers: pointers rebuilt from 64-bit integers are "flat" to the compiler; naming the
// address space keeps global traffic on global_load/global_store and LDS traffic on ds_read/ds_write.
#define PST_AS_GLOBAL __attribute__((address_space(1)))
#define PST_AS_LDS __attribute__((address_space(3)))
#define PST_AS_CONST __attribute__((address_space(4)))
typedef PST_AS_GLOBAL uint8_t* gptr_t;
typedef const PST_AS_GLOBAL uint8_t* cgptr_t;
typedef PST_AS_LDS uint8_t* lptr_t;
typedef const PST_AS_LDS uint8_t* clptr_t;

__device__ __forceinline__ gptr_t as_global(uint64_t addr) { return (gptr_t)addr; }

// XCD-aware logical workgroup id.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
// L2), so with tile = blockIdx every XCD touches every eighth tile of a stream.  Numbering the tiles (b % 8) * (G / 8) + b / 8
// instead gives each XCD one contiguous eighth of the range: +2-5 % on the Vec3f64 stream kernel and +4 % on interleaved ->
// columnar tiles, but -1..-4 % on the column, encoder, decoder and columnar -> interleaved kernels (same-box A/B) — applied selectively.  The host rounds the
// grid G up to a multiple of 8; callers skip logical ids beyond their tile count.
__device__ __forceinline__ uint32_t xcd_block_id() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }

// Unaligned typed access (packed(1) layouts have no natural alignment: Vec3f64 at offset 14, u16 at odd offsets).
// gfx950 executes unaligned global and LDS accesses natively, so align-1 accesses stay single instructions.
template <typename T>
struct Unaligned {
  typedef T __attribute__((aligned(1))) type;
};
template <typename T>
__device__ __forceinline__ T load_un(cgptr_t p) {
  return *reinterpret_cast<const PST_AS_GLOBAL typename Unaligned<T>::type*>(p);
}
template <typename T>
__device__ __forceinline__ T load_un(clptr_t p) {
  return *reinterpret_cast<const PST_AS_LDS typename Unaligned<T>::type*>(p);
}
template <typename T>
__device__ __forceinline__ void store_un(gptr_t p, T v) {
  *reinterpret_cast<PST_AS_GLOBAL typename Unaligned<T>::type*>(p) = v;
}
template <typename T>
__device__ __forceinline__ void store_un(lptr_t p, T v) {
  *reinterpret_cast<PST_AS_LDS typename Unaligned<T>::type*>(p) = v;
}

// LDS accesses to packed(1) records.  gfx950 executes unaligned ds_read/ds_write, but an access that is not naturally aligned
// stalls the LDS pipe for tens of cycles (measured: a 35-byte-stride record tile read with unaligned b64 accesses ran the
// whole kernel at 3.3 TB/s, the same reads as aligned dwords + v_alignbyte at 5.2 TB/s).  So: read the covering aligned
// dwords and re-align in registers; split stores by alignment class into naturally aligned pieces.
// lds_load may read up to 7 bytes past the value (tiles carry 32 bytes of slack).
template <typename T>
__device__ __forceinline__ T lds_load(clptr_t p) {
  if constexpr (sizeof(T) == 1) {
    return load_un<T>(p);
  } else {
    const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
    const PST_AS_LDS uint32_t* q = (const PST_AS_LDS uint32_t*)(p - m);
    const uint32_t d0 = q[0], d1 = q[1];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, m);
    if constexpr (sizeof(T) == 8) {
      const uint32_t d2 = q[2];
      const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, m);
      return __builtin_bit_cast(T, (uint64_t)lo | ((uint64_t)hi << 32));
    } else if constexpr (sizeof(T) == 4) {
      return __builtin_bit_cast(T, lo);
    } else {
      return __builtin_bit_cast(T, (uint16_t)lo);
    }
  }
}
template <typename T>
__device__ __forceinline__ void lds_store(lptr_t p, T v) {
  typedef PST_AS_LDS uint8_t* p8;
  typedef PST_AS_LDS uint16_t* p16;
  typedef PST_AS_LDS uint32_t* p32;
  if constexpr (sizeof(T) == 1) {
    store_un<T>(p, v);
  } else if constexpr (sizeof(T) == 2) {
    const uint16_t b = __builtin_bit_cast(uint16_t, v);
    if (((uint32_t)(uintptr_t)p & 1u) == 0) *(p16)p = b;
    else { *(p8)p = (uint8_t)b; *(p8)(p + 1) = (uint8_t)(b >> 8); }
  } else if constexpr (sizeof(T) == 4) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v), m = (uint32_t)(uintptr_t)p & 3u;
    if (m == 0) *(p32)p = b;
    else if (m == 2) { *(p16)p = (uint16_t)b; *(p16)(p + 2) = (uint16_t)(b >> 16); }
    else { *(p8)p = (uint8_t)b; *(p16)(p + 1) = (uint16_t)(b >> 8); *(p8)(p + 3) = (uint8_t)(b >> 24); }
  } else {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
    if (m == 0) { *(p32)p = (uint32_t)b; *(p32)(p + 4) = (uint32_t)(b >> 32); }
    else if (m == 2) { *(p16)p = (uint16_t)b; *(p32)(p + 2) = (uint32_t)(b >> 16); *(p16)(p + 6) = (uint16_t)(b >> 48); }
    else if (m == 1) { *(p8)p = (uint8_t)b; *(p16)(p + 1) = (uint16_t)(b >> 8); *(p32)(p + 3) = (uint32_t)(b >> 24); *(p8)(p + 7) = (uint8_t)(b >> 56); }
    else { *(p8)p = (uint8_t)b; *(p32)(p + 1) = (uint32_t)(b >> 8); *(p16)(p + 5) = (uint16_t)(b >> 40); *(p8)(p + 7) = (uint8_t)(b >> 56); }
  }
}

// Rust `as` (attribute_conversion.rs:310-343 via num_traits::AsPrimitive):
//   int->int   two's complement truncate / sign- or zero-extend
//   int->float round to nearest even
//   float->int truncate toward zero, saturate at To::MIN / To::MAX, NaN -> 0
//   f64->f32   RNE, overflow -> +-inf, subnormals kept;  f32->f64 exact
template <typename To, typename From>
__device__ __forceinline__ To rust_as(From v) {
  if constexpr (std::is_same<To, From>::value) {
    return v;
  } else if constexpr (std::is_floating_point<From>::value && std::is_integral<To>::value) {
    constexpr int digits = std::is_signed<To>::value ? (int)sizeof(To) * 8 - 1 : (int)sizeof(To) * 8;
    // 2^digits is exactly representable in f32 and f64 for digits <= 64
    const From hi = (From)__builtin_ldexp(1.0, digits);
    // SELECTS, not early returns (round 6): written with `if (...) return`, every conversion became two or three execution-mask branches in the
    // kernels that convert in bulk (the LAS encoder: twelve per lane and tile).  The value handed to the conversion instruction is always in range
    // (0 stands in for NaN and for the saturating cases), so the C++ conversion is defined; the results of those cases are selected afterwards.
    const bool over = v >= hi;
    bool under;
    if constexpr (std::is_signed<To>::value) under = v <= -hi;  // -2^digits == To::MIN exactly
    else under = v < (From)0;                                   // (-1, 0) truncates to 0 as well
    const bool in_range = v == v && !over && !under;
    const From safe = in_range ? v : (From)0;
    To r;
    if constexpr (sizeof(To) <= 4) {
      if constexpr (std::is_signed<To>::value) r = (To)(int32_t)safe;
      else r = (To)(uint32_t)safe;
    } else {
      if constexpr (std::is_signed<To>::value) r = (To)(int64_t)safe;
      else r = (To)(uint64_t)safe;
    }
    r = over ? std::numeric_limits<To>::max() : r;
    r = under ? (std::is_signed<To>::value ? std::numeric_limits<To>::min() : (To)0) : r;
    return r;
  } else {
    return static_cast<To>(v);
  }
}

// Select one of three VALUES by component index.  By-value parameters on purpose: a conditional over lvalues
// (`c == 0 ? e.scale[0] : ...`) or a dynamic `scale[c]` selects an ADDRESS and drags the plan entry into scratch.
__device__ __forceinline__ double pick3(uint32_t c, double a, double b, double d) { return c == 0 ? a : (c == 1 ? b : d); }

// Closed-set transformation applied to one component; sc / of are that component's scale and offset.
template <typename T>
__device__ __forceinline__ T apply_xf(T v, uint32_t kind, double sc, double of, uint32_t shift, uint64_t mask) {
  if (kind == 1 /*AFFINE*/) {
    // (p * scale) + offset with two roundings — pasture-io/src/las/raw_readers.rs:42-55.  This file is compiled with
    // -ffp-contract=off; the contraction pragma is belt and braces.
#pragma clang fp contract(off)
    if constexpr (std::is_same<T, double>::value) {
      double m = v * sc;
      return m + of;
    } else if constexpr (std::is_same<T, float>::value) {
      double m = (double)v * sc;
      return (float)(m + of);
    } else {
      return v;
    }
  } else if (kind == 2 /*BITFIELD*/) {
    if constexpr (std::is_integral<T>::value && std::is_unsigned<T>::value) return (T)(((uint64_t)v >> shift) & mask);
    else return v;
  }
  return v;
}

// ---- min / max folding with the reference's semantics ---------------------------------------------------
// Floats: `if v < m { m = v }` never lets a NaN win (bounds.rs:34-51, math/minmax.rs:78-94) == fmin/fmax with a
// non-NaN accumulator.  Integers: cmp::min / cmp::max.
template <typename T>
__device__ __forceinline__ T fold_min(T a, T b) {
  if constexpr (std::is_same<T, double>::value) return __builtin_fmin(a, b);
  else if constexpr (std::is_same<T, float>::value) return __builtin_fminf(a, b);
  else return b < a ? b : a;
}
template <typename T>
__device__ __forceinline__ T fold_max(T a, T b) {
  if constexpr (std::is_same<T, double>::value) return __builtin_fmax(a, b);
  else if constexpr (std::is_same<T, float>::value) return __builtin_fmaxf(a, b);
  else return b > a ? b : a;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_any(T v, int mask) {
  if constexpr (sizeof(T) == 8) {
    uint64_t u;
    __builtin_memcpy(&u, &v, 8);
    uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
    lo = __shfl_xor((int)lo, mask, 64);
    hi = __shfl_xor((int)hi, mask, 64);
    u = ((uint64_t)hi << 32) | lo;
    __builtin_memcpy(&v, &u, 8);
    return v;
  } else {
    uint32_t u = 0;
    __builtin_memcpy(&u, &v, sizeof(T));
    u = __shfl_xor((int)u, mask, 64);
    __builtin_memcpy(&v, &u, sizeof(T));
    return v;
  }
}

// Block-wide (BLK threads) reduction of NV min-values and NV max-values; result valid in thread 0.
// `scratch` must hold 4 * 2 * NV elements of T.
template <typename T, int NV, int BLK = kBlock>
__device__ __forceinline__ void block_reduce_minmax(T (&mn)[NV], T (&mx)[NV], T* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      mn[i] = fold_min(mn[i], shfl_xor_any(mn[i], off));
      mx[i] = fold_max(mx[i], shfl_xor_any(mx[i], off));
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      scratch[wave * 2 * NV + i] = mn[i];
      scratch[wave * 2 * NV + NV + i] = mx[i];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < BLK / 64; ++w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        mn[i] = fold_min(mn[i], scratch[w * 2 * NV + i]);
        mx[i] = fold_max(mx[i], scratch[w * 2 * NV + NV + i]);
      }
    }
  }
}

}  // namespace pstd
