// Compile-time facts about the LAS point formats shared by the encoder (las_encode.hip) and the decoder (las_decode.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "tile_io.hpp"

namespace pstlas {

constexpr int kMaxAttrs = 24;

struct Fmt { bool ext, gps, color, nir, wave; };
__host__ __device__ constexpr Fmt fmt_of(int n) {
  return Fmt{n >= 6, n == 1 || n == 3 || n == 4 || n == 5 || n >= 6, n == 2 || n == 3 || n == 5 || n == 7 || n == 8 || n == 10, n == 8 || n == 10,
             n == 4 || n == 5 || n == 9 || n == 10};
}
__host__ __device__ constexpr uint32_t raw_size(Fmt f) {
  return (f.ext ? 30u : 20u) + (f.gps && !f.ext ? 8u : 0u) + (f.color ? 6u : 0u) + (f.nir ? 2u : 0u) + (f.wave ? 29u : 0u);
}

// LasPointFormatN::layout() is packed in field order (las_types.rs), so slot offsets are prefix sums of the field sizes.
__host__ __device__ constexpr uint32_t typed_slot_offset(Fmt f, int slot) {
  uint32_t sizes[kMaxAttrs] = {};
  int n = 0;
  sizes[n++] = 24; sizes[n++] = 2; sizes[n++] = 1; sizes[n++] = 1;
  if (f.ext) { sizes[n++] = 1; sizes[n++] = 1; }
  sizes[n++] = 1; sizes[n++] = 1; sizes[n++] = 1;
  if (f.ext) { sizes[n++] = 1; sizes[n++] = 2; } else { sizes[n++] = 1; sizes[n++] = 1; }
  sizes[n++] = 2;
  if (f.gps) sizes[n++] = 8;
  if (f.color) sizes[n++] = 6;
  if (f.nir) sizes[n++] = 2;
  if (f.wave) { sizes[n++] = 1; sizes[n++] = 8; sizes[n++] = 4; sizes[n++] = 4; sizes[n++] = 12; }
  uint32_t o = 0;
  for (int i = 0; i < slot && i < n; ++i) o += sizes[i];
  return o;
}
__host__ __device__ constexpr uint32_t typed_size(Fmt f) { return typed_slot_offset(f, kMaxAttrs); }


using namespace pstd;

// ---- register-level building blocks shared by the LAS kernels (encoder, decoder, transposer) ------------------------------
// ---- columnar fast path: four consecutive points per lane ------------------------------------------------------------
// A column of B-byte values is read as one 4*B-byte vector per lane (wave = 256 consecutive points, fully coalesced);
// the words are then cut apart with compile-time shifts.  QuadCol<B>::w holds the 4 values of this lane's points.
template <int B>
struct QuadCol {
  uint32_t w[B + 2];  // 4 * B bytes + two zero words so that bytes_at() needs no guards
  __device__ __forceinline__ void load(cgptr_t p) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    w[B] = 0; w[B + 1] = 0;
    constexpr int K4 = B / 4 * 4, K2 = K4 + ((B - K4) >= 2 ? 2 : 0);
#pragma unroll
    for (int k = 0; k < K4; k += 4) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x4>::type*>(p + 4 * k));
      w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
    if constexpr (K2 > K4) {
      const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL Unaligned<u32x2>::type*>(p + 4 * K4));
      w[K4] = v.x; w[K4 + 1] = v.y;
    }
    if constexpr (B > K2) w[K2] = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL Unaligned<uint32_t>::type*>(p + 4 * K2));
  }
  // up to 8 bytes starting at byte `off` of the 4*B-byte vector (compile-time `off` after unrolling)
  __device__ __forceinline__ uint64_t bytes_at(int off) const {
    const int wi = off >> 2, sh = (off & 3) * 8;
    const uint64_t lo = w[wi], mid = w[wi + 1], hi = w[wi + 2];
    const uint64_t v = lo | (mid << 32);
    return sh == 0 ? v : ((v >> sh) | (hi << (64 - sh)));
  }
  __device__ __forceinline__ uint64_t value(int t) const {  // point t's B bytes (B <= 8), zero-extended
    const uint64_t v = bytes_at(t * B);
    return B >= 8 ? v : (v & ((1ull << (8 * (B & 7))) - 1ull));
  }
};

// The non-position bytes of one record, assembled in registers at compile-time offsets and written with word stores.
template <int NB>
struct RecTail {
  uint32_t w[(NB + 3) / 4] = {};
  __device__ __forceinline__ void put(int off, int nbytes, uint64_t v) {  // v zero-extended to 8 bytes
    const int wi = off >> 2, sh = (off & 3) * 8;
    w[wi] |= (uint32_t)(v << sh);
    if (sh + 8 * nbytes > 32) w[wi + 1] |= (uint32_t)(sh == 0 ? (v >> 32) : (v >> (32 - sh)));
    if (sh + 8 * nbytes > 64) w[wi + 2] |= (uint32_t)(v >> (64 - sh));
  }
  __device__ __forceinline__ void store(lptr_t p) const {
    int k = 0;
#pragma unroll
    for (; 4 * k + 4 <= NB; ++k) store_un<uint32_t>(p + 4 * k, w[k]);
    if (NB - 4 * k >= 2) { store_un<uint16_t>(p + 4 * k, (uint16_t)w[k]); if (NB - 4 * k == 3) store_un<uint8_t>(p + 4 * k + 2, (uint8_t)(w[k] >> 16)); }
    else if (NB - 4 * k == 1) store_un<uint8_t>(p + 4 * k, (uint8_t)w[k]);
  }
};

// N bytes at an arbitrarily aligned LDS address as aligned dwords re-aligned in registers (r[0] = bytes 0..3, ...)
template <int N>
struct LdsBytes {
  static constexpr int NW = (N + 3) / 4;
  uint32_t r[NW + 2];
  __device__ __forceinline__ explicit LdsBytes(clptr_t p) {
    const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
    const PST_AS_LDS uint32_t* q = (const PST_AS_LDS uint32_t*)(p - m);
    uint32_t d[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], m);
    r[NW] = 0; r[NW + 1] = 0;
  }
  __device__ __forceinline__ uint64_t at(int off) const {  // 8 bytes starting at byte `off` (compile-time after unrolling)
    const int wi = off >> 2, sh = (off & 3) * 8;
    const uint64_t lo = r[wi], mid = r[wi + 1], hi = r[wi + 2];
    const uint64_t v = lo | (mid << 32);
    return sh == 0 ? v : ((v >> sh) | (hi << (64 - sh)));
  }
};

// The values of four consecutive points of a B-byte attribute, stored with as few vector stores as possible.
template <int B>
struct Pack4 {
  uint32_t w[B] = {};
  __device__ __forceinline__ void put_at(int off, int nbytes, uint64_t v) {  // v zero-extended
    const int wi = off >> 2, sh = (off & 3) * 8;
    w[wi] |= (uint32_t)(v << sh);
    if (sh + 8 * nbytes > 32) w[wi + 1] |= (uint32_t)(sh == 0 ? (v >> 32) : (v >> (32 - sh)));
    if (sh + 8 * nbytes > 64) w[wi + 2] |= (uint32_t)(v >> (64 - sh));
  }
  __device__ __forceinline__ void put(int t, uint64_t v) { put_at(t * B, B < 8 ? B : 8, B >= 8 ? v : (v & ((1ull << (8 * (B & 7))) - 1ull))); }
  __device__ __forceinline__ void store(gptr_t dst) const {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    constexpr int K4 = B / 4 * 4, K2 = K4 + ((B - K4) >= 2 ? 2 : 0);
#pragma unroll
    for (int k = 0; k < K4; k += 4) {
      u32x4 v; v.x = w[k]; v.y = w[k + 1]; v.z = w[k + 2]; v.w = w[k + 3];
      __builtin_nontemporal_store(v, reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x4>::type*>(dst + 4 * k));
    }
    if constexpr (K2 > K4) {
      u32x2 v; v.x = w[K4]; v.y = w[K4 + 1];
      __builtin_nontemporal_store(v, reinterpret_cast<PST_AS_GLOBAL Unaligned<u32x2>::type*>(dst + 4 * K4));
    }
    if constexpr (B > K2) __builtin_nontemporal_store(w[K2], reinterpret_cast<PST_AS_GLOBAL Unaligned<uint32_t>::type*>(dst + 4 * K2));
  }
};

using pstd::RecordImage;  // (tile_io.hpp)


}  // namespace pstlas
