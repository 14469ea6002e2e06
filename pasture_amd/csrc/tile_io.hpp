// Tile staging helpers shared by the LDS-tiled kernels (convert.hip, las_encode.hip).
#pragma once
#include "device_common.hpp"

namespace pstd {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16-byte vector (dwordx4 / b128 accesses)
typedef PST_AS_GLOBAL u32x4* g4ptr_t;
typedef const PST_AS_GLOBAL u32x4* cg4ptr_t;
typedef PST_AS_LDS u32x4* l4ptr_t;
typedef const PST_AS_LDS u32x4* cl4ptr_t;

// Coalesced global -> LDS copy of the 16-byte aligned span [gbase, gbase + nbytes16) with LDS-DMA
// (`global_load_lds_dwordx4`, gfx950): every lane names its own 16 global bytes, the wave's 1 KiB lands at the
// wave-uniform LDS base + lane*16 without a VGPR round trip, so all of a wave's loads are in flight at once.
// Completion is tracked by vmcnt: callers must `s_waitcnt vmcnt(0)` before the barrier that publishes the tile.
template <int BLK>
__device__ __forceinline__ void tile_load(lptr_t lds, cgptr_t gbase, uint32_t nbytes16) {
  const uint32_t nvec = nbytes16 >> 4;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i = threadIdx.x; i < nvec; i += BLK) {
    __builtin_amdgcn_global_load_lds((const PST_AS_GLOBAL void*)(gbase + (uint64_t)i * 16), (PST_AS_LDS void*)(lds + (i - lane) * 16u), 16, 0, 0);
  }
}
__device__ __forceinline__ void wait_tile_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS -> global copy of the bytes [mis, mis + nbytes) of the staged span; 16-byte stores for whole chunks, byte stores
// on the two ragged edges so that no byte outside the target range is ever written.
template <int BLK>
__device__ __forceinline__ void tile_store(clptr_t lds, gptr_t gbase, uint32_t mis, uint32_t nbytes) {
  const uint32_t end = mis + nbytes;
  const uint32_t v_first = (mis + 15u) >> 4, v_last = end >> 4;  // whole 16-byte chunks [v_first, v_last)
  cl4ptr_t l = reinterpret_cast<cl4ptr_t>(lds);
  g4ptr_t g = reinterpret_cast<g4ptr_t>(gbase);
  constexpr uint32_t kBatch = 4;  // LDS reads in flight per lane before the first store
  for (uint32_t i0 = v_first + threadIdx.x; i0 < v_last; i0 += kBatch * BLK) {
    u32x4 v[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) if (i0 + u * BLK < v_last) v[u] = l[i0 + u * BLK];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) if (i0 + u * BLK < v_last) __builtin_nontemporal_store(v[u], &g[i0 + u * BLK]);
  }
  // ragged edges: bytes [mis, v_first*16) and [v_last*16, end) — never touch a byte outside the target range
  const uint32_t head_end = v_first * 16u < end ? v_first * 16u : end;
  for (uint32_t b = mis + threadIdx.x; b < head_end; b += BLK) gbase[b] = lds[b];
  const uint32_t tail_begin = v_last * 16u > head_end ? v_last * 16u : head_end;
  for (uint32_t b = tail_begin + threadIdx.x; b < end; b += BLK) gbase[b] = lds[b];
}


// A packed record as a little-endian byte string in dword registers, assembled at compile-time offsets and written to an LDS record tile.
template <int NB>
struct RecordImage {
  uint32_t w[(NB + 3) / 4 + 1] = {};
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
  // The same bytes with naturally aligned LDS stores only (device_common.hpp: an LDS access that is not naturally aligned stalls the pipe):
  // the image is shifted in registers to the dword grid of its target (v_alignbyte with the lane's own phase), the dwords that are covered for
  // every phase go out as aligned b32 stores, the ragged head and tail as b16 / b8 pieces.  m = 1..4 bytes of dword 0 precede the record
  // (m = 4: the record starts on a dword boundary and dword 0 is not written at all).
  __device__ __forceinline__ void store_aligned(lptr_t p) const {
    if constexpr (NB < 3) {  // (the head pieces below assume the record reaches the end of its first dword)
      store(p);
      return;
    }
    typedef PST_AS_LDS uint8_t* p8;
    typedef PST_AS_LDS uint16_t* p16;
    typedef PST_AS_LDS uint32_t* p32;
    constexpr int NW = (NB + 3) / 4 + 1;      // dwords of w (the last one is zero padding)
    constexpr int NO = (NB + 7) / 4;          // dwords of the shifted image: bytes [0, 4 NO) hold the record at [m, m + NB)
    constexpr int KF = (NB - 3) / 4;          // dwords 1 .. KF are covered whatever the phase
    const uint32_t ph = (uint32_t)(uintptr_t)p & 3u, m = ph ? ph : 4u;
    lptr_t pa = p - m;
    const uint32_t sh = 4u - m;               // out[k] = bytes [4 k - m, 4 k - m + 4) of the record
    uint32_t o[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) o[k] = __builtin_amdgcn_alignbyte(k < NW ? w[k] : 0u, (k >= 1 && k - 1 < NW) ? w[k - 1] : 0u, sh);
    if (m & 1u) *(p8)(pa + m) = (uint8_t)(o[0] >> (8u * m));
    if (m <= 2u) *(p16)(pa + 2) = (uint16_t)(o[0] >> 16);
#pragma unroll
    for (int k = 1; k <= KF; ++k) *(p32)(pa + 4 * k) = o[k];
#pragma unroll
    for (int k = KF + 1; k < NO; ++k) {
      const int c = (int)m + NB - 4 * k;      // record bytes in dword k
      if (c >= 4) *(p32)(pa + 4 * k) = o[k];
      else if (c > 0) {
        if (c & 2) *(p16)(pa + 4 * k) = (uint16_t)o[k];
        if (c & 1) *(p8)(pa + 4 * k + (c & 2)) = (uint8_t)(o[k] >> (8 * (c & 2)));
      }
    }
  }
};

}  // namespace pstd
