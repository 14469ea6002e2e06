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


}  // namespace pstd
