// Predicate compaction for gfx950: HashMapBuffer::filter_into / filter
// (pasture-core/src/containers/point_buffer.rs:1064-1136) with the predicate given as a byte mask
// (benches/buffer_filter_bench.rs:62-64: `|idx| random_matches[idx]`).
//
// The reference walks the matching indices once per attribute; here the order-preserving rank of every selected point is
// computed once (count per tile -> exclusive scan of the tile counts -> ranks inside the tile) and every attribute of the
// tile's selected points is copied in the same launch:
//
//   mask_count_kernel     one tile of the mask per block: number of non-zero bytes                (1 B/pt read)
//   tile_scan_kernel      exclusive scan of the tile counts (one block; <= 10^5 values)           (negligible)
//   filter_stream_body<P> (filter_stream.hpp; any layout of at most 64 bytes per point, either target kind) the columns are READ like a
//                         conversion reads them -- a lane owns four consecutive points -- and selected points go to an LDS record tile / to LDS
//                         column spans at their rank; P in-tree for the bench layout and typed LAS-0 points, otherwise compiled at run time
//   filter_scatter_kernel per tile: selected local indices compacted into LDS (order preserved), then attribute by
//                         attribute: gather from the source columns, store to the tile's contiguous output span —
//                         columnar targets directly (coalesced, narrow values packed four/two per dword), interleaved
//                         targets through an LDS record tile and 16-byte stores.
// HBM-bound gather/scatter: no MFMA.
#include "device_common.hpp"
#include "kernels.hpp"
#include "las_device.hpp"
#include "tile_io.hpp"
#include "filter_stream.hpp"

#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <string>

#include "jit.hpp"

using namespace pstd;

namespace {

__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) {  // number of non-zero bytes of a dword
  const uint32_t t = (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
  return (uint32_t)__builtin_popcount(t);
}

// One WAVE per tile, kCountTilesPerWave consecutive tiles per wave: no LDS, no barrier, and 32 KiB of mask per block instead of 2 KiB
// (48,829 blocks of one 2 KiB tile each: 36 us per 10^8 mask bytes; now 26 us).
constexpr uint32_t kCountTilesPerWave = 4;
__global__ __launch_bounds__(kBlock) void mask_count_kernel(const uint8_t* __restrict__ mask, uint64_t n, uint32_t tile, uint32_t n_tiles,
                                                            uint32_t* __restrict__ counts) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
  if (tile == 2048u && (uint64_t)(wave + 1u) * kCountTilesPerWave * 2048u <= n) {
    // four whole 2048-byte tiles: all eight 16-byte loads of the lane in flight before the first count (23.1 -> 20.2 us per 10^8 mask bytes)
    cgptr_t m = (cgptr_t)((uint64_t)(uintptr_t)mask + (uint64_t)wave * kCountTilesPerWave * 2048u);
    u32x4 v[2 * kCountTilesPerWave];
#pragma unroll
    for (uint32_t q = 0; q < 2 * kCountTilesPerWave; ++q) v[q] = load_un<u32x4>(m + 1024u * q + 16u * lane);
#pragma unroll
    for (uint32_t u = 0; u < kCountTilesPerWave; ++u) {
      uint32_t c = 0;
#pragma unroll
      for (uint32_t h = 0; h < 2; ++h) { const u32x4 x = v[2 * u + h]; c += nonzero_bytes(x.x) + nonzero_bytes(x.y) + nonzero_bytes(x.z) + nonzero_bytes(x.w); }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
      if (lane == 0) counts[wave * kCountTilesPerWave + u] = c;
    }
    return;
  }
  for (uint32_t u = 0; u < kCountTilesPerWave; ++u) {
    const uint32_t t = wave * kCountTilesPerWave + u;
    if (t >= n_tiles) return;
    const uint64_t first = (uint64_t)t * tile;
    const uint32_t cnt = (uint32_t)((n - first) < tile ? (n - first) : tile);
    cgptr_t m = (cgptr_t)((uint64_t)(uintptr_t)mask + first);
    uint32_t c = 0;
    const uint32_t nvec = cnt >> 4;
    for (uint32_t i = lane; i < nvec; i += 64u) {
      const u32x4 v = load_un<u32x4>(m + 16u * i);
      c += nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
    }
    for (uint32_t i = (nvec << 4) + lane; i < cnt; i += 64u) c += m[i] != 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
    if (lane == 0) counts[t] = c;
  }
}

// offsets[t] = sum of counts[0..t); offsets[n_tiles] = total.  One block; every thread takes kScanPer consecutive counts per round
// (one 16-byte load), so 10^8 points (48,829 tiles) need 12 rounds of a 1024-thread scan (16 counts per thread and 3 rounds measured
// slower: the 16 strided 8-byte stores per thread cost more than the saved rounds).
constexpr uint32_t kScanPer = 4;
// The counts of kScanAhead rounds are loaded before the first of them is scanned: a round's 16-byte load was the round's latency (12 dependent
// rounds for 10^8 points: 28 us; with the loads of eight rounds in flight at once a round is shuffles, one barrier and the stores; sixteen
// rounds ahead spill at the 128 registers a 1024-thread block has).
constexpr uint32_t kScanAhead = 8;
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint32_t* __restrict__ counts, uint32_t n_tiles, unsigned long long* __restrict__ offsets,
                                                         unsigned long long* __restrict__ total_also) {  // total_also: a caller's word that receives the total too (or null)
  __shared__ unsigned long long wave_tot[2][16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long carry = 0;
  uint32_t round = 0;
  for (uint32_t base0 = 0; base0 < n_tiles; base0 += 1024u * kScanPer * kScanAhead) {
    uint32_t c[kScanAhead][kScanPer];
#pragma unroll
    for (uint32_t a = 0; a < kScanAhead; ++a) {
      const uint32_t i0 = base0 + a * 1024u * kScanPer + threadIdx.x * kScanPer;
#pragma unroll
      for (uint32_t k = 0; k < kScanPer; ++k) c[a][k] = 0;
      if (i0 + kScanPer <= n_tiles) {
        const u32x4 v = load_un<u32x4>((cgptr_t)(counts + i0));
        c[a][0] = v.x; c[a][1] = v.y; c[a][2] = v.z; c[a][3] = v.w;
      } else {
#pragma unroll
        for (uint32_t k = 0; k < kScanPer; ++k) if (i0 + k < n_tiles) c[a][k] = counts[i0 + k];
      }
    }
#pragma unroll
    for (uint32_t a = 0; a < kScanAhead; ++a) {
      const uint32_t base = base0 + a * 1024u * kScanPer;
      if (base < n_tiles) {  // (uniform)
      const uint32_t i0 = base + threadIdx.x * kScanPer;
      unsigned long long v = 0;
#pragma unroll
      for (uint32_t k = 0; k < kScanPer; ++k) v += c[a][k];
      unsigned long long incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t l = (uint32_t)__shfl_up((int)(uint32_t)incl, off, 64), h = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off, 64);
        if ((int)lane >= off) incl += ((unsigned long long)h << 32) | l;
      }
      unsigned long long (&tot)[16] = wave_tot[round & 1u];  // double-buffered: one barrier per round
      ++round;
      if (lane == 63) tot[wave] = incl;
      __syncthreads();
      unsigned long long before = carry + incl - v, all = 0;
#pragma unroll
      for (uint32_t w = 0; w < 16; ++w) { if (w < wave) before += tot[w]; all += tot[w]; }
#pragma unroll
      for (uint32_t k = 0; k < kScanPer; ++k) { if (i0 + k < n_tiles) offsets[i0 + k] = before; before += c[a][k]; }
      carry += all;
      }
    }
  }
  if (threadIdx.x == 0) { offsets[n_tiles] = carry; if (total_also) *total_also = carry; }
}

// The same scan by MANY blocks without any hand-over between them: block b owns the 1024 tile counts [1024 b, 1024 b + 1024) and finds its own base
// by summing every count before them (the last block of a 10^8-point call reads 48 k counts = 190 KiB from L2; all blocks together 4.6 MB) -- no
// flags, no atomics, no ordering assumption, and the 12 dependent rounds of the one-block scan (24.5 us) become one block-wide reduction and one
// block-wide scan per block, all blocks at once.  The work grows with the square of the tile count: taken up to 2^19 tiles (10^9 points).
constexpr uint32_t kScanBlockTiles = 1024;
__global__ __launch_bounds__(kScanBlockTiles) void tile_scan_blocks_kernel(const uint32_t* __restrict__ counts, uint32_t n_tiles, unsigned long long* __restrict__ offsets,
                                                                           unsigned long long* __restrict__ total_also) {
  __shared__ unsigned long long before_tot[16], own_tot[16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t first = blockIdx.x * kScanBlockTiles;
  unsigned long long s = 0;
  for (uint32_t i = threadIdx.x * 4u; i < first; i += kScanBlockTiles * 4u) {  // (first is a multiple of 1024: whole 16-byte groups)
    const u32x4 v = load_un<u32x4>((cgptr_t)(counts + i));
    s += (unsigned long long)v.x + v.y + v.z + v.w;
  }
  const uint32_t t = first + threadIdx.x;
  const uint32_t c = t < n_tiles ? counts[t] : 0u;
  uint32_t incl = c;  // (a block's own counts sum to at most 1024 x 2048 points)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += o;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t l = (uint32_t)__shfl_xor((int)(uint32_t)s, off, 64), h = (uint32_t)__shfl_xor((int)(uint32_t)(s >> 32), off, 64);
    s += ((unsigned long long)h << 32) | l;
  }
  if (lane == 63) { own_tot[wave] = incl; before_tot[wave] = s; }
  __syncthreads();
  unsigned long long base = 0, own_before = 0, own_all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) { base += before_tot[w]; if (w < wave) own_before += own_tot[w]; own_all += own_tot[w]; }
  if (t < n_tiles) offsets[t] = base + own_before + incl - c;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { offsets[n_tiles] = base + own_all; if (total_also) *total_also = base + own_all; }
}

using pstf::kMaxFilterAttrs;
using pstf::FilterAttr;
using pstf::FilterArgs;

template <typename U>
__device__ __forceinline__ void copy_granules(const FilterAttr& a, const uint16_t* sel, uint64_t first, uint64_t out0, uint32_t m, bool dst_aos, lptr_t lds,
                                              uint32_t mis, uint32_t dst_stride) {
  constexpr uint32_t kBatch = 4;
  const uint32_t total = m * a.cnt;
  cgptr_t src = (cgptr_t)as_global(a.src);
  gptr_t dst = as_global(a.dst) + out0 * a.cnt * sizeof(U);
  constexpr uint32_t PACK = sizeof(U) >= 4 ? 1u : 4u / (uint32_t)sizeof(U);  // narrow granules travel four / two per dword
  if constexpr (PACK > 1) if (!dst_aos) {
    // columnar target, 1- or 2-byte granules: each lane produces one dword of PACK consecutive granules
    const uint32_t head = (uint32_t)((0u - (uint32_t)(uintptr_t)dst) & 3u) / (uint32_t)sizeof(U);  // granules before the first aligned dword
    const uint32_t h = head < total ? head : total;
    auto fetch = [&](uint32_t k) -> U {
      uint32_t j, c;
      if (a.cnt == 1) { j = k; c = 0; } else if (a.cnt == 3) { j = (k * 43691u) >> 17; c = k - 3u * j; } else { j = k / a.cnt; c = k - j * a.cnt; }
      return load_un<U>(src + ((first + sel[j]) * a.src_stride + c * (uint32_t)sizeof(U)));
    };
    if (threadIdx.x < h) store_un<U>(dst + threadIdx.x * sizeof(U), fetch(threadIdx.x));
    const uint32_t nd = (total - h) / PACK;
    for (uint32_t q = threadIdx.x; q < nd; q += kBlock) {
      uint32_t packed = 0;
#pragma unroll
      for (uint32_t i = 0; i < PACK; ++i) packed |= (uint32_t)fetch(h + q * PACK + i) << (8u * (uint32_t)sizeof(U) * i);
      store_un<uint32_t>(dst + (h + q * PACK) * sizeof(U), packed);
    }
    for (uint32_t k = h + nd * PACK + threadIdx.x; k < total; k += kBlock) store_un<U>(dst + k * sizeof(U), fetch(k));
    return;
  }
  for (uint32_t k0 = threadIdx.x; k0 < total; k0 += kBatch * kBlock) {
    U v[kBatch];
    uint32_t jj[kBatch], cc[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) {
      const uint32_t k = k0 + u * kBlock;
      uint32_t j = 0, c = 0;
      if (k < total) {
        if (a.cnt == 1) { j = k; } else if (a.cnt == 3) { j = (k * 43691u) >> 17; c = k - 3u * j; } else { j = k / a.cnt; c = k - j * a.cnt; }
        v[u] = load_un<U>(src + ((first + sel[j]) * a.src_stride + c * (uint32_t)sizeof(U)));
      }
      jj[u] = j; cc[u] = c;
    }
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) {
      const uint32_t k = k0 + u * kBlock;
      if (k < total) {
        if (dst_aos) {
          // unaligned LDS stores stall the LDS pipeline (SQ_LDS_UNALIGNED_STALL): split by the alignment class of the address
          lptr_t q = lds + (mis + jj[u] * dst_stride + a.dst_off + cc[u] * (uint32_t)sizeof(U));
          if constexpr (sizeof(U) == 16) {
            lds_store<uint64_t>(q, (uint64_t)v[u].x | ((uint64_t)v[u].y << 32));
            lds_store<uint64_t>(q + 8, (uint64_t)v[u].z | ((uint64_t)v[u].w << 32));
          } else {
            lds_store<U>(q, v[u]);
          }
        }
        else store_un<U>(dst + (uint64_t)k * sizeof(U), v[u]);
      }
    }
  }
}

template <int PPL, bool DST_AOS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(DST_AOS ? 8 : 4, 8))) void filter_scatter_kernel(const FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  uint16_t* sel = (uint16_t*)lds_raw;                         // [tile] local indices of the selected points, ascending
  lptr_t lds = (lptr_t)lds_raw + ((a.tile * 2u + 15u) & ~15u);  // interleaved target: record tile
  __shared__ uint32_t wave_tot[kBlock / 64];

  const uint32_t tile_id = blockIdx.x + a.tile0;
  const uint64_t first = (uint64_t)tile_id * a.tile;
  const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
  const uint64_t out0 = a.offsets[tile_id];
  uint32_t m = a.counts[tile_id];
  if (m == 0 || out0 >= a.limit) return;

  // ranks: lane t owns the PPL consecutive points t*PPL ..
  const uint32_t p0 = threadIdx.x * PPL;
  uint8_t mb[PPL];
  cgptr_t mp = (cgptr_t)((uint64_t)(uintptr_t)a.mask + first);
  if (p0 + PPL <= cnt) {
    if constexpr (PPL == 8) { const uint64_t w = load_un<uint64_t>(mp + p0); for (int i = 0; i < 8; ++i) mb[i] = (uint8_t)(w >> (8 * i)); }
    else if constexpr (PPL == 4) { const uint32_t w = load_un<uint32_t>(mp + p0); for (int i = 0; i < 4; ++i) mb[i] = (uint8_t)(w >> (8 * i)); }
    else if constexpr (PPL == 2) { const uint16_t w = load_un<uint16_t>(mp + p0); mb[0] = (uint8_t)w; mb[1] = (uint8_t)(w >> 8); }
    else mb[0] = mp[p0];
  } else {
#pragma unroll
    for (int i = 0; i < PPL; ++i) mb[i] = p0 + i < cnt ? mp[p0 + i] : (uint8_t)0;
  }
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < PPL; ++i) c += mb[i] != 0;
  uint32_t incl = c;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t r = incl - c;
  for (uint32_t w = 0; w < wave; ++w) r += wave_tot[w];
#pragma unroll
  for (int i = 0; i < PPL; ++i) if (mb[i] != 0) sel[r++] = (uint16_t)(p0 + i);

  if (out0 + m > a.limit) m = (uint32_t)(a.limit - out0);  // more matches than num_matches: the host raises the panic
  m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);  // block-uniform by construction: keep the chunk loop on the scalar unit
  if constexpr (DST_AOS) {
    // The tile's selected records go through the LDS record tile in balanced chunks of <= a.chunk records: ranks and `sel` are
    // computed once per 2048 input points while the LDS footprint (and with it the number of resident blocks) stays small.
    const uint32_t nch = (m + a.chunk - 1) / a.chunk;
    const uint32_t mc = ((m + nch - 1) / nch + 15u) & ~15u;  // <= a.chunk (a multiple of 16)
    for (uint32_t j0 = 0; j0 < m; j0 += mc) {
      const uint32_t cm = (m - j0) < mc ? (m - j0) : mc;
      const uint64_t ga = a.dst_aos + (out0 + j0) * a.dst_stride;
      const uint32_t mis = (uint32_t)(ga & 15u);
      if (!a.dst_covered) {  // padding bytes of the target records must survive
        tile_load<kBlock>(lds, as_global(ga - mis), (mis + cm * a.dst_stride + 15u) & ~15u);
        wait_tile_loads();
      }
      __syncthreads();
      for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
        const FilterAttr& at = a.attrs[ai];
        switch (at.unit) {
          case 16: copy_granules<u32x4>(at, sel + j0, first, 0, cm, true, lds, mis, a.dst_stride); break;
          case 8: copy_granules<uint64_t>(at, sel + j0, first, 0, cm, true, lds, mis, a.dst_stride); break;
          case 4: copy_granules<uint32_t>(at, sel + j0, first, 0, cm, true, lds, mis, a.dst_stride); break;
          case 2: copy_granules<uint16_t>(at, sel + j0, first, 0, cm, true, lds, mis, a.dst_stride); break;
          default: copy_granules<uint8_t>(at, sel + j0, first, 0, cm, true, lds, mis, a.dst_stride); break;
        }
      }
      __syncthreads();
      tile_store<kBlock>(lds, as_global(ga - mis), mis, cm * a.dst_stride);
      if (j0 + mc < m) __syncthreads();  // the next chunk overwrites the record tile
    }
  } else {
    __syncthreads();
    for (uint32_t ai = 0; ai < a.n_attrs; ++ai) {
      const FilterAttr& at = a.attrs[ai];
      switch (at.unit) {
        case 16: copy_granules<u32x4>(at, sel, first, out0, m, false, lds, 0, 0); break;
        case 8: copy_granules<uint64_t>(at, sel, first, out0, m, false, lds, 0, 0); break;
        case 4: copy_granules<uint32_t>(at, sel, first, out0, m, false, lds, 0, 0); break;
        case 2: copy_granules<uint16_t>(at, sel, first, out0, m, false, lds, 0, 0); break;
        default: copy_granules<uint8_t>(at, sel, first, out0, m, false, lds, 0, 0); break;
      }
    }
  }
}

// ---- compile-time attribute lists (see static_plans.hpp for the idea): the layout of the reference's own filter bench
// (buffer_filter_bench.rs:71-74: CustomPointTypeBig, HashMapBuffer source) with every size, granule and record offset an immediate ----
struct StaticFilterAttr { uint32_t src_stride, dst_off, unit, cnt; };
struct BigFilterPlan {
  static constexpr int n = 5;
  static constexpr uint32_t dst_stride = 41;
  __host__ __device__ static constexpr StaticFilterAttr attr(int i) {
    // GpsTime F64 @0 | ColorRGB Vec3u16 @8 | Position3D Vec3f64 @14 | Classification U8 @38 | Intensity I16 @39; columnar source: stride = size
    constexpr StaticFilterAttr t[n] = {{8, 0, 8, 1}, {6, 8, 2, 3}, {24, 14, 8, 3}, {1, 38, 1, 1}, {2, 39, 2, 1}};
    return t[i];
  }
};

template <int PPL, bool DST_AOS, typename SP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(DST_AOS ? 8 : 4, 8))) void filter_scatter_static_kernel(const FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  uint16_t* sel = (uint16_t*)lds_raw;
  lptr_t lds = (lptr_t)lds_raw + ((a.tile * 2u + 15u) & ~15u);
  __shared__ uint32_t wave_tot[kBlock / 64];
  const uint64_t first = (uint64_t)blockIdx.x * a.tile;
  const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
  const uint64_t out0 = a.offsets[blockIdx.x];
  uint32_t m = a.counts[blockIdx.x];
  if (m == 0 || out0 >= a.limit) return;
  const uint32_t p0 = threadIdx.x * PPL;
  uint8_t mb[PPL];
  cgptr_t mp = (cgptr_t)((uint64_t)(uintptr_t)a.mask + first);
  if (p0 + PPL <= cnt) {
    static_assert(PPL == 8, "static filter plans use the 2048-point tile");
    const uint64_t w = load_un<uint64_t>(mp + p0);
    for (int i = 0; i < 8; ++i) mb[i] = (uint8_t)(w >> (8 * i));
  } else {
#pragma unroll
    for (int i = 0; i < PPL; ++i) mb[i] = p0 + i < cnt ? mp[p0 + i] : (uint8_t)0;
  }
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < PPL; ++i) c += mb[i] != 0;
  uint32_t incl = c;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t r = incl - c;
  for (uint32_t w = 0; w < wave; ++w) r += wave_tot[w];
#pragma unroll
  for (int i = 0; i < PPL; ++i) if (mb[i] != 0) sel[r++] = (uint16_t)(p0 + i);
  if (out0 + m > a.limit) m = (uint32_t)(a.limit - out0);
  m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
  auto copy_attr = [&](int ai, const uint16_t* s, uint64_t o0, uint32_t mm, uint32_t mis) __attribute__((always_inline)) {
    const StaticFilterAttr sa = SP::attr(ai);
    FilterAttr at;
    at.src = a.attrs[ai].src; at.dst = a.attrs[ai].dst;
    at.src_stride = sa.src_stride; at.dst_off = sa.dst_off; at.unit = sa.unit; at.cnt = sa.cnt;
    if (sa.unit == 8) copy_granules<uint64_t>(at, s, first, o0, mm, DST_AOS, lds, mis, SP::dst_stride);
    else if (sa.unit == 2) copy_granules<uint16_t>(at, s, first, o0, mm, DST_AOS, lds, mis, SP::dst_stride);
    else copy_granules<uint8_t>(at, s, first, o0, mm, DST_AOS, lds, mis, SP::dst_stride);
  };
  if constexpr (DST_AOS) {
    const uint32_t nch = (m + a.chunk - 1) / a.chunk;
    const uint32_t mc = ((m + nch - 1) / nch + 15u) & ~15u;
    for (uint32_t j0 = 0; j0 < m; j0 += mc) {
      const uint32_t cm = (m - j0) < mc ? (m - j0) : mc;
      const uint64_t ga = a.dst_aos + (out0 + j0) * SP::dst_stride;
      const uint32_t mis = (uint32_t)(ga & 15u);
      __syncthreads();
#pragma unroll
      for (int ai = 0; ai < SP::n; ++ai) copy_attr(ai, sel + j0, 0, cm, mis);
      __syncthreads();
      tile_store<kBlock>(lds, as_global(ga - mis), mis, cm * SP::dst_stride);
      if (j0 + mc < m) __syncthreads();
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int ai = 0; ai < SP::n; ++ai) copy_attr(ai, sel, out0, m, 0);
  }
}

// Point-major form for the same layout: ONE lane gathers the seven pieces of a selected point (all loads in flight together) and writes
// one 41-byte record image, assembled in registers at compile-time offsets, into the LDS record tile.  Nine granule copies per point
// (each with its own index arithmetic and LDS store) become seven loads and eleven dword stores.
template <int PPL>
__global__ __launch_bounds__(kBlock) void filter_big_records_kernel(const FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  uint16_t* sel = (uint16_t*)lds_raw;
  lptr_t lds = (lptr_t)lds_raw + ((a.tile * 2u + 15u) & ~15u);
  __shared__ uint32_t wave_tot[kBlock / 64];
  constexpr uint32_t STRIDE = 41;
  const uint64_t first = (uint64_t)blockIdx.x * a.tile;
  const uint32_t cnt = (uint32_t)((a.n - first) < a.tile ? (a.n - first) : a.tile);
  const uint64_t out0 = a.offsets[blockIdx.x];
  uint32_t m = a.counts[blockIdx.x];
  if (m == 0 || out0 >= a.limit) return;
  const uint32_t p0 = threadIdx.x * PPL;
  uint8_t mb[PPL];
  cgptr_t mp = (cgptr_t)((uint64_t)(uintptr_t)a.mask + first);
  if (p0 + PPL <= cnt) {
    static_assert(PPL == 8, "2048-point tiles");
    const uint64_t w = load_un<uint64_t>(mp + p0);
    for (int i = 0; i < 8; ++i) mb[i] = (uint8_t)(w >> (8 * i));
  } else {
#pragma unroll
    for (int i = 0; i < PPL; ++i) mb[i] = p0 + i < cnt ? mp[p0 + i] : (uint8_t)0;
  }
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < PPL; ++i) c += mb[i] != 0;
  uint32_t incl = c;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t r = incl - c;
  for (uint32_t w = 0; w < wave; ++w) r += wave_tot[w];
#pragma unroll
  for (int i = 0; i < PPL; ++i) if (mb[i] != 0) sel[r++] = (uint16_t)(p0 + i);
  if (out0 + m > a.limit) m = (uint32_t)(a.limit - out0);
  m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
  cgptr_t gps = (cgptr_t)as_global(a.attrs[0].src), col = (cgptr_t)as_global(a.attrs[1].src), pos = (cgptr_t)as_global(a.attrs[2].src),
          cls = (cgptr_t)as_global(a.attrs[3].src), inten = (cgptr_t)as_global(a.attrs[4].src);
  const uint32_t nch = (m + a.chunk - 1) / a.chunk;
  const uint32_t mc = ((m + nch - 1) / nch + 15u) & ~15u;
  __syncthreads();
  for (uint32_t j0 = 0; j0 < m; j0 += mc) {
    const uint32_t cm = (m - j0) < mc ? (m - j0) : mc;
    const uint64_t ga = a.dst_aos + (out0 + j0) * STRIDE;
    const uint32_t mis = (uint32_t)(ga & 15u);
    // a chunk is up to three points per lane (24 KiB of 41-byte records, 256 lanes): the pieces of ALL of a lane's points are requested before
    // the first record is assembled (one round trip per chunk instead of one per point)
    constexpr int UG = 3;
    for (uint32_t jb = threadIdx.x; jb < cm; jb += kBlock * UG) {
      uint64_t g[UG], pz[UG];
      uint32_t c0[UG];
      uint16_t c1[UG], in[UG];
      u32x4 pa[UG];
      uint8_t cl[UG];
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t j = jb + (uint32_t)u * kBlock < cm ? jb + (uint32_t)u * kBlock : jb;
        const uint64_t i = first + sel[j0 + j];
        g[u] = load_un<uint64_t>(gps + i * 8);
        c0[u] = load_un<uint32_t>(col + i * 6);
        c1[u] = load_un<uint16_t>(col + i * 6 + 4);
        pa[u] = load_un<u32x4>(pos + i * 24);
        pz[u] = load_un<uint64_t>(pos + i * 24 + 16);
        cl[u] = load_un<uint8_t>(cls + i);
        in[u] = load_un<uint16_t>(inten + i * 2);
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t j = jb + (uint32_t)u * kBlock;
        if (j >= cm) break;
        pstlas::RecordImage<STRIDE> img;
        img.put(0, 8, g[u]);
        img.put(8, 6, (uint64_t)c0[u] | ((uint64_t)c1[u] << 32));
        img.put(14, 8, (uint64_t)pa[u].x | ((uint64_t)pa[u].y << 32));
        img.put(22, 8, (uint64_t)pa[u].z | ((uint64_t)pa[u].w << 32));
        img.put(30, 8, pz[u]);
        img.put(38, 1, cl[u]);
        img.put(39, 2, in[u]);
        img.store(lds + (mis + j * STRIDE));
      }
    }
    __syncthreads();
    tile_store<kBlock>(lds, as_global(ga - mis), mis, cm * STRIDE);
    if (j0 + mc < m) __syncthreads();
  }
}

template <typename SP>
static bool filter_plan_equals(const FilterArgs& a, bool dst_aos) {
  if (a.n_attrs != (uint32_t)SP::n || a.tile != 2048u) return false;
  if (dst_aos && (a.dst_stride != SP::dst_stride || !a.dst_covered)) return false;
  for (int i = 0; i < SP::n; ++i) {
    const StaticFilterAttr s = SP::attr(i);
    if (a.attrs[i].src_stride != s.src_stride || a.attrs[i].unit != s.unit || a.attrs[i].cnt != s.cnt || (dst_aos && a.attrs[i].dst_off != s.dst_off)) return false;
  }
  return true;
}


// ---- plan-specialised streaming compaction (filter_stream.hpp): in-tree for typed LAS-0 points, run-time compiled for every other layout ------
struct StreamSig {
  int n = 0;
  bool dst_columns = true, covered = true;
  uint32_t dst_stride = 0, cap = 0, total = 0;
  uint32_t size[kMaxFilterAttrs] = {}, dst_off[kMaxFilterAttrs] = {}, piece[kMaxFilterAttrs] = {};
};
// points per LDS round: about 52 KiB of values (three 512-lane blocks per CU), a multiple of 16 -- but at least the 1088 matches a tile of a
// half-dense mask has (1024 +- 23), as long as that stays below 72 KiB (two blocks per CU): a second round over a few records costs more than the
// third block buys
__host__ __device__ constexpr uint32_t stream_cap_for(uint32_t total) {
  const uint32_t t = total ? total : 1u;
  uint32_t c = (52u * 1024u / t) / 16u * 16u;
  if (c < 1088u && 1088u * t <= 72u * 1024u) c = 1088u;
  return c > 2048u ? 2048u : c;
}
// Can this launch take a streaming kernel?  Columnar source (every attribute contiguous), at most 64 (columns) / 96 (records) bytes per point (the
// lane holds four points in registers), records of at most 128 bytes.  Values go to LDS in pieces as wide as the target's alignment allows.
static uint32_t aligned_piece(uint32_t size, uint64_t a0, uint64_t a1 = 0, uint64_t a2 = 0) {
  uint32_t p = pstf::piece_of(size);
  while (p > 1u && (a0 % p != 0 || a1 % p != 0 || a2 % p != 0)) p >>= 1;
  return p;
}
static bool stream_sig_from_args(const FilterArgs& a, bool dst_aos, StreamSig* sig) {
  if (a.tile != pstf::kStreamTile || a.n_attrs == 0 || a.n_attrs > (uint32_t)kMaxFilterAttrs) return false;
  sig->n = (int)a.n_attrs;
  sig->dst_columns = !dst_aos;
  uint32_t total = 0;
  for (uint32_t i = 0; i < a.n_attrs; ++i) {
    const uint32_t size = a.attrs[i].unit * a.attrs[i].cnt;
    if (size == 0 || a.attrs[i].src_stride != size) return false;
    sig->size[i] = size;
    sig->dst_off[i] = dst_aos ? a.attrs[i].dst_off : 0u;
    total += size;
  }
  // bytes per point: 64 into columns (typed LAS-9, 75 bytes in 18 spans: 0.36 of peak against the gather kernel's 0.58 -- the span bookkeeping
  // spills), 96 into records (LAS-9 0.43 -> 0.66, LAS-5 0.44 -> 0.67: every LAS point format fits); PST_FILTER_STREAM_MAX_BYTES overrides both
  static const uint32_t max_env = [] { const char* v = std::getenv("PST_FILTER_STREAM_MAX_BYTES"); return v && *v ? (uint32_t)std::atoi(v) : 0u; }();
  if (total > (max_env ? max_env : dst_aos ? 96u : 64u)) return false;
  // records whose every byte is written are assembled as images; records with padding (or with attributes the source lacks) are staged from
  // the target and only the attributes' bytes replaced
  sig->covered = !dst_aos || (a.dst_covered && a.dst_stride == total);
  if (dst_aos && a.dst_stride > 128u) return false;
  // the staged form pays for the span it reads and for per-piece stores: measured (profiles/r04_ab_compaction_padded.txt) +17 % against the
  // gather kernel for ten narrow attributes in a 40-byte record, -3 % for six wide ones in a 64-byte record -- taken below 4 bytes per attribute
  if (dst_aos && !sig->covered && total > 4u * a.n_attrs) return false;
  for (uint32_t i = 0; i < a.n_attrs; ++i)
    sig->piece[i] = !dst_aos ? aligned_piece(sig->size[i], a.attrs[i].dst)
                    : sig->covered ? pstf::piece_of(sig->size[i]) : aligned_piece(sig->size[i], a.dst_aos, a.dst_stride, a.attrs[i].dst_off);
  sig->dst_stride = dst_aos ? a.dst_stride : 0u;
  sig->total = total;
  sig->cap = stream_cap_for(dst_aos ? a.dst_stride : total);
  return true;
}
// the translation unit hipRTC compiles for `sig` (also the cache key)
static std::string stream_source(const StreamSig& s, const pstk::FilterPredicate* pred = nullptr) {
  std::ostringstream o;
  o << "#include \"filter_stream.hpp\"\n";
  if (pred) o << pred->function_text;  // PstV3 and pst_pred(<attributes by name>, i, p0 .. p3), written by expr.cpp
  o << "struct PstFilterPlan {\n";
  o << "  static constexpr int n = " << s.n << ";\n";
  o << "  static constexpr bool dst_columns = " << (s.dst_columns ? "true" : "false") << ", covered = " << (s.covered ? "true" : "false") << ", has_pred = " << (pred ? "true" : "false") << ";\n";
  if (pred) {
    // the lane's four points are in `w`, attribute after attribute (4 x size(k) bytes each): the predicate's arguments are cut out of it at
    // compile-time offsets and the four answers packed like four mask bytes
    o << "  template <int W> __device__ static __forceinline__ uint32_t pred_mask(const uint32_t (&w)[W], const uint64_t i0, const double* const (&p)[4]) {\n";
    o << "    using namespace pstd;\n    uint32_t m = 0;\n";
    o << "    pstq::static_for<0, 4>([&](auto I) __attribute__((always_inline)) {\n      constexpr uint32_t t = (uint32_t) decltype(I)::value;\n";
    std::string call;
    for (size_t q = 0; q < pred->attrs.size(); ++q) {
      const pstk::FilterPredicate::Attr& pa = pred->attrs[q];
      uint32_t before = 0;
      for (int j = 0; j < pa.slot; ++j) before += s.size[j];
      const uint32_t S = s.size[pa.slot], cs = S / pa.ncomp;
      const std::string T = pa.type_name, base = std::to_string(4u * before) + "u + t * " + std::to_string(S) + "u";
      auto comp = [&](uint32_t c) { return "pstq::from_bits<" + T + ">(pstq::img_get<" + base + " + " + std::to_string(c * cs) + "u, " + std::to_string(cs) + "u>(w))"; };
      if (pa.ncomp == 3) o << "      const PstV3<" << T << "> a" << q << " = {" << comp(0) << ", " << comp(1) << ", " << comp(2) << "};\n";
      else o << "      const " << T << " a" << q << " = " << comp(0) << ";\n";
      call += "a" + std::to_string(q) + ", ";
    }
    o << "      m |= (pst_pred(" << call << "i0 + t, p[0], p[1], p[2], p[3]) ? 1u : 0u) << (8u * t);\n    });\n    return m;\n  }\n";
  }
  o << "  static constexpr uint32_t dst_stride = " << s.dst_stride << ", cap = " << s.cap << ";\n";
  o << "  __host__ __device__ static constexpr uint32_t size(int k) {\n    constexpr uint32_t t[n] = {";
  for (int i = 0; i < s.n; ++i) o << (i ? ", " : "") << s.size[i];
  o << "};\n    return t[k];\n  }\n";
  o << "  __host__ __device__ static constexpr uint32_t dst_off(int k) {\n    constexpr uint32_t t[n] = {";
  for (int i = 0; i < s.n; ++i) o << (i ? ", " : "") << s.dst_off[i];
  o << "};\n    return t[k];\n  }\n";
  o << "  __host__ __device__ static constexpr uint32_t piece(int k) {\n    constexpr uint32_t t[n] = {";
  for (int i = 0; i < s.n; ++i) o << (i ? ", " : "") << s.piece[i];
  o << "};\n    return t[k];\n  }\n};\n";
  o << "extern \"C\" __global__ __launch_bounds__(" << pstf::kStreamThreads << ") void pst_jit_filter(const pstf::FilterArgs a) {\n";
  o << "  pstf::filter_stream_body<PstFilterPlan>(a);\n}\n";
  return o.str();
}

// typed LAS points (LasPointFormatN::layout(), las_types.rs: packed in field order; sizes from las_device.hpp): the clouds a pasture user filters
// most often -- formats 0-3 (LAS 1.2) and 6-8 (LAS 1.4) in-tree, the waveform formats through the run-time compiler
template <int FORMAT, bool COLUMNS>
struct LasStreamPlan {
  static constexpr pstlas::Fmt F = pstlas::fmt_of(FORMAT);
  __host__ __device__ static constexpr int slots() {
    int k = 0;
    while (pstlas::typed_slot_offset(F, k) < pstlas::typed_size(F)) ++k;
    return k;
  }
  static constexpr int n = slots();
  static constexpr bool dst_columns = COLUMNS, covered = true, has_pred = false;
  static constexpr uint32_t dst_stride = COLUMNS ? 0u : pstlas::typed_size(F), cap = stream_cap_for(pstlas::typed_size(F));
  __host__ __device__ static constexpr uint32_t size(int k) { return pstlas::typed_slot_offset(F, k + 1) - pstlas::typed_slot_offset(F, k); }
  __host__ __device__ static constexpr uint32_t dst_off(int k) { return COLUMNS ? 0u : pstlas::typed_slot_offset(F, k); }
  __host__ __device__ static constexpr uint32_t piece(int k) { return pstf::piece_of(size(k)); }
};
static_assert(LasStreamPlan<0, false>::n == 10 && LasStreamPlan<0, false>::dst_stride == 35 && LasStreamPlan<3, true>::n == 12 && LasStreamPlan<8, false>::dst_stride == 54, "typed LAS layouts");
// CustomPointTypeBig (test_utils.rs:19-31; buffer_filter_bench.rs:71-74): GpsTime, ColorRGB, Position3D, Classification, Intensity (i16); 41 bytes
template <bool COLUMNS>
struct BigStreamPlan {
  static constexpr int n = 5;
  static constexpr bool dst_columns = COLUMNS, covered = true, has_pred = false;
  static constexpr uint32_t dst_stride = COLUMNS ? 0u : 41u, cap = stream_cap_for(41u);
  __host__ __device__ static constexpr uint32_t size(int k) {
    constexpr uint32_t t[n] = {8, 6, 24, 1, 2};
    return t[k];
  }
  __host__ __device__ static constexpr uint32_t dst_off(int k) {
    constexpr uint32_t t[n] = {0, 8, 14, 38, 39};
    return COLUMNS ? 0u : t[k];
  }
  __host__ __device__ static constexpr uint32_t piece(int k) { return pstf::piece_of(size(k)); }
};
template <typename P>
__global__ __launch_bounds__(pstf::kStreamThreads) void filter_stream_static_kernel(const FilterArgs a) { pstf::filter_stream_body<P>(a); }
// Three workgroups per CU instead of two (round 6): the kernel waits for data 0.45 of its wave cycles (profiles/r05_record_side_pmc.txt) and is held at
// two 512-lane workgroups per CU by its registers (97 for typed LAS-0 into columns; the LDS spans were sized for three all along).  Six waves per SIMD
// mean 80 registers: with two chunks in flight per lane instead of four, typed LAS-0 -> columns and the bench layout -> columns fit WITHOUT scratch --
// same box, 5 alternating pairs: LAS-0 0.970 -> 0.940 ms (+3.3 %, IQRs disjoint), the bench layout 1.065 -> 1.077 (-1.1 %), and every plan that
// spills loses (LAS-0 into records, 8 bytes of scratch: -7 %).  So: the one plan that gains (profiles/r06_experiments.txt E13).
template <typename P> struct StreamOccupancy { static constexpr bool three_blocks = false; };
template <> struct StreamOccupancy<LasStreamPlan<0, true>> { static constexpr bool three_blocks = true; };
template <typename P>
__global__ __launch_bounds__(pstf::kStreamThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void filter_stream_static3_kernel(const FilterArgs a) {
  pstf::filter_stream_body<P, 2>(a);
}

template <typename P>
static bool stream_sig_is(const StreamSig& s) {
  if (s.n != P::n || s.dst_columns != P::dst_columns || s.covered != P::covered || s.dst_stride != P::dst_stride || s.cap != P::cap) return false;
  for (int i = 0; i < P::n; ++i)
    if (s.size[i] != P::size(i) || s.dst_off[i] != P::dst_off(i) || s.piece[i] != P::piece(i)) return false;
  return true;
}
template <typename P>
static void launch_stream_static(unsigned grid, hipStream_t stream, const FilterArgs& a) {
  const uint32_t lds = pstk::lds_with_resident_cap(pstf::stream_lds_bytes<P>(), pstk::kResidentFilterStream);
  static const bool three = [] { const char* v = std::getenv("PST_FILTER_THREE_BLOCKS"); return !(v && *v == '0'); }();  // (0: the A/B switch)
  auto kfn = filter_stream_static_kernel<P>;
  if constexpr (StreamOccupancy<P>::three_blocks) {
    if (three) kfn = filter_stream_static3_kernel<P>;
  }
  if (lds > 64u * 1024u) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(pstf::kStreamThreads), lds, stream, a);
}
// The full tiles of the launch through a streaming kernel, if one is at hand (in-tree, compiled, or PST_JIT=sync); returns the number of tiles it
// covered (0: none -- the caller takes the gather kernel for everything) and the plan family.
static uint32_t launch_stream_tiles(const FilterArgs& a, bool dst_aos, hipStream_t stream, uint32_t* kind, const pstk::FilterPredicate* pred = nullptr,
                                    std::string* error = nullptr) {
  static const bool enabled = [] { const char* v = std::getenv("PST_FILTER_STREAM"); return !(v && *v == '0'); }();
  static const bool in_tree = [] { const char* v = std::getenv("PST_STATIC_PLANS"); return !(v && *v == '0'); }();
  const uint64_t n_full = a.n / pstf::kStreamTile;
  StreamSig sig;
  const pstjit::Mode mode = pstjit::mode();
  if (!enabled || mode == pstjit::Mode::Off || n_full == 0 || n_full > (1ull << 30) || !stream_sig_from_args(a, dst_aos, &sig)) return 0;
  if (pred) {  // the predicate is part of the translation unit: always run-time compiled, in the calling thread (its syntax errors are this call's error)
    const std::string source = stream_source(sig, pred);
    const uint32_t lds = sig.dst_columns ? sig.total * sig.cap + 32u * (uint32_t)sig.n : sig.cap * sig.dst_stride + 64u;
    pstjit::Kernel k;
    if (!pstjit::acquire_source(source, "pst_jit_filter", pstf::kStreamThreads, lds, pstf::kStreamTile, pstjit::Acquire::Wait, &k, error)) return 0;
    FilterArgs b = a;
    for (int q = 0; q < 4; ++q) b.p[q] = pred->p[q];
    void* args[] = {(void*)&b};
    if (hipModuleLaunchKernel(k.fn, (unsigned)n_full, 1, 1, k.blk, 1, 1, pstk::lds_with_resident_cap(k.lds_bytes, pstk::kResidentFilterStream), stream, args, nullptr) != hipSuccess) {
      if (error) *error = std::string("launch failed: ") + hipGetErrorString(hipGetLastError());
      return 0;
    }
    *kind = PST_PLAN_JIT;
    return (uint32_t)n_full;
  }
  if (in_tree && stream_sig_is<BigStreamPlan<true>>(sig)) { launch_stream_static<BigStreamPlan<true>>((unsigned)n_full, stream, a); *kind = PST_PLAN_STATIC; return (uint32_t)n_full; }
  if (in_tree && stream_sig_is<BigStreamPlan<false>>(sig)) { launch_stream_static<BigStreamPlan<false>>((unsigned)n_full, stream, a); *kind = PST_PLAN_STATIC; return (uint32_t)n_full; }
#define PST_TRY_LAS(FMT)                                                                                                                   \
  if (in_tree && stream_sig_is<LasStreamPlan<FMT, true>>(sig)) { launch_stream_static<LasStreamPlan<FMT, true>>((unsigned)n_full, stream, a); *kind = PST_PLAN_STATIC; return (uint32_t)n_full; }   \
  if (in_tree && stream_sig_is<LasStreamPlan<FMT, false>>(sig)) { launch_stream_static<LasStreamPlan<FMT, false>>((unsigned)n_full, stream, a); *kind = PST_PLAN_STATIC; return (uint32_t)n_full; }
  PST_TRY_LAS(0) PST_TRY_LAS(1) PST_TRY_LAS(2) PST_TRY_LAS(3) PST_TRY_LAS(6) PST_TRY_LAS(7) PST_TRY_LAS(8)
#undef PST_TRY_LAS
  const std::string source = stream_source(sig);
  const uint32_t lds = sig.dst_columns ? sig.total * sig.cap + 32u * (uint32_t)sig.n : sig.cap * sig.dst_stride + 64u;  // (= stream_lds_bytes<P>())
  const pstjit::Acquire how = mode == pstjit::Mode::Sync ? pstjit::Acquire::Wait : a.n >= pstjit::min_points() ? pstjit::Acquire::Enqueue : pstjit::Acquire::IfReady;
  pstjit::Kernel k;
  if (!pstjit::acquire_source(source, "pst_jit_filter", pstf::kStreamThreads, lds, pstf::kStreamTile, how, &k)) return 0;  // not ready (or failed): gather
  FilterArgs b = a;
  void* args[] = {(void*)&b};
  if (hipModuleLaunchKernel(k.fn, (unsigned)n_full, 1, 1, k.blk, 1, 1, pstk::lds_with_resident_cap(k.lds_bytes, pstk::kResidentFilterStream), stream, args, nullptr) != hipSuccess) { (void)hipGetLastError(); return 0; }
  *kind = PST_PLAN_JIT;
  return (uint32_t)n_full;
}

}  // namespace

namespace pstk {

size_t filter_workspace_bytes(uint64_t n) {
  const uint64_t max_tiles = (n + 255) / 256 + 1;
  return (size_t)(max_tiles * (sizeof(uint32_t) + sizeof(unsigned long long)) + 64);
}

// Input points per tile (ranks are computed once per tile).
uint32_t filter_tile(bool, uint32_t) { return 2048; }

// Interleaved targets: records per LDS chunk -- about 15 KiB of records (same-box sweep, 41-byte records: 8 KiB 3.99, 12 KiB 4.51,
// 16 KiB 4.60, 21 KiB 4.43, 32 KiB 3.76 TB/s; 15 KiB + the 4 KiB index list leave room for eight blocks per CU, which the kernel's
// register budget -- amdgpu_waves_per_eu(8) -- matches: +3 %), a multiple of 16.  PST_FILTER_TILE_LDS overrides the byte budget (tuning).
static uint32_t filter_chunk(uint32_t dst_stride, long default_budget = 15L * 1024L) {
  static const long forced = [] { const char* v = std::getenv("PST_FILTER_TILE_LDS"); return v && *v ? std::strtol(v, nullptr, 10) : 0L; }();
  const long budget = forced > 0 ? forced : default_budget;
  uint64_t c = (uint64_t)budget / (dst_stride ? dst_stride : 1u);
  c = c / 16 * 16;
  if (c < 16) c = 16;
  if (c > 2048) c = 2048;
  return (uint32_t)c;
}

// Interleaved targets go through an LDS record tile of at least 16 records: does it fit?  (Records beyond ~10 KB -- ByteArray attributes -- do not; the
// caller then compacts into columns and transposes, filter_api.cpp.)
bool filter_record_tile_fits(uint32_t tile, uint32_t dst_stride) {
  const size_t lds_bytes = (((size_t)tile * 2 + 15) & ~(size_t)15) + (size_t)filter_chunk(dst_stride) * dst_stride + 48;
  return lds_bytes <= 160 * 1024 - 256;
}

// Phase 1: counts + offsets (offsets[n_tiles] = number of matches), device memory inside `workspace`.
void launch_filter_count(const uint8_t* mask_dev, uint64_t n, uint32_t tile, uint8_t* workspace, const unsigned long long** out_total_dev,
                         hipStream_t stream, unsigned long long* total_also) {
  const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
  uint32_t* counts = filter_counts(workspace, n, tile);
  const uint32_t tiles_per_block = (kBlock / 64) * kCountTilesPerWave;
  hipLaunchKernelGGL(mask_count_kernel, dim3((n_tiles + tiles_per_block - 1) / tiles_per_block), dim3(kBlock), 0, stream, mask_dev, n, tile, n_tiles, counts);
  launch_filter_scan(n, tile, workspace, out_total_dev, stream, total_also);
}
uint32_t* filter_counts(uint8_t* workspace, uint64_t n, uint32_t tile) {
  const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
  return (uint32_t*)(workspace + ((size_t)n_tiles + 1) * sizeof(unsigned long long));
}
static bool synthetic_stream_sig(const uint32_t* size, int n_attrs, bool dst_aos, uint32_t dst_stride, bool dst_covered, StreamSig* sig) {
  if (n_attrs <= 0 || n_attrs > kMaxFilterAttrs) return false;
  FilterArgs a{};
  a.tile = pstf::kStreamTile;
  a.n_attrs = (uint32_t)n_attrs;
  a.dst_stride = dst_stride;
  a.dst_covered = dst_covered ? 1u : 0u;
  uint32_t off = 0;
  for (int i = 0; i < n_attrs; ++i) {
    const uint32_t sz = size[i];
    a.attrs[i].unit = sz % 16 == 0 ? 16u : sz % 8 == 0 ? 8u : sz % 4 == 0 ? 4u : sz % 2 == 0 ? 2u : 1u;
    a.attrs[i].cnt = sz / a.attrs[i].unit;
    a.attrs[i].src_stride = sz;
    a.attrs[i].dst_off = off;
    off += sz;
  }
  return stream_sig_from_args(a, dst_aos, sig);
}
bool filter_predicate_streams(const uint32_t* size, int n_attrs, bool dst_aos, uint32_t dst_stride, bool dst_covered) {
  static const bool enabled = [] { const char* v = std::getenv("PST_FILTER_STREAM"); return !(v && *v == '0'); }();
  StreamSig sig;
  return enabled && pstjit::mode() != pstjit::Mode::Off && synthetic_stream_sig(size, n_attrs, dst_aos, dst_stride, dst_covered, &sig);
}
std::string filter_stream_source(const uint32_t* size, int n_attrs, bool dst_aos, uint32_t dst_stride, bool dst_covered, const FilterPredicate* pred) {
  StreamSig sig;
  return synthetic_stream_sig(size, n_attrs, dst_aos, dst_stride, dst_covered, &sig) ? stream_source(sig, pred) : std::string();
}
void launch_filter_scan(uint64_t n, uint32_t tile, uint8_t* workspace, const unsigned long long** out_total_dev, hipStream_t stream, unsigned long long* total_also) {
  const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
  unsigned long long* offsets = (unsigned long long*)workspace;
  uint32_t* counts = (uint32_t*)(workspace + ((size_t)n_tiles + 1) * sizeof(unsigned long long));
  static const bool scan_blocks = [] { const char* v = std::getenv("PST_FILTER_SCAN_BLOCKS"); return !(v && *v == '0'); }();  // (0: the one-block scan, the A/B)
  if (scan_blocks && n_tiles > kScanBlockTiles && n_tiles <= (1u << 19))
    hipLaunchKernelGGL(tile_scan_blocks_kernel, dim3((n_tiles + kScanBlockTiles - 1) / kScanBlockTiles), dim3(kScanBlockTiles), 0, stream, (const uint32_t*)counts, n_tiles,
                       offsets, total_also);
  else
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)counts, n_tiles, offsets, total_also);
  *out_total_dev = offsets + n_tiles;
}

// Phase 2: attribute copies.  attrs: all attributes of the layout (launched in groups of kMaxFilterAttrs).
bool launch_filter_scatter(const uint8_t* mask_dev, uint64_t n, uint32_t tile, uint8_t* workspace, uint64_t limit, const uint64_t* src_addr,
                           const uint32_t* src_stride, const uint64_t* dst_addr, const uint32_t* dst_off, const uint32_t* size, int n_attrs,
                           bool dst_aos, uint64_t dst_aos_base, uint32_t dst_stride, bool dst_covered, hipStream_t stream, const FilterPredicate* pred,
                           std::string* error) {
  const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
  if (pred && n_attrs > kMaxFilterAttrs) { if (error) *error = "more attributes than one streaming launch takes"; return false; }
  FilterArgs a{};
  a.mask = mask_dev;
  a.offsets = (const unsigned long long*)workspace;
  a.counts = (const uint32_t*)(workspace + ((size_t)n_tiles + 1) * sizeof(unsigned long long));
  a.n = n;
  a.limit = limit;
  a.dst_aos = dst_aos_base;
  a.dst_stride = dst_stride;
  a.tile = tile;
  a.chunk = dst_aos ? filter_chunk(dst_stride) : 0u;
  const size_t lds_bytes = (((size_t)tile * 2 + 15) & ~(size_t)15) + (dst_aos ? (size_t)a.chunk * dst_stride + 48 : 0);
  if (lds_bytes > 160 * 1024 - 256) return false;  // records too large for the LDS record tile
  for (int g = 0; g < n_attrs; g += kMaxFilterAttrs) {
    const int ng = std::min(kMaxFilterAttrs, n_attrs - g);
    a.n_attrs = (uint32_t)ng;
    // read-modify-write of the record tile is needed unless this launch writes every byte of the records
    a.dst_covered = (dst_covered && n_attrs <= kMaxFilterAttrs) ? 1u : 0u;
    for (int i = 0; i < ng; ++i) {
      FilterAttr& f = a.attrs[i];
      f.src = src_addr[g + i];
      f.dst = dst_aos ? 0 : dst_addr[g + i];
      f.src_stride = src_stride[g + i];
      f.dst_off = dst_off[g + i];
      const uint32_t sz = size[g + i];
      f.unit = sz % 16 == 0 ? 16u : sz % 8 == 0 ? 8u : sz % 4 == 0 ? 4u : sz % 2 == 0 ? 2u : 1u;
      f.cnt = sz / f.unit;
    }
#define PST_FILTER(PPL, AOS)                                                                                                            \
  {                                                                                                                                     \
    if (lds_bytes > 64 * 1024)                                                                                                          \
      (void)hipFuncSetAttribute((const void*)filter_scatter_kernel<PPL, AOS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    hipLaunchKernelGGL((filter_scatter_kernel<PPL, AOS>), dim3(n_tiles), dim3(kBlock), lds_bytes, stream, a);                          \
  }
    if (g == 0) reset_plan_kinds();
    // 1. the plan-specialised streaming kernel over the full tiles (filter_stream.hpp: in-tree for the bench layout and typed LAS-0 points,
    //    run-time compiled for every other layout whose points fit four to a lane), the gather kernel for the ragged last tile.
    //    PST_JIT=0 / pst_jit_set_mode(0) switches every plan-specialised kernel off, the in-tree ones included (bench.py --plan interpreted).
    if (n_attrs <= kMaxFilterAttrs) {
      uint32_t kind = 0;
      const uint32_t covered = launch_stream_tiles(a, dst_aos, stream, &kind, pred, error);
      if (pred && !covered && n / pstf::kStreamTile > 0) return false;  // (the caller falls back to a byte mask)
      if (covered || pred) {
        if (covered) note_plan_kind(kind);
        if ((uint64_t)covered * tile < n) {
          FilterArgs b = a;
          b.tile0 = covered;
          if (dst_aos) {
            if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)filter_scatter_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL((filter_scatter_kernel<8, true>), dim3(1), dim3(kBlock), lds_bytes, stream, b);
          } else {
            hipLaunchKernelGGL((filter_scatter_kernel<8, false>), dim3(1), dim3(kBlock), lds_bytes, stream, b);
          }
        }
        continue;
      }
    }
    // 2. (PST_FILTER_STREAM=0 only: the A/B forms of rounds 2-3) the bench layout into records with compile-time attribute lists on the GATHER
    //    side: point-major record assembly (filter_big_records_kernel: same-box A/B against the granule-major constants 0.657 -> 0.673 of peak,
    //    0.680 with a 24 KiB record tile), PST_FILTER_PM=0 the granule-major constants; the columnar target LOST with constants and has none.
    static const bool static_plans_env = [] { const char* v = std::getenv("PST_STATIC_PLANS"); return !(v && *v == '0'); }();
    const bool static_plans = static_plans_env && pstjit::mode() != pstjit::Mode::Off;
    if (static_plans && dst_aos && n_attrs <= kMaxFilterAttrs && filter_plan_equals<BigFilterPlan>(a, dst_aos)) {
      note_plan_kind(PST_PLAN_STATIC);
      static const int pm = [] { const char* v = std::getenv("PST_FILTER_PM"); return v && *v ? std::atoi(v) : 1; }();
      if (pm) {
        FilterArgs b = a;
        b.chunk = filter_chunk(dst_stride, 24L * 1024L);
        const size_t lds_pm = (((size_t)tile * 2 + 15) & ~(size_t)15) + (size_t)b.chunk * dst_stride + 48;
        hipLaunchKernelGGL((filter_big_records_kernel<8>), dim3(n_tiles), dim3(kBlock), lds_pm, stream, b);
      } else {
        hipLaunchKernelGGL((filter_scatter_static_kernel<8, true, BigFilterPlan>), dim3(n_tiles), dim3(kBlock), lds_bytes, stream, a);
      }
      continue;
    }
    note_plan_kind(PST_PLAN_INTERPRETED);
    note_slow_family("a compaction (filter)", n, pstjit::mode() == pstjit::Mode::Off ? "the run-time compiler is switched off: PST_JIT=0"
                                                 : n_attrs > kMaxFilterAttrs ? "more attributes than one streaming launch takes"
                                                 : pstjit::mode() == pstjit::Mode::Async ? "no streaming kernel for this layout yet (compiling in the background, or points beyond 64 / 96 bytes): the gather kernels"
                                                                                         : "no streaming kernel for this layout (points beyond 64 / 96 bytes, or padded records with wide attributes): the gather kernels");
    switch (tile / kBlock) {
      case 8: if (dst_aos) PST_FILTER(8, true) else PST_FILTER(8, false) break;
      case 4: if (dst_aos) PST_FILTER(4, true) else PST_FILTER(4, false) break;
      case 2: if (dst_aos) PST_FILTER(2, true) else PST_FILTER(2, false) break;
      case 1: if (dst_aos) PST_FILTER(1, true) else PST_FILTER(1, false) break;
      default: return false;
    }
#undef PST_FILTER
  }
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
