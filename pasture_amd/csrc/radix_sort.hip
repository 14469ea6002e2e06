// Stable LSD radix sort of (u32 or u64 key, u32 value) pairs for the spatial-index builders, written for gfx950 (wave64, 160 KB LDS per CU), and the two
// scans they need (exclusive sum, suffix minimum).  Round 6: 64-bit keys (fine voxel grids, Morton keys of the hash-grid kNN) and the scans moved here from
// rocPRIM -- no library code is left on any path of this library.
//
// What it sorts: voxel keys of voxelgrid_filter (27 bits for the bench cloud; pasture-algorithms/src/voxel_grid.rs:109-166 groups points by
// voxel -- the sequential per-voxel centroid sums :124-166 need the points of a voxel in their ORIGINAL order, hence stable) and the row-major
// cell numbers of the kNN grid (normal_estimation.rs:103-120 is a kd-tree there; here ~26 bits).  The library sort (rocPRIM) takes four
// 7-bit passes for such keys at ~2.3 TB/s of moved bytes per pass; this one takes THREE passes of ceil(bits / 3) <= 9 bits (four of <= 8 bits
// for 28 .. 32 key bits: the kNN grid of a surface in a 3-D box has ~2 * 10^9 cells).
//
// One pass = three kernels, no spinning on other workgroups (no decoupled look-back: every kernel boundary is a device-wide barrier):
//   1. radix_hist_kernel     per tile of 8192 keys a 2^d-bin digit histogram (LDS atomics), stored digit-major: counts[digit][tile];
//   2. radix_scan_kernel     one workgroup per digit turns its row into exclusive prefix sums and leaves the row's total; then one
//                            workgroup scans the 2^d totals: dbase[digit] = first output position of the digit;
//   3. radix_scatter_kernel  per tile: keys are loaded WAVE-STRIPED (item i of lane l of wave w = element w * 1024 + i * 64 + l of the tile), so
//                            the order (wave, item, lane) is memory order.  Rank of a key among the tile's keys of the same digit =
//                            (same-digit keys of earlier waves) + (of earlier items of its wave) + (of lower lanes in its item): the last by
//                            MATCHING the digit across the wave -- d ballots, each lane keeps the lanes that agree with it in every bit --
//                            the middle one from a per-wave counter array in LDS that the lowest matching lane advances, the first by a scan
//                            over the waves' counters.  Keys and values are then placed in LDS in digit order and leave in runs: consecutive
//                            lanes write consecutive addresses of a digit's output range (16 pairs per digit and tile on average: 64-byte runs).
// Traffic per pass: 4 (histogram) + 8 + 8 bytes per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_sort.hpp"

namespace pstk {

namespace {

constexpr int kThreads = 512, kWaves = kThreads / 64;
constexpr int kMaxRadix = 512;
// pairs per thread and tile: 16 for 32-bit keys (8192 pairs, 64 KiB of staging), 8 for 64-bit keys (4096 pairs, 48 KiB)
template <typename KeyT> struct SortShape { static constexpr int items = sizeof(KeyT) == 4 ? 16 : 8, tile = kThreads * items; };
constexpr int kTile = SortShape<uint32_t>::tile;  // (32-bit keys: the tile the callers' key kernels count the first histogram over)

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, uint32_t shift, uint32_t mask) { return (uint32_t)(key >> shift) & mask; }
// Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  A tile's run for a digit (64 bytes on average) is
// followed in the output by the NEXT tile's run for that digit: with tile = workgroup id the two halves of a 128-byte line are written
// through different L2s; numbering the tiles so that every XCD owns a contiguous eighth lets neighbouring runs meet in one L2.
__device__ __forceinline__ uint32_t logical_tile() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }

template <typename KeyT>
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(const KeyT* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t mask, uint32_t tiles,
                                                              uint32_t* __restrict__ counts) {
  constexpr int kItems = SortShape<KeyT>::items, kTile = SortShape<KeyT>::tile;
  constexpr int kPerVec = 16 / (int)sizeof(KeyT);  // keys per 16-byte load
  __shared__ uint32_t hist[kMaxRadix];
  const uint32_t tile = logical_tile(), tid = threadIdx.x;
  if (tile >= tiles) return;
  for (uint32_t d = tid; d <= mask; d += kThreads) hist[d] = 0;
  const uint64_t base = (uint64_t)tile * kTile;
  // all of the thread's keys are requested before the first is counted (any order will do for counting: 16 bytes per lane and load; the tile
  // starts at a multiple of 32 KiB of the key array; an array that is not 16-byte aligned takes the element loop)
  typedef KeyT keyvec __attribute__((ext_vector_type(kPerVec)));
  keyvec k[kItems / kPerVec];
  const bool whole = base + kTile <= n && ((uintptr_t)keys & 15u) == 0;
  if (whole) {
#pragma unroll
    for (int i = 0; i < kItems / kPerVec; ++i) k[i] = *reinterpret_cast<const keyvec*>(keys + base + ((uint64_t)i * kThreads + tid) * kPerVec);
  }
  __syncthreads();
  if (whole) {
#pragma unroll
    for (int i = 0; i < kItems / kPerVec; ++i) {
#pragma unroll
      for (int j = 0; j < kPerVec; ++j) atomicAdd(&hist[digit_of<KeyT>(k[i][j], shift, mask)], 1u);
    }
  } else {
    for (int i = 0; i < kItems; ++i) {
      const uint64_t e = base + (uint64_t)i * kThreads + tid;
      if (e < n) atomicAdd(&hist[digit_of<KeyT>(keys[e], shift, mask)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = tid; d <= mask; d += kThreads) counts[(uint64_t)d * tiles + tile] = hist[d];
}

// one workgroup per digit: counts[d][0 .. tiles) -> exclusive prefix sums in place, totals[d] = the row's sum
__global__ __launch_bounds__(kThreads) void radix_scan_rows_kernel(uint32_t* __restrict__ counts, uint32_t tiles, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[kWaves];
  __shared__ uint32_t carry;
  uint32_t* row = counts + (uint64_t)blockIdx.x * tiles;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t t0 = 0; t0 < tiles; t0 += kThreads) {
    const uint32_t t = t0 + tid;
    const uint32_t v = t < tiles ? row[t] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
      if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = carry;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) before += (uint32_t)w < wave ? wsum[w] : 0u;
    if (t < tiles) row[t] = before + inc - v;
    __syncthreads();
    if (tid == kThreads - 1) carry = before + inc;
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry;
}

// one workgroup: dbase[d] = sum of totals[0 .. d)
__global__ __launch_bounds__(kThreads) void radix_scan_totals_kernel(const uint32_t* __restrict__ totals, uint32_t radix, uint32_t* __restrict__ dbase) {
  __shared__ uint32_t wsum[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t v = tid < radix ? totals[tid] : 0u;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) before += (uint32_t)w < wave ? wsum[w] : 0u;
  if (tid < radix) dbase[tid] = before + inc - v;
}

// IOTA: the values of the first pass are the element numbers 0 .. n-1 and are not read (the key kernels of the callers do not write them)
template <typename KeyT, int BITS, bool IOTA>
__global__ __launch_bounds__(kThreads, 4) void radix_scatter_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t n, uint32_t shift,
                                                                    uint32_t tiles, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ dbase,
                                                                    KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  constexpr int kItems = SortShape<KeyT>::items, kTile = SortShape<KeyT>::tile;
  constexpr uint32_t RADIX = 1u << BITS, MASK = RADIX - 1u;
  __shared__ KeyT stage_k[kTile];
  __shared__ uint32_t stage_v[kTile];
  __shared__ uint16_t wcount[kWaves][RADIX];  // per wave and digit: keys seen so far; afterwards: keys of earlier waves
  __shared__ uint32_t dstart[RADIX];          // first LDS position of the digit's keys in this tile
  __shared__ uint32_t gbase[RADIX];           // global position of the digit's first key of this tile, minus dstart
  __shared__ uint32_t wsum[kWaves];
  const uint32_t tile = logical_tile(), tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tile >= tiles) return;
  const uint64_t base = (uint64_t)tile * kTile;
  const uint32_t tile_n = (uint32_t)((n - base) < (uint64_t)kTile ? (n - base) : (uint64_t)kTile);
  for (uint32_t d = lane; d < RADIX; d += 64) wcount[wave][d] = 0;
  KeyT key[kItems];
  uint32_t val[kItems];
  const uint32_t e0 = wave * (kItems * 64u) + lane;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const uint32_t e = e0 + (uint32_t)i * 64u;
    key[i] = e < tile_n ? keys_in[base + e] : (KeyT)~(KeyT)0;
    if constexpr (IOTA) val[i] = (uint32_t)(base + e);
    else val[i] = e < tile_n ? vals_in[base + e] : 0u;
  }
  // ---- rank within the wave, item by item (memory order) -----------------------------------------------------------------------------
  uint16_t rank[kItems];
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64u - lane));  // lanes below this one
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const bool valid = e0 + (uint32_t)i * 64u < tile_n;
    const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);  // lanes whose item exists ...
#pragma unroll
    for (int b = 0; b < BITS; ++b) {                       // ... and whose digit agrees with this lane's in every bit
      const uint64_t m = __builtin_amdgcn_ballot_w64((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t below = (uint32_t)__builtin_popcountll(peers & lt);
    const uint32_t seen = wcount[wave][d];  // (every peer reads the counter before its lowest lane advances it: one wave, program order)
    if (valid && below == 0) wcount[wave][d] = (uint16_t)(seen + (uint32_t)__builtin_popcountll(peers));
    rank[i] = (uint16_t)(seen + below);
  }
  __syncthreads();
  // ---- per digit: keys of earlier waves, the tile's count; then the digit's first position in the tile and in the output ----------
  uint32_t mine = 0;  // the tile's number of keys with digit `tid` (RADIX <= kThreads: one digit per thread)
  if (tid < RADIX) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { const uint32_t c = wcount[w][tid]; wcount[w][tid] = (uint16_t)run; run += c; }
    mine = run;
  }
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (tid < RADIX) {
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) before += (uint32_t)w < wave ? wsum[w] : 0u;
    const uint32_t start = before + inc - mine;
    dstart[tid] = start;
    gbase[tid] = dbase[tid] + counts[(uint64_t)tid * tiles + tile] - start;
  }
  __syncthreads();
  // ---- place pairs in LDS in digit order (stable), then write runs -------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    if (e0 + (uint32_t)i * 64u < tile_n) {
      const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
      const uint32_t pos = dstart[d] + wcount[wave][d] + rank[i];
      stage_k[pos] = key[i];
      stage_v[pos] = val[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const uint32_t p = (uint32_t)i * kThreads + tid;
    if (p < tile_n) {
      const KeyT k = stage_k[p];
      const uint64_t g = (uint64_t)gbase[digit_of<KeyT>(k, shift, MASK)] + p;
      keys_out[g] = k;
      vals_out[g] = stage_v[p];
    }
  }
}

struct Plan { unsigned passes, bits[8]; uint32_t tiles; size_t counts_bytes, total_bytes; };
Plan plan_for(size_t n, unsigned end_bit, int tile = kTile) {
  Plan p{};
  // up to 9 bits one pass, up to 27 three (an odd number: the result lands in the second pair of buffers); 28 .. 32 bits four passes of
  // <= 8 bits and one copy of the result into the second pair (two 4 n-byte copies: 0.15 ms per 10^8 pairs, less than a fifth pass);
  // beyond 32 bits (64-bit keys) as many passes of <= 9 bits as the key needs
  p.passes = end_bit <= 9 ? 1u : (end_bit <= 27 ? 3u : (end_bit <= 32 ? 4u : (end_bit + 8u) / 9u));
  unsigned left = end_bit ? end_bit : 1u;
  for (unsigned i = 0; i < p.passes; ++i) { p.bits[i] = (left + (p.passes - i) - 1) / (p.passes - i); left -= p.bits[i]; }
  p.tiles = (uint32_t)((n + (size_t)tile - 1) / (size_t)tile);
  p.counts_bytes = ((size_t)kMaxRadix * (p.tiles ? p.tiles : 1u) * 4 + 255) & ~(size_t)255;
  p.total_bytes = p.counts_bytes + 2 * kMaxRadix * 4;
  return p;
}

}  // namespace

bool radix_sort_pairs_supported(size_t n, unsigned end_bit) { return end_bit <= 32 && n < 0xFFFFFFF0ull; }

// Where a caller's key kernel leaves the digit histogram of the FIRST pass (it has every key in a register anyway: one kernel and one read of
// the keys less): counts[digit * tiles + tile] = number of keys of tile `tile` (tile_size consecutive keys) whose lowest `bits` bits are `digit`.
RadixFirstPass radix_sort_first_pass(void* tmp, size_t n, unsigned end_bit) {
  const Plan p = plan_for(n, end_bit);
  return RadixFirstPass{(uint32_t*)tmp, p.tiles, p.bits[0], (uint32_t)kTile};
}

// Sorts (keys_a, vals_a) by key bits [0, end_bit); BOTH pairs of buffers are scratch, the result is in (keys_b, vals_b).
// vals_a == nullptr: the values are the element numbers (nothing is read; vals_a is still needed as scratch by sorts of more than one pass).
template <typename KeyT>
static hipError_t radix_sort_pairs_any(void* tmp, size_t& bytes, KeyT* keys_a, KeyT* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned end_bit,
                                       hipStream_t stream, bool iota, bool first_hist_ready) {
  const Plan p = plan_for(n, end_bit, SortShape<KeyT>::tile);
  if (!tmp) { bytes = p.total_bytes; return hipSuccess; }
  if (bytes < p.total_bytes) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  uint32_t* counts = (uint32_t*)tmp;
  uint32_t* totals = (uint32_t*)((uint8_t*)tmp + p.counts_bytes);
  uint32_t* dbase = totals + kMaxRadix;
  KeyT *ki = keys_a, *ko = keys_b;
  uint32_t *vi = vals_a, *vo = vals_b;
  const unsigned grid = (p.tiles + 7u) & ~7u;  // (logical_tile: a multiple of 8 workgroups)
  unsigned shift = 0;
  for (unsigned pass = 0; pass < p.passes; ++pass) {
    const unsigned b = p.bits[pass];
    const uint32_t radix = 1u << b, mask = radix - 1u;
    if (!(pass == 0 && first_hist_ready))
      hipLaunchKernelGGL(radix_hist_kernel<KeyT>, dim3(grid), dim3(kThreads), 0, stream, (const KeyT*)ki, (uint64_t)n, shift, mask, p.tiles, counts);
    hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(radix), dim3(kThreads), 0, stream, counts, p.tiles, totals);
    hipLaunchKernelGGL(radix_scan_totals_kernel, dim3(1), dim3(kThreads), 0, stream, totals, radix, dbase);
#define PST_SCATTER(B)                                                                                                                                           \
  case B:                                                                                                                                                      \
    if (pass == 0 && iota) hipLaunchKernelGGL((radix_scatter_kernel<KeyT, B, true>), dim3(grid), dim3(kThreads), 0, stream, (const KeyT*)ki, (const uint32_t*)vi, (uint64_t)n, shift, p.tiles, counts, dbase, ko, vo); \
    else hipLaunchKernelGGL((radix_scatter_kernel<KeyT, B, false>), dim3(grid), dim3(kThreads), 0, stream, (const KeyT*)ki, (const uint32_t*)vi, (uint64_t)n, shift, p.tiles, counts, dbase, ko, vo);               \
    break;
    switch (b) {
      PST_SCATTER(1) PST_SCATTER(2) PST_SCATTER(3) PST_SCATTER(4) PST_SCATTER(5) PST_SCATTER(6) PST_SCATTER(7) PST_SCATTER(8) PST_SCATTER(9)
      default: return hipErrorInvalidValue;
    }
#undef PST_SCATTER
    shift += b;
    KeyT* t = ki; ki = ko; ko = t;
    uint32_t* u = vi; vi = vo; vo = u;
  }
  if (p.passes % 2 == 0) {  // an even number of passes leaves the result in the first pair (now `ki` / `vi`): copy it over
    hipError_t e = hipMemcpyAsync(keys_b, keys_a, n * sizeof(KeyT), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(vals_b, vals_a, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}
hipError_t radix_sort_pairs_u32(void* tmp, size_t& bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned end_bit,
                                hipStream_t stream, bool iota, bool first_hist_ready) {
  return radix_sort_pairs_any<uint32_t>(tmp, bytes, keys_a, keys_b, vals_a, vals_b, n, end_bit, stream, iota, first_hist_ready);
}
// 64-bit keys (end_bit <= 64): ceil(end_bit / 9) passes over 4096-pair tiles
hipError_t radix_sort_pairs_u64(void* tmp, size_t& bytes, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned end_bit,
                                hipStream_t stream) {
  if (end_bit > 64 || n >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
  return radix_sort_pairs_any<uint64_t>(tmp, bytes, keys_a, keys_b, vals_a, vals_b, n, end_bit, stream, false, false);
}

// ---- the two scans of the index builders ------------------------------------------------------------------------------------------------------
// Reduce-then-scan over blocks of kScanBlock elements: block sums, one workgroup scans the sums (any number of them: it walks them in rounds with
// a carry), every block scans its elements on top of its sum.  OP: 0 = exclusive sum of u32 into u64, front to back; 1 = inclusive minimum of u32,
// BACK TO FRONT, in place (a directory whose run heads were scattered into a 0xFFFFFFFF-filled array).
namespace {
constexpr uint32_t kScanPerThread = 8, kScanBlock = kThreads * kScanPerThread;  // 4096 elements per workgroup
template <int OP> struct ScanAcc;
template <> struct ScanAcc<0> { typedef unsigned long long type; static __device__ __forceinline__ type id() { return 0ull; } static __device__ __forceinline__ type op(type a, type b) { return a + b; } };
template <> struct ScanAcc<1> { typedef uint32_t type; static __device__ __forceinline__ type id() { return 0xFFFFFFFFu; } static __device__ __forceinline__ type op(type a, type b) { return a < b ? a : b; } };
// logical element i of the scan order (OP 1 runs from the back)
template <int OP> __device__ __forceinline__ uint64_t scan_index(uint64_t i, uint64_t n) { return OP == 1 ? n - 1 - i : i; }

template <int OP, typename A = typename ScanAcc<OP>::type>
__device__ __forceinline__ A block_inclusive(A v, A* wsum, A* total) {  // inclusive scan of one value per thread across the block; *total = the block's result
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  A inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    A o;
    if constexpr (sizeof(A) == 8) o = (A)__shfl_up((unsigned long long)inc, off, 64);
    else o = (A)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc = ScanAcc<OP>::op(o, inc);
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  A before = ScanAcc<OP>::id();
#pragma unroll
  for (int w = 0; w < kWaves; ++w) before = (uint32_t)w < wave ? ScanAcc<OP>::op(before, wsum[w]) : before;
  A all = ScanAcc<OP>::id();
#pragma unroll
  for (int w = 0; w < kWaves; ++w) all = ScanAcc<OP>::op(all, wsum[w]);
  *total = all;
  __syncthreads();
  return ScanAcc<OP>::op(before, inc);
}

template <int OP>
__global__ __launch_bounds__(kThreads) void scan_block_sums_kernel(const uint32_t* __restrict__ in, uint64_t n, typename ScanAcc<OP>::type* __restrict__ sums) {
  typedef typename ScanAcc<OP>::type A;
  __shared__ A wsum[kWaves];
  const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
  A acc = ScanAcc<OP>::id();
#pragma unroll
  for (uint32_t k = 0; k < kScanPerThread; ++k) {
    const uint64_t i = base + (uint64_t)threadIdx.x * kScanPerThread + k;
    if (i < n) acc = ScanAcc<OP>::op(acc, (A)in[scan_index<OP>(i, n)]);
  }
  A total;
  (void)block_inclusive<OP>(acc, wsum, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: sums[b] <- combination of sums[0 .. b) (exclusive), in rounds of kThreads with a carry
template <int OP>
__global__ __launch_bounds__(kThreads) void scan_sums_kernel(typename ScanAcc<OP>::type* __restrict__ sums, uint32_t n_blocks) {
  typedef typename ScanAcc<OP>::type A;
  __shared__ A wsum[kWaves];
  A carry = ScanAcc<OP>::id();
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += kThreads) {
    const uint32_t b = b0 + threadIdx.x;
    const A v = b < n_blocks ? sums[b] : ScanAcc<OP>::id();
    A total;
    const A inc = block_inclusive<OP>(v, wsum, &total);
    // exclusive = everything before this element: the inclusive value of the previous thread
    A prev;
    if constexpr (sizeof(A) == 8) prev = (A)__shfl_up((unsigned long long)inc, 1, 64);
    else prev = (A)__shfl_up((int)inc, 1, 64);
    __shared__ A last_of_wave[kWaves];
    if ((threadIdx.x & 63u) == 63u) last_of_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) prev = threadIdx.x == 0 ? ScanAcc<OP>::id() : last_of_wave[(threadIdx.x >> 6) - 1];
    if (b < n_blocks) sums[b] = ScanAcc<OP>::op(carry, prev);
    carry = ScanAcc<OP>::op(carry, total);
    __syncthreads();
  }
}
template <int OP>
__global__ __launch_bounds__(kThreads) void scan_apply_kernel(const uint32_t* __restrict__ in, uint64_t n, const typename ScanAcc<OP>::type* __restrict__ sums,
                                                              typename ScanAcc<OP>::type* __restrict__ out) {
  typedef typename ScanAcc<OP>::type A;
  __shared__ A wsum[kWaves];
  const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * kScanPerThread;
  A v[kScanPerThread];
  A acc = ScanAcc<OP>::id();
#pragma unroll
  for (uint32_t k = 0; k < kScanPerThread; ++k) {
    const uint64_t i = base + k;
    v[k] = i < n ? (A)in[scan_index<OP>(i, n)] : ScanAcc<OP>::id();
    acc = ScanAcc<OP>::op(acc, v[k]);
  }
  A total;
  const A inc = block_inclusive<OP>(acc, wsum, &total);
  // what precedes this thread's first element: the block's offset and the threads before it (inclusive value minus the thread's own part is not
  // defined for a minimum: the previous thread's inclusive value is taken instead)
  A prev;
  if constexpr (sizeof(A) == 8) prev = (A)__shfl_up((unsigned long long)inc, 1, 64);
  else prev = (A)__shfl_up((int)inc, 1, 64);
  __shared__ A last_of_wave[kWaves];
  if ((threadIdx.x & 63u) == 63u) last_of_wave[threadIdx.x >> 6] = inc;
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) prev = threadIdx.x == 0 ? ScanAcc<OP>::id() : last_of_wave[(threadIdx.x >> 6) - 1];
  A run = ScanAcc<OP>::op(sums[blockIdx.x], prev);
#pragma unroll
  for (uint32_t k = 0; k < kScanPerThread; ++k) {
    const uint64_t i = base + k;
    if (OP == 0) { if (i < n) out[i] = run; run = ScanAcc<OP>::op(run, v[k]); }           // exclusive
    else { run = ScanAcc<OP>::op(run, v[k]); if (i < n) out[scan_index<OP>(i, n)] = run; }  // inclusive, written where it was read
  }
}
template <int OP>
hipError_t scan_any(void* tmp, size_t& bytes, const uint32_t* in, typename ScanAcc<OP>::type* out, size_t n, hipStream_t stream) {
  typedef typename ScanAcc<OP>::type A;
  const uint32_t n_blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
  const size_t need = ((size_t)(n_blocks ? n_blocks : 1u) * sizeof(A) + 255) & ~(size_t)255;
  if (!tmp) { bytes = need; return hipSuccess; }
  if (bytes < need) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  A* sums = (A*)tmp;
  hipLaunchKernelGGL(scan_block_sums_kernel<OP>, dim3(n_blocks), dim3(kThreads), 0, stream, in, (uint64_t)n, sums);
  hipLaunchKernelGGL(scan_sums_kernel<OP>, dim3(1), dim3(kThreads), 0, stream, sums, n_blocks);
  hipLaunchKernelGGL(scan_apply_kernel<OP>, dim3(n_blocks), dim3(kThreads), 0, stream, in, (uint64_t)n, (const A*)sums, out);
  return hipGetLastError();
}
}  // namespace

// exclusive prefix sum of u32 counts into u64 offsets (out[0] = 0)
hipError_t exclusive_sum_u32_u64(void* tmp, size_t& bytes, const uint32_t* in, unsigned long long* out, size_t n, hipStream_t stream) {
  return scan_any<0>(tmp, bytes, in, out, n, stream);
}
// in place: data[i] = min(data[i], data[i + 1], ..., data[n - 1])
hipError_t suffix_min_u32(void* tmp, size_t& bytes, uint32_t* data, size_t n, hipStream_t stream) {
  return scan_any<1>(tmp, bytes, data, data, n, stream);
}

}  // namespace pstk
