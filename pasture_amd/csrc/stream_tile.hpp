// K1 / K2 fast path, round 5 body — columnar Vec3f64 stream: copy | affine | AABB in ONE pass over HBM (gfx950).
//
// Same memory mapping as the round 1-4 kernel (stream.hip header): lanes move 16-byte double2 vectors, a tile is KLOADS x 256
// vectors with KLOADS a multiple of 3, so the xyz phase of a lane's j-th vector does not depend on the tile.  What changed
// (review of round 4: 143 lane-instructions per point where the arithmetic needs ~16):
//   * ROTATED accumulators.  256 = 1 (mod 3), so the phase of half hh of lane t's j-th vector is (c0 + 2j + hh) mod 3 with
//     c0 = (vec_first + 2t) mod 3: the lane's doubles come in the order c0, c0+1, c0+2, c0, ...  The kernel keeps THREE running
//     minima / maxima and three scale / offset pairs in that rotated order (index (2j + hh) mod 3 is a compile-time constant) and
//     rotates back once per block: 12 three-way selects in all, none of them in the loop, no 64-bit modulo.
//   * v_min_f64 / v_max_f64 issued as such.  In IEEE mode (the default for compute kernels) they return the non-NaN operand, which is
//     the reference's strict `<` / `>` (bounds.rs:34-51); `__builtin_fmin` would first canonicalise every operand it cannot prove
//     quiet (one extra v_max_f64 x, x each).
//   * the block reduction goes through LDS once (six values per lane, maxima negated so that every fold is a minimum) and is folded
//     by 48 lanes of one wave instead of 6 x 6 shuffle-and-fold steps in every wave.
//   * any number of loads per lane and any block size: the phase of a tile is (2 * tile * tile vectors) mod 3, taken from the tile index in
//     32-bit scalar arithmetic, so tiles need not be whole xyz periods.
//   * FEW BYTES IN FLIGHT.  The fused read + write stream peaks at about 48 KiB of loads in flight per CU and loses 8-12 % at the 192 KiB
//     the round 1-4 launch kept there (profiles/r05_stream_sweeps.txt: memory-side queues, not the CUs, are what saturates); the launch caps the
//     blocks resident per CU through its dynamic LDS size, and the block fold's rows live in that allocation.
//   * blocks are numbered so that each XCD streams one contiguous eighth of the range (as before).  What round 4 read as a dependence on the
//     DISTANCE between the XCD streams is a dependence on the span of addresses one launch covers (the same 10^8 points' worth of work spread over
//     5 GiB instead of 2.3 GiB runs 6-11 % faster, whatever the rotation of the streams inside their regions and however many streams there are):
//     nothing a kernel can choose for a dense column, and with the shallow queues above most of it is gone (6.66 TB/s at 10^8, 6.70-6.76 at
//     2-8 10^8, 6.93 at 10^9 points, fold launches included).
#pragma once
#include "device_common.hpp"

namespace pstd {

typedef double f64x2_t __attribute__((ext_vector_type(2)));

struct Stream2Params {
  const double* src;   // first double of the source range (x of point s0)
  double* dst;         // first double of the target range (may equal src for in-place)
  uint64_t n_doubles;  // 3 * points
  uint64_t n_vec;      // 16-byte vectors in the body
  double scale[3];
  double offset[3];
  double* partials;     // [gridDim.x][6] = {min xyz, max xyz}
  uint64_t xcd_stride;  // tiles between the starts of two neighbouring XCDs' regions = tiles per XCD (gridDim.x = 8 * that); the tuning harness sweeps it
  uint32_t vec_first;   // first double covered by the vector body (0 or 1)
  uint32_t plain;       // 1: tile = blockIdx.x (no XCD-aware numbering)
};

__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// A wave-uniform pointer pinned to scalar registers: accesses `p[lane offset]` then take the scalar-base form of global_load / global_store
// (base in SGPRs + 32-bit lane offset) instead of a 64-bit vector address per access.
template <typename P>
__device__ __forceinline__ P* uniform_ptr(P* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (P*)(((uint64_t)hi << 32) | lo);
}

// Block-wide fold of three minima and three maxima (BLK threads); the result is valid in lanes 0, 8, 16 (minima x y z) and
// 24, 32, 40 (maxima) of wave 0, as `out`; `lds` holds 6 * (BLK + 8) doubles.  No NaN can sit in an accumulator (seeds are finite, a quiet
// NaN never wins a fold and signalling ones are quieted before they are folded), so max(a, b) = -min(-a, -b) exactly.
template <int BLK>
__device__ __forceinline__ bool block_reduce_minmax3_lds(const double (&mn)[3], const double (&mx)[3], double* lds, double& out, uint32_t& which) {
  constexpr int kStride = BLK + 8;  // rows shifted by 16 banks against each other
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    lds[i * kStride + t] = mn[i];
    lds[(3 + i) * kStride + t] = -mx[i];
  }
  __syncthreads();
  if (t >= 48) return false;
  const uint32_t i = t >> 3, seg = t & 7u;
  const double* row = lds + i * kStride + seg;
  double a = row[0], b = row[8];
#pragma unroll
  for (int r = 2; r < BLK / 8; r += 2) {
    a = vmin_f64(a, row[8 * r]);
    b = vmin_f64(b, row[8 * r + 8]);
  }
  a = vmin_f64(a, b);
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) a = vmin_f64(a, shfl_xor_any(a, off));
  out = i < 3 ? a : -a;
  which = i;
  return seg == 0;
}

// KLOADS = 16-byte loads in flight per lane (any count >= 1); one tile of KLOADS x BLK vectors per block.
template <bool AFFINE, bool WRITE, bool BOUNDS, int KLOADS, int BLK = kBlock>
__global__ __launch_bounds__(BLK) void vec3f64_stream2_kernel(const Stream2Params p) {
  constexpr uint32_t kTileVec = KLOADS * BLK;
  constexpr uint32_t kBm = BLK % 3;  // a row of BLK vectors advances the xyz phase by 2 * kBm
  const uint32_t t = threadIdx.x;
  const PST_AS_GLOBAL f64x2_t* __restrict__ src = (const PST_AS_GLOBAL f64x2_t*)(p.src + p.vec_first);
  PST_AS_GLOBAL f64x2_t* __restrict__ dst = (PST_AS_GLOBAL f64x2_t*)(p.dst + p.vec_first);

  // block -> tile: consecutive workgroup ids land on consecutive XCDs; XCD x streams the x-th contiguous eighth of the range
  uint64_t tile = blockIdx.x;
  if (!p.plain) tile = (uint64_t)(blockIdx.x & 7u) * p.xcd_stride + (blockIdx.x >> 3);
  const uint64_t tile_first = tile * kTileVec;
  const bool live = tile_first < p.n_vec;  // surplus blocks of the rounded-up grid still write their (identity) record

  // xyz phase of the lane's first double: (vec_first + 2 * (tile_first + t)) mod 3, in 32-bit arithmetic
  const uint32_t tile_phase = (uint32_t)(tile % 3u) * ((2u * kTileVec) % 3u);
  const uint32_t c0 = (p.vec_first + 2u * t + tile_phase) % 3u;
  double sr[3] = {1.0, 1.0, 1.0}, orr[3] = {0.0, 0.0, 0.0};
  if constexpr (AFFINE) {
    sr[0] = pick3(c0, p.scale[0], p.scale[1], p.scale[2]);
    sr[1] = pick3(c0, p.scale[1], p.scale[2], p.scale[0]);
    sr[2] = pick3(c0, p.scale[2], p.scale[0], p.scale[1]);
    orr[0] = pick3(c0, p.offset[0], p.offset[1], p.offset[2]);
    orr[1] = pick3(c0, p.offset[1], p.offset[2], p.offset[0]);
    orr[2] = pick3(c0, p.offset[2], p.offset[0], p.offset[1]);
  }
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};

  auto body = [&](f64x2_t v, int j) __attribute__((always_inline)) -> f64x2_t {
    const int k0 = (2 * j * (int)kBm) % 3, k1 = (2 * j * (int)kBm + 1) % 3;
    double a = v.x, b = v.y;
    if constexpr (AFFINE) {
#pragma clang fp contract(off)
      a = a * sr[k0];
      a = a + orr[k0];
      b = b * sr[k1];
      b = b + orr[k1];
    }
    if constexpr (BOUNDS) {
      // Raw v_min_f64 / v_max_f64 return the non-NaN operand only for QUIET NaNs: in IEEE mode a signalling NaN comes back quieted and would
      // poison the accumulator (the next fold then replaces it: the lane's earlier minimum is lost).  Products and sums are always quiet, so the
      // affine modes need nothing; the plain copy folds the LOADED bits, which are canonicalised first (one v_max_f64 v, v per value on a kernel
      // with vector slots to spare) -- the reference's strict `<` / `>` ignore every NaN (bounds.rs:34-51).  The stored value keeps its bits.
      double fa = a, fb = b;
      if constexpr (!AFFINE) { fa = __builtin_canonicalize(a); fb = __builtin_canonicalize(b); }
      mn[k0] = vmin_f64(mn[k0], fa);
      mx[k0] = vmax_f64(mx[k0], fa);
      mn[k1] = vmin_f64(mn[k1], fb);
      mx[k1] = vmax_f64(mx[k1], fb);
    }
    f64x2_t r;
    r.x = a;
    r.y = b;
    return r;
  };

  if (live) {
    // wave-uniform tile base (SGPRs) + 32-bit lane offset: the loads and stores take the scalar-base addressing form, no 64-bit
    // vector address arithmetic per access
    const PST_AS_GLOBAL f64x2_t* __restrict__ ts = src + tile_first;
    PST_AS_GLOBAL f64x2_t* __restrict__ td = dst + tile_first;
    if (tile_first + kTileVec <= p.n_vec) {
      f64x2_t v[KLOADS];
#pragma unroll
      for (int j = 0; j < KLOADS; ++j) v[j] = __builtin_nontemporal_load(&uniform_ptr(ts + j * BLK)[t]);
#pragma unroll
      for (int j = 0; j < KLOADS; ++j) {
        const f64x2_t o = body(v[j], j);
        if constexpr (WRITE) __builtin_nontemporal_store(o, &uniform_ptr(td + j * BLK)[t]);
      }
    } else {
      const uint32_t left = (uint32_t)(p.n_vec - tile_first);  // < kTileVec
#pragma unroll
      for (int j = 0; j < KLOADS; ++j) {
        const uint32_t i = t + (uint32_t)j * BLK;
        if (i < left) {
          const f64x2_t o = body(ts[i], j);
          if constexpr (WRITE) td[i] = o;
        }
      }
    }
  }

  // rotate the accumulators back to x / y / z: component c sits at rotated index (c - c0) mod 3
  double bmn[3] = {kF64Max, kF64Max, kF64Max}, bmx[3] = {-kF64Max, -kF64Max, -kF64Max};
  if constexpr (BOUNDS) {
    bmn[0] = pick3(c0, mn[0], mn[2], mn[1]);
    bmn[1] = pick3(c0, mn[1], mn[0], mn[2]);
    bmn[2] = pick3(c0, mn[2], mn[1], mn[0]);
    bmx[0] = pick3(c0, mx[0], mx[2], mx[1]);
    bmx[1] = pick3(c0, mx[1], mx[0], mx[2]);
    bmx[2] = pick3(c0, mx[2], mx[1], mx[0]);
  }

  // ragged doubles outside the 16-byte aligned vector body (at most one in front, two behind): block 0, lanes 0..2
  if (blockIdx.x == 0 && t < 3) {
    const uint64_t tail_first = p.vec_first + 2 * p.n_vec;
    uint64_t idx = ~0ull;
    if (t == 0 && p.vec_first == 1) idx = 0;
    if (t >= 1 && tail_first + (t - 1) < p.n_doubles) idx = tail_first + (t - 1);
    if (idx != ~0ull) {
      const uint32_t c = (uint32_t)(idx % 3ull);
      double a = p.src[idx];
      if constexpr (AFFINE) {
#pragma clang fp contract(off)
        a = a * pick3(c, p.scale[0], p.scale[1], p.scale[2]);
        a = a + pick3(c, p.offset[0], p.offset[1], p.offset[2]);
      }
      if constexpr (WRITE) p.dst[idx] = a;
      if constexpr (BOUNDS) {
        const double fa = AFFINE ? a : __builtin_canonicalize(a);  // (see body: a signalling NaN must not reach the raw fold)
#pragma unroll
        for (uint32_t cc = 0; cc < 3; ++cc) {
          bmn[cc] = vmin_f64(bmn[cc], cc == c ? fa : kF64Max);
          bmx[cc] = vmax_f64(bmx[cc], cc == c ? fa : -kF64Max);
        }
      }
    }
  }

  if constexpr (BOUNDS) {
    extern __shared__ double red[];  // 6 * (BLK + 8) doubles, plus whatever the launch adds to cap the blocks resident per CU
    double out;
    uint32_t which;
    if (block_reduce_minmax3_lds<BLK>(bmn, bmx, red, out, which)) p.partials[(uint64_t)blockIdx.x * 6 + which] = out;
  }
}

}  // namespace pstd
