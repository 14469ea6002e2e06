// Tuning harness #2 (not part of the product): structural variants of the Vec3f64 stream kernel.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace pstk { int device_cus() { return 256; } }
#include "../pasture_amd/csrc/stream.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// Variant P: persistent, software-pipelined (next tile's loads issued before the current tile is consumed)
template <int K, int BLK>
__global__ __launch_bounds__(BLK) void stream_pipelined(const StreamParams p) {
  constexpr int kTileVec = K * BLK;
  const uint32_t t = threadIdx.x;
  const PST_AS_GLOBAL f64x2* __restrict__ src = (const PST_AS_GLOBAL f64x2*)(p.src + p.vec_first);
  PST_AS_GLOBAL f64x2* __restrict__ dst = (PST_AS_GLOBAL f64x2*)(p.dst + p.vec_first);
  double sc[3][2], of[3][2]; uint32_t comp[3][2];
  for (int j = 0; j < 3; ++j) for (int hh = 0; hh < 2; ++hh) {
    const uint32_t c = (uint32_t)((p.vec_first + 2ull * (uint64_t)(j * BLK + t) + hh) % 3ull);
    comp[j][hh] = c; sc[j][hh] = pick3(c, p.scale[0], p.scale[1], p.scale[2]); of[j][hh] = pick3(c, p.offset[0], p.offset[1], p.offset[2]);
  }
  double mn[3][2], mx[3][2];
  for (int j = 0; j < 3; ++j) { mn[j][0] = mn[j][1] = kF64Max; mx[j][0] = mx[j][1] = -kF64Max; }
  const uint64_t n_tiles = p.n_vec / kTileVec;  // harness: full tiles only
  uint64_t tile = blockIdx.x;
  f64x2 cur[K], nxt[K];
  if (tile < n_tiles) {
#pragma unroll
    for (int j = 0; j < K; ++j) cur[j] = __builtin_nontemporal_load(&src[tile * kTileVec + t + (uint64_t)j * BLK]);
  }
  while (tile < n_tiles) {
    const uint64_t next = tile + gridDim.x;
    if (next < n_tiles) {
#pragma unroll
      for (int j = 0; j < K; ++j) nxt[j] = __builtin_nontemporal_load(&src[next * kTileVec + t + (uint64_t)j * BLK]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int jj = j % 3;
      double a = cur[j].x, b = cur[j].y;
      a = a * sc[jj][0]; a = a + of[jj][0]; b = b * sc[jj][1]; b = b + of[jj][1];
      mn[jj][0] = __builtin_fmin(mn[jj][0], a); mx[jj][0] = __builtin_fmax(mx[jj][0], a);
      mn[jj][1] = __builtin_fmin(mn[jj][1], b); mx[jj][1] = __builtin_fmax(mx[jj][1], b);
      f64x2 r; r.x = a; r.y = b;
      __builtin_nontemporal_store(r, &dst[tile * kTileVec + t + (uint64_t)j * BLK]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) cur[j] = nxt[j];
    tile = next;
  }
  double bmn[3] = {kF64Max, kF64Max, kF64Max}, bmx[3] = {-kF64Max, -kF64Max, -kF64Max};
  for (int j = 0; j < 3; ++j) for (int hh = 0; hh < 2; ++hh) for (uint32_t c = 0; c < 3; ++c) {
    const bool hit = comp[j][hh] == c;
    bmn[c] = __builtin_fmin(bmn[c], hit ? mn[j][hh] : kF64Max); bmx[c] = __builtin_fmax(bmx[c], hit ? mx[j][hh] : -kF64Max);
  }
  // cheap epilogue for the harness: lane 0 of each wave writes its partial after a wave reduce
  for (int off = 32; off >= 1; off >>= 1) for (int i = 0; i < 3; ++i) {
    bmn[i] = __builtin_fmin(bmn[i], shfl_xor_any(bmn[i], off)); bmx[i] = __builtin_fmax(bmx[i], shfl_xor_any(bmx[i], off));
  }
  if ((t & 63) == 0) { double* out = p.partials + ((uint64_t)blockIdx.x * (BLK / 64) + (t >> 6)) * 6; for (int i = 0; i < 3; ++i) { out[i] = bmn[i]; out[3 + i] = bmx[i]; } }
}

// Variant N: non-persistent, one tile per block, block size BLK, wave-level partials only (no LDS reduce / barrier)
template <int K, int BLK, bool BOUNDS>
__global__ __launch_bounds__(BLK) void stream_onetile(const StreamParams p) {
  constexpr int kTileVec = K * BLK;
  const uint32_t t = threadIdx.x;
  const PST_AS_GLOBAL f64x2* __restrict__ src = (const PST_AS_GLOBAL f64x2*)(p.src + p.vec_first);
  PST_AS_GLOBAL f64x2* __restrict__ dst = (PST_AS_GLOBAL f64x2*)(p.dst + p.vec_first);
  const uint64_t base = (uint64_t)blockIdx.x * kTileVec + t;
  f64x2 v[K];
#pragma unroll
  for (int j = 0; j < K; ++j) v[j] = __builtin_nontemporal_load(&src[base + (uint64_t)j * BLK]);
  double mn[3][2], mx[3][2];
  for (int j = 0; j < 3; ++j) { mn[j][0] = mn[j][1] = kF64Max; mx[j][0] = mx[j][1] = -kF64Max; }
  uint32_t comp[3][2];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int jj = j % 3;
    const uint32_t c0 = (uint32_t)((p.vec_first + 2ull * (uint64_t)(jj * BLK + t)) % 3ull);
    const uint32_t c1 = (c0 + 1) % 3;
    comp[jj][0] = c0; comp[jj][1] = c1;
    double a = v[j].x, b = v[j].y;
    a = a * pick3(c0, p.scale[0], p.scale[1], p.scale[2]); a = a + pick3(c0, p.offset[0], p.offset[1], p.offset[2]);
    b = b * pick3(c1, p.scale[0], p.scale[1], p.scale[2]); b = b + pick3(c1, p.offset[0], p.offset[1], p.offset[2]);
    if constexpr (BOUNDS) {
      mn[jj][0] = __builtin_fmin(mn[jj][0], a); mx[jj][0] = __builtin_fmax(mx[jj][0], a);
      mn[jj][1] = __builtin_fmin(mn[jj][1], b); mx[jj][1] = __builtin_fmax(mx[jj][1], b);
    }
    f64x2 r; r.x = a; r.y = b;
    __builtin_nontemporal_store(r, &dst[base + (uint64_t)j * BLK]);
  }
  if constexpr (BOUNDS) {
    double bmn[3] = {kF64Max, kF64Max, kF64Max}, bmx[3] = {-kF64Max, -kF64Max, -kF64Max};
    for (int j = 0; j < 3; ++j) for (int hh = 0; hh < 2; ++hh) for (uint32_t c = 0; c < 3; ++c) {
      const bool hit = comp[j][hh] == c;
      bmn[c] = __builtin_fmin(bmn[c], hit ? mn[j][hh] : kF64Max); bmx[c] = __builtin_fmax(bmx[c], hit ? mx[j][hh] : -kF64Max);
    }
    for (int off = 32; off >= 1; off >>= 1) for (int i = 0; i < 3; ++i) {
      bmn[i] = __builtin_fmin(bmn[i], shfl_xor_any(bmn[i], off)); bmx[i] = __builtin_fmax(bmx[i], shfl_xor_any(bmx[i], off));
    }
    if ((t & 63) == 0) { double* out = p.partials + ((uint64_t)blockIdx.x * (BLK / 64) + (t >> 6)) * 6; for (int i = 0; i < 3; ++i) { out[i] = bmn[i]; out[3 + i] = bmx[i]; } }
  }
}

template <typename F> float timeit(F&& launch, int iters, hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0, s)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

__global__ void fill_random(double* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) * 1000.0;
  }
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
  double *src, *dst, *partials;
  CK(hipMalloc(&src, n * 24)); CK(hipMalloc(&dst, n * 24)); CK(hipMalloc(&partials, 256 << 20));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, 3 * n);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  StreamParams p{}; p.src = src; p.dst = dst; p.n_doubles = 3 * n; p.vec_first = 0; p.n_vec = p.n_doubles / 2;
  for (int c = 0; c < 3; ++c) { p.scale[c] = 0.001; p.offset[c] = 5000.0 * (c + 1); }
  p.partials = partials;
  const int iters = 20;
  auto rep = [&](const char* name, unsigned grid, float ms, double bpp) { printf("%-34s grid=%7u %8.4f ms %8.1f GB/s\n", name, grid, ms, bpp * n / ms / 1e6); fflush(stdout); };
#define BASE(K, G) { unsigned tiles = (unsigned)(p.n_vec / ((K) * 256)); unsigned g = (G) ? std::min<unsigned>(G, tiles) : tiles; \
    rep("product K=" #K, g, timeit([&] { hipLaunchKernelGGL((vec3f64_stream_kernel<true, true, true, K, true, true>), dim3(g), dim3(256), 0, s, p); }, iters, s), 48); }
  BASE(6, 2048) BASE(6, 0) BASE(6, 8192)
#define PIPE(K, BLK, G) rep("pipelined K=" #K " blk=" #BLK, G, timeit([&] { hipLaunchKernelGGL((stream_pipelined<K, BLK>), dim3(G), dim3(BLK), 0, s, p); }, iters, s), 48);
  PIPE(3, 256, 1024) PIPE(3, 256, 2048) PIPE(6, 256, 1024) PIPE(6, 256, 1280) PIPE(6, 256, 2048) PIPE(3, 512, 512) PIPE(3, 512, 1024) PIPE(6, 512, 512)
#define ONE(K, BLK, B) { unsigned g = (unsigned)(p.n_vec / ((K) * (BLK))); rep("onetile K=" #K " blk=" #BLK " bounds=" #B, g, timeit([&] { hipLaunchKernelGGL((stream_onetile<K, BLK, B>), dim3(g), dim3(BLK), 0, s, p); }, iters, s), 48); }
  ONE(3, 256, true) ONE(6, 256, true) ONE(6, 256, false) ONE(9, 256, true) ONE(3, 512, true) ONE(6, 512, true) ONE(3, 1024, true) ONE(6, 1024, true) ONE(6, 128, true) ONE(12, 128, true) ONE(6, 64, true) ONE(12, 64, true)
  // plain device-to-device copy for reference
  rep("hipMemcpyDtoD", 0, timeit([&] { CK(hipMemcpyAsync(dst, src, n * 24, hipMemcpyDeviceToDevice, s)); }, iters, s), 48);
  return 0;
}
