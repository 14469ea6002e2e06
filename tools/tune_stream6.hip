// Tuning harness of the round-5 Vec3f64 stream body (pasture_amd/csrc/stream_tile.hpp): fused convert + affine + AABB (48 B/point), AABB only (24 B/point).
//   shapes  : loads per lane K x threads per block x blocks resident per CU (capped through dynamic LDS), at 10^8 and 10^9 points
//   spacing : the same amount of work (10^8 points' worth) with the eight XCD regions spread further and further apart
//   sizes   : the chosen shape over point counts
// (An earlier revision, tune_stream5, also swept the rotation of the XCD streams inside their regions, the number of streams (8 / 4 / 2 / 1) and
//  the source -> target distance: none of them moves the rate; profiles/r05_stream_sweeps.txt keeps those tables.)
// Build here, run on the GPU box:  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/tune_stream6.hip -o tools/bin/ts6
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace pstk { int device_cus() { return 256; } }
#include "../pasture_amd/csrc/stream.hip"
#include "../pasture_amd/csrc/stream_tile.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static hipStream_t g_s;
template <typename F> float timeit(F&& launch, int iters) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  for (int i = 0; i < 2; ++i) launch();
  CK(hipEventRecord(e0, g_s)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1, g_s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
__global__ void fill_random(double* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) * 1000.0;
  }
}
__global__ void checksum_kernel(const uint64_t* p, uint64_t n, unsigned long long* out) {
  unsigned long long a = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a += p[i] * (2 * i + 1);
  atomicAdd(out, a);
}

using pstd::Stream2Params;
static int g_mode = 7;  // 7 affine + write + bounds, 3 affine + write, 4 bounds only
static const unsigned kLdsCu = 160 * 1024;
// dynamic LDS of one block: the reduction rows, padded so that at most `blocks` blocks fit a CU
static unsigned lds_for(int blk, unsigned blocks) {
  const unsigned need = 6 * (blk + 8) * 8;
  return blocks >= 8 ? need : std::max(need, kLdsCu / (blocks + 1) + 64);
}
template <int K, int BLK> void launch_kb(const Stream2Params& p, unsigned grid, unsigned lds) {
  static bool attr = false;
  if (!attr) {
    attr = true;
    CK(hipFuncSetAttribute((const void*)pstd::vec3f64_stream2_kernel<true, true, true, K, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void*)pstd::vec3f64_stream2_kernel<true, true, false, K, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void*)pstd::vec3f64_stream2_kernel<false, false, true, K, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  }
  if (g_mode == 7) hipLaunchKernelGGL((pstd::vec3f64_stream2_kernel<true, true, true, K, BLK>), dim3(grid), dim3(BLK), lds, g_s, p);
  else if (g_mode == 3) hipLaunchKernelGGL((pstd::vec3f64_stream2_kernel<true, true, false, K, BLK>), dim3(grid), dim3(BLK), lds, g_s, p);
  else hipLaunchKernelGGL((pstd::vec3f64_stream2_kernel<false, false, true, K, BLK>), dim3(grid), dim3(BLK), lds, g_s, p);
}
static void launch_dyn(int K, int BLK, const Stream2Params& p, unsigned grid, unsigned lds) {
#define C(k, b) if (K == k && BLK == b) return launch_kb<k, b>(p, grid, lds);
  C(1, 256) C(2, 256) C(3, 256) C(4, 256) C(6, 256) C(12, 256)
  C(2, 128) C(3, 128) C(6, 128) C(12, 128)
  C(1, 512) C(2, 512) C(3, 512)
  C(1, 1024) C(2, 1024) C(3, 1024) C(2, 768) C(3, 768) C(4, 384) C(3, 384) C(6, 384) C(6, 512) C(4, 512)
#undef C
  printf("no instance K=%d BLK=%d\n", K, BLK); exit(1);
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "shapes";
  const uint64_t n = 100000000ull;
  const uint64_t cap_pts = 1300000000ull;
  double *buf, *partials, *out6;
  CK(hipMalloc(&buf, 2 * cap_pts * 24 + (1ull << 30)));
  CK(hipMalloc(&partials, 1ull << 30)); CK(hipMalloc(&out6, 64));
  double* src = buf;
  double* dst0 = buf + 3 * cap_pts + (64ull << 20) / 8;
  hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, src, 3 * cap_pts);
  CK(hipDeviceSynchronize());
  CK(hipStreamCreate(&g_s));
  const double scale[3] = {0.001, 0.001, 0.001}, offset[3] = {500000.0, 5400000.0, 100.0};
  const int iters = 6;
  unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
  auto checksum = [&](const double* p, uint64_t nd) { unsigned long long h = 0; CK(hipMemsetAsync(d_sum, 0, 8, g_s));
    hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, g_s, (const uint64_t*)p, nd, d_sum); CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, g_s)); CK(hipStreamSynchronize(g_s)); return h; };
  auto mk = [&](const double* s0, uint64_t pts, int K, int BLK, double* dst, unsigned* grid) {
    Stream2Params p{}; p.src = s0; p.dst = dst; p.n_doubles = 3 * pts; p.vec_first = ((uintptr_t)s0 & 15u) ? 1 : 0; p.n_vec = (p.n_doubles - p.vec_first) / 2;
    for (int c = 0; c < 3; ++c) { p.scale[c] = scale[c]; p.offset[c] = offset[c]; }
    p.partials = partials;
    const uint64_t tile_vec = (uint64_t)K * BLK, n_tiles = std::max<uint64_t>(1, (p.n_vec + tile_vec - 1) / tile_vec);
    p.xcd_stride = (n_tiles + 7) / 8; p.plain = 0;
    *grid = (unsigned)(8 * p.xcd_stride);
    return p;
  };
  auto bpp = [&] { return g_mode == 4 ? 24.0 : 48.0; };

  // ---- correctness against the round-4 kernel: bounds and a checksum of the written column, aligned and misaligned, ragged sizes
  {
    int bad = 0;
    for (uint64_t pts : {100000000ull, 99999989ull, 1000003ull, 1023ull, 5ull, 1ull}) for (int mis = 0; mis < 2; ++mis) {
      const double* s0 = src + mis;
      double* d0 = dst0 + mis;
      double ref6[6], new6[6];
      CK(hipMemsetAsync(dst0, 0, (3 * pts + 4) * 8, g_s));
      pstk::launch_vec3f64_stream(s0, d0, pts, scale, offset, 7u, partials, out6, g_s);
      CK(hipMemcpyAsync(ref6, out6, 48, hipMemcpyDeviceToHost, g_s)); CK(hipStreamSynchronize(g_s));
      const unsigned long long href = checksum(dst0, 3 * pts + 4);
      const int Ks[] = {1, 2, 3, 4, 6, 3, 2, 2}, Bs[] = {256, 256, 256, 256, 256, 128, 512, 1024};
      for (int v = 0; v < 8; ++v) {
        CK(hipMemsetAsync(dst0, 0, (3 * pts + 4) * 8, g_s));
        unsigned grid; Stream2Params p = mk(s0, pts, Ks[v], Bs[v], d0, &grid);
        launch_dyn(Ks[v], Bs[v], p, grid, lds_for(Bs[v], 8));
        pstk::launch_finalize_bounds(partials, grid, out6, g_s);
        CK(hipMemcpyAsync(new6, out6, 48, hipMemcpyDeviceToHost, g_s)); CK(hipStreamSynchronize(g_s));
        const unsigned long long h = checksum(dst0, 3 * pts + 4);
        const bool ok = memcmp(ref6, new6, 48) == 0 && h == href;
        if (!ok) { ++bad; printf("check n=%llu mis=%d K=%d BLK=%d: bounds %s, column %s\n", (unsigned long long)pts, mis, Ks[v], Bs[v], memcmp(ref6, new6, 48) == 0 ? "identical" : "DIFFER", h == href ? "identical" : "DIFFER"); }
      }
    }
    printf("check: %d mismatches against the round-4 kernel (6 sizes x 2 alignments x 8 shapes)\n", bad);
    fflush(stdout);
  }

  if (!strcmp(what, "shapes")) {
    struct Shape { int K, B; };
    const Shape shapes[] = {{1, 256}, {2, 256}, {3, 256}, {4, 256}, {6, 256}, {12, 256}, {2, 128}, {3, 128}, {6, 128}, {12, 128}, {1, 512}, {2, 512}, {3, 512}, {1, 1024}, {2, 1024}, {3, 1024}, {2, 768}, {3, 768}, {4, 384}, {3, 384}, {6, 384}, {6, 512}, {4, 512}};
    for (int mode : {7, 4, 3}) for (uint64_t pts : {100000000ull, 1000000000ull}) {
      g_mode = mode;
      const int it = pts > 400000000ull ? 3 : iters;
      float ms = timeit([&] { pstk::launch_vec3f64_stream(src, dst0, pts, scale, offset, (unsigned)mode, partials, out6, g_s); }, it);
      printf("shapes mode=%d n=%10llu r4-kernel (with its fold launches)      %8.4f ms %7.3f TB/s\n", mode, (unsigned long long)pts, ms, bpp() * pts / ms / 1e9);
      for (const Shape& sh : shapes) {
        const unsigned max_blocks = std::min(8u, 2048u / sh.B);
        for (unsigned blocks = max_blocks; blocks >= 1; --blocks) {
          if ((uint64_t)blocks * sh.B * sh.K * 16 < 24 * 1024) continue;  // under 24 KiB in flight per CU: starved (K=3 x 2 blocks: 4.5 TB/s)
          unsigned grid; Stream2Params p = mk(src, pts, sh.K, sh.B, dst0, &grid);
          const unsigned lds = lds_for(sh.B, blocks);
          ms = timeit([&] { launch_dyn(sh.K, sh.B, p, grid, lds); }, it);
          printf("shapes mode=%d n=%10llu K=%2d BLK=%4d blocks/CU<=%u in-flight/CU=%3u KiB  %8.4f ms %7.3f TB/s\n", mode, (unsigned long long)pts, sh.K, sh.B, blocks,
                 blocks * sh.B * sh.K * 16 / 1024, ms, bpp() * pts / ms / 1e9);
        }
        fflush(stdout);
      }
    }
    g_mode = 7;
  }

  if (!strcmp(what, "spacing")) {
    const int K = argc > 2 ? atoi(argv[2]) : 3, B = argc > 4 ? atoi(argv[4]) : 512;
    const unsigned blocks = argc > 3 ? atoi(argv[3]) : 2;
    unsigned grid; Stream2Params p0 = mk(src, n, K, B, dst0, &grid);
    const uint64_t L = p0.xcd_stride, maxD = (3 * cap_pts / 2) / (K * B) / 8 - 1;
    for (uint64_t D = L; D <= maxD; D += L / 8) {
      Stream2Params p = p0; p.xcd_stride = D; p.n_vec = (7 * D + L) * (uint64_t)(K * B); p.n_doubles = 2 * p.n_vec;
      const float ms = timeit([&] { launch_dyn(K, B, p, grid, lds_for(B, blocks)); }, iters);
      printf("spacing K=%d blocks/CU<=%u D=%8llu tiles (%9.3f MiB between the XCD regions) %8.4f ms %7.3f TB/s\n", K, blocks, (unsigned long long)D, D * K * B * 16.0 / 1048576.0, ms,
             48.0 * 8 * L * K * B * 2 / 3 / ms / 1e9);
    }
    fflush(stdout);
  }

  if (!strcmp(what, "sizes")) {
    const int K = argc > 2 ? atoi(argv[2]) : 3, B = argc > 4 ? atoi(argv[4]) : 512;
    const unsigned blocks = argc > 3 ? atoi(argv[3]) : 2;
    for (int mode : {7, 3, 4}) for (int rep = 0; rep < 2; ++rep)
    for (uint64_t pts : {1000000ull, 3000000ull, 10000000ull, 25000000ull, 50000000ull, 100000000ull, 150000000ull, 200000000ull, 300000000ull, 500000000ull, 600000000ull, 800000000ull, 1000000000ull}) {
      g_mode = mode;
      const int it = pts > 400000000ull ? 3 : iters;
      float ms0 = timeit([&] { pstk::launch_vec3f64_stream(src, dst0, pts, scale, offset, (unsigned)mode, partials, out6, g_s); }, it);
      unsigned grid; Stream2Params p = mk(src, pts, K, B, dst0, &grid);
      const float ms = timeit([&] { launch_dyn(K, B, p, grid, lds_for(B, blocks)); if (mode & 4) pstk::launch_finalize_bounds(partials, grid, out6, g_s); }, it);
      printf("sizes mode=%d n=%10llu  r4 %8.4f ms %7.3f TB/s | r5 K=%d BLK=%d blocks/CU<=%u (with the fold launches) %8.4f ms %7.3f TB/s\n", mode, (unsigned long long)pts, ms0, bpp() * pts / ms0 / 1e9, K, B, blocks,
             ms, bpp() * pts / ms / 1e9);
      fflush(stdout);
    }
    g_mode = 7;
  }
  return 0;
}
