// Tuning harness (not part of the product): sweeps variants of the columnar Vec3f64 stream kernel on one GPU and prints
// achieved GB/s.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pasture_amd/csrc tools/tune_stream.hip -o /tmp/tune_stream
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace pstk { int device_cus() { return 256; } }
#include "../pasture_amd/csrc/stream.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool A, bool W, bool B, int K, bool NTL, bool NTS>
float run(const StreamParams& p, unsigned grid, int iters, hipStream_t s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((vec3f64_stream_kernel<A, W, B, K, NTL, NTS>), dim3(grid), dim3(kBlock), 0, s, p);
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((vec3f64_stream_kernel<A, W, B, K, NTL, NTS>), dim3(grid), dim3(kBlock), 0, s, p);
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
  double *src, *dst, *partials;
  CK(hipMalloc(&src, n * 24)); CK(hipMalloc(&dst, n * 24)); CK(hipMalloc(&partials, 64 << 20));
  CK(hipMemset(src, 0x3f, n * 24)); CK(hipMemset(dst, 0, n * 24));
  // random-ish data so DVFS behaves like the real workload
  {
    pstk::SynthAttr a{}; (void)a;
  }
  hipStream_t s; CK(hipStreamCreate(&s));
  StreamParams p{};
  p.src = src; p.dst = dst; p.n_doubles = 3 * n; p.vec_first = 0; p.n_vec = p.n_doubles / 2;
  for (int c = 0; c < 3; ++c) { p.scale[c] = 0.001; p.offset[c] = 5000.0 * (c + 1); }
  p.partials = partials;
  const int iters = 20;
  auto report = [&](const char* name, int k, bool ntl, bool nts, unsigned grid, float ms, double bytes_pp) {
    printf("%-10s K=%2d ntl=%d nts=%d grid=%6u  %8.4f ms  %8.1f GB/s\n", name, k, ntl, nts, grid, ms, bytes_pp * n / ms / 1e6);
    fflush(stdout);
  };
  const unsigned cus = 256;
  std::vector<unsigned> grids = {cus * 2, cus * 4, cus * 6, cus * 8, cus * 12, cus * 16, cus * 32, 0 /* one block per tile */};
#define SWEEP(K, NTL, NTS)                                                                                       \
  for (unsigned g : grids) {                                                                                     \
    unsigned tiles = (unsigned)((p.n_vec + (K) * kBlock - 1) / ((K) * kBlock));                                  \
    unsigned grid = g ? std::min(g, tiles) : tiles;                                                              \
    report("cvt+aabb", K, NTL, NTS, grid, run<true, true, true, K, NTL, NTS>(p, grid, iters, s), 48);           \
  }
  SWEEP(3, true, true) SWEEP(6, true, true) SWEEP(9, true, true) SWEEP(12, true, true)
  SWEEP(6, false, false) SWEEP(6, true, false) SWEEP(6, false, true)
  SWEEP(12, false, false)
#define SWEEPB(K, NTL)                                                                                           \
  for (unsigned g : grids) {                                                                                     \
    unsigned tiles = (unsigned)((p.n_vec + (K) * kBlock - 1) / ((K) * kBlock));                                  \
    unsigned grid = g ? std::min(g, tiles) : tiles;                                                              \
    report("aabb", K, NTL, false, grid, run<false, false, true, K, NTL, false>(p, grid, iters, s), 24);         \
  }
  SWEEPB(6, true) SWEEPB(12, true) SWEEPB(6, false) SWEEPB(12, false)
  return 0;
}
