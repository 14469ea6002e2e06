// Tuning harness #3: read-only AABB variants on random data.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
namespace pstk { int device_cus() { return 256; } }
#include "../pasture_amd/csrc/stream.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
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
  double *src, *partials;
  CK(hipMalloc(&src, n * 24)); CK(hipMalloc(&partials, 256 << 20));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, 3 * n);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  StreamParams p{}; p.src = src; p.dst = src; p.n_doubles = 3 * n; p.vec_first = 0; p.n_vec = p.n_doubles / 2;
  for (int c = 0; c < 3; ++c) { p.scale[c] = 1; p.offset[c] = 0; }
  p.partials = partials;
  const int iters = 20;
  for (int rep = 0; rep < 2; ++rep) {
#define RUN(K, NTL, G) { unsigned tiles = (unsigned)((p.n_vec + (K) * 256 - 1) / ((K) * 256)); unsigned g = (G) ? std::min<unsigned>(G, tiles) : tiles; \
    float ms = timeit([&] { hipLaunchKernelGGL((vec3f64_stream_kernel<false, false, true, K, NTL, false>), dim3(g), dim3(256), 0, s, p); }, iters, s); \
    printf("aabb K=%2d ntl=%d grid=%7u %8.4f ms %8.1f GB/s\n", K, NTL, g, ms, 24.0 * n / ms / 1e6); fflush(stdout); }
    RUN(6, true, 512) RUN(6, true, 768) RUN(6, true, 1024) RUN(6, true, 1280) RUN(6, true, 1536) RUN(6, true, 1792) RUN(6, true, 2048) RUN(6, true, 4096) RUN(6, true, 0)
    RUN(12, true, 256) RUN(12, true, 512) RUN(12, true, 768) RUN(12, true, 1024) RUN(12, true, 1280) RUN(12, true, 2048) RUN(12, true, 0)
    RUN(3, true, 1024) RUN(3, true, 2048) RUN(3, true, 4096)
    RUN(6, false, 1024) RUN(12, false, 512)
  }
  return 0;
}
