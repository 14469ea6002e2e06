// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS PATTERN (run by tools/pmc_calibrate.sh under
// `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`).  The guide establishes "FETCH_SIZE reports half the
// bytes" only for wide coalesced streaming reads (16 B per lane); gathers and scatters are calibrated here on known access counts over a
// region far larger than the 256 MiB Infinity Cache, so that every access goes to HBM.
//   stream_read16/8/4  N x 16 / 8 / 4 B, lane-contiguous                   (requested = moved)
//   gather_read<B>     N random B-byte reads at B-aligned (8-aligned for 24) addresses, B = 8, 24, 32, 64, 128 (one lane each; 64 and 128
//                      as 4 / 8 lanes x 16 B inside one aligned line)
//   stream_write16     N x 16 B stores, lane-contiguous
//   scatter_write<B>   N random B-byte stores, B = 8, 12 (4-aligned), 32 (aligned, two 16-B stores of one lane)
// Every kernel is launched 3 times; the summary divides the counter by the number of accesses.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern "C" __global__ void stream_read16(const u32x4* __restrict__ src, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(src + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// 8 bytes per lane, lane-contiguous (512 B per wave instruction): the access shape of a row copy of f64 coordinates
extern "C" __global__ void stream_read8(const uint64_t* __restrict__ src, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = src[i];
    acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
  }
  if (acc == 0x12345678u) *sink = acc;
}
// 4 bytes per lane, lane-contiguous (256 B per wave instruction): index / key columns
extern "C" __global__ void stream_read4(const uint32_t* __restrict__ src, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc ^= src[i];
  if (acc == 0x12345678u) *sink = acc;
}
extern "C" __global__ void stream_write16(u32x4* __restrict__ dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(u32x4{(uint32_t)i, 1u, 2u, 3u}, dst + i);
}
// one lane = one random access of BYTES (8, 24, 32) -- 24 is read as three 8-byte loads at an 8-aligned address, like a Vec3f64 gather
template <int BYTES>
__global__ void gather_read(const uint8_t* __restrict__ src, uint64_t region_bytes, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t align = BYTES == 24 ? 8 : BYTES;
  const uint64_t slots = (region_bytes - 64) / align;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t* p = src + (splitmix64(i * 0x51ull + BYTES) % slots) * align;
#pragma unroll
    for (int b = 0; b < BYTES; b += 8) { const uint64_t v = *reinterpret_cast<const uint64_t*>(p + b); acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
  }
  if (acc == 0x12345678u) *sink = acc;
}
// LANES consecutive lanes read one random aligned line of LANES * 16 bytes (64 or 128)
template <int LANES>
__global__ void gather_line(const uint8_t* __restrict__ src, uint64_t region_bytes, uint64_t n_lines, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t line = LANES * 16, slots = region_bytes / line;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = t / LANES; i < n_lines; i += total / LANES) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (splitmix64(i * 0x77ull + LANES) % slots) * line + (t % LANES) * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
template <int BYTES>  // 8: one 8-byte store; 12: three 4-byte stores at a 4-aligned address (a Vec3f32); 32: two 16-byte stores, aligned
__global__ void scatter_write(uint8_t* __restrict__ dst, uint64_t region_bytes, uint64_t n) {
  const uint64_t align = BYTES == 12 ? 4 : BYTES;
  const uint64_t slots = (region_bytes - 64) / align;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint8_t* p = dst + (splitmix64(i * 0x33ull + BYTES) % slots) * align;
    if (BYTES == 8) *reinterpret_cast<uint64_t*>(p) = i;
    else if (BYTES == 12) { uint32_t* q = reinterpret_cast<uint32_t*>(p); q[0] = (uint32_t)i; q[1] = 1u; q[2] = 2u; }
    else { u32x4* q = reinterpret_cast<u32x4*>(p); q[0] = u32x4{(uint32_t)i, 1u, 2u, 3u}; q[1] = u32x4{4u, 5u, 6u, 7u}; }
  }
}

int main() {
  const uint64_t region = 4ull << 30;   // 4 GiB >> 256 MiB Infinity Cache
  const uint64_t n = 50'000'000ull;     // accesses per launch
  uint8_t* buf = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&buf, region));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, region));
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_read16, grid, block, 0, 0, (const u32x4*)buf, n, sink);
    hipLaunchKernelGGL(stream_read8, grid, block, 0, 0, (const uint64_t*)buf, n, sink);
    hipLaunchKernelGGL(stream_read4, grid, block, 0, 0, (const uint32_t*)buf, n, sink);
    hipLaunchKernelGGL((gather_read<8>), grid, block, 0, 0, buf, region, n, sink);
    hipLaunchKernelGGL((gather_read<24>), grid, block, 0, 0, buf, region, n, sink);
    hipLaunchKernelGGL((gather_read<32>), grid, block, 0, 0, buf, region, n, sink);
    hipLaunchKernelGGL((gather_line<4>), grid, block, 0, 0, buf, region, n, sink);
    hipLaunchKernelGGL((gather_line<8>), grid, block, 0, 0, buf, region, n, sink);
    hipLaunchKernelGGL(stream_write16, grid, block, 0, 0, (u32x4*)buf, n);
    hipLaunchKernelGGL((scatter_write<8>), grid, block, 0, 0, buf, region, n);
    hipLaunchKernelGGL((scatter_write<12>), grid, block, 0, 0, buf, region, n);
    hipLaunchKernelGGL((scatter_write<32>), grid, block, 0, 0, buf, region, n);
    CK(hipDeviceSynchronize());
  }
  CK(hipGetLastError());
  printf("accesses_per_launch %llu region_bytes %llu\n", (unsigned long long)n, (unsigned long long)region);
  return 0;
}
