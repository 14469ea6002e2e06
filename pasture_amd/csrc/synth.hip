// Deterministic synthetic point attributes generated directly in HBM (bench / test inputs).
//   positions : u = splitmix64(seed ^ (3*g + c)); f = (u >> 11) * 2^-53; x,y in [0,1000), z in [0,100)
//   raw LAS   : X,Y,Z = splitmix64(seed ^ (3*g + c)) mod 2,000,000 (i32)
//   the rest  : e = splitmix64((seed ^ 0xD1B54A32D192ED03) ^ (g*1024 + (slot mod 32)*32 + (c mod 32)))
//               ints: low bits (ReturnNumber/NumberOfReturns & 7, ScanDirectionFlag/EdgeOfFlightLine & 1),
//               floats: unit(e) * 1000, opaque types: bytes of successive e's
// g = global point index (first_index + i), so every shard / the CPU oracle generate identical data.
// The CPU checker used by the tests restates this generator; tests/test_gpu_parity.py::test_synth_matches_oracle compares them bit for bit.
#include "device_common.hpp"
#include "kernels.hpp"

#include <algorithm>

using namespace pstd;

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double unit(uint64_t u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }
__device__ __forceinline__ uint64_t extra(uint64_t seed, uint64_t g, uint64_t slot, uint64_t c) {
  return splitmix64((seed ^ 0xD1B54A32D192ED03ull) ^ (g * 1024 + (slot % 32) * 32 + (c % 32)));
}

// kind codes = PST_* of include/pasture_amd.h
__global__ __launch_bounds__(kBlock) void synth_kernel(const pstk::SynthAttr a, uint64_t n, uint64_t seed, uint64_t first_index) {
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    const uint64_t g = first_index + i;
    gptr_t out = as_global(a.base) + i * a.stride;
    if (a.special == 1) {  // Position3D as Vec3f64 (kind 14) or Vec3f32 (kind 12)
      const double scl[3] = {1000.0, 1000.0, 100.0};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v = unit(splitmix64(seed ^ (3 * g + c))) * scl[c];
        if (a.kind == 14) store_un<double>(out + 8 * c, v);
        else store_un<float>(out + 4 * c, (float)v);
      }
      continue;
    }
    if (a.special == 2) {  // LASLocalPosition Vec3i32
#pragma unroll
      for (int c = 0; c < 3; ++c) store_un<int32_t>(out + 4 * c, (int32_t)(splitmix64(seed ^ (3 * g + c)) % 2000000ull));
      continue;
    }
    const uint64_t mask = a.special == 3 ? 7ull : (a.special == 4 ? 1ull : ~0ull);
    switch (a.kind) {
      case 0: case 1: store_un<uint8_t>(out, (uint8_t)(extra(seed, g, a.slot, 0) & mask)); break;
      case 2: case 3: store_un<uint16_t>(out, (uint16_t)(extra(seed, g, a.slot, 0) & mask)); break;
      case 4: case 5: store_un<uint32_t>(out, (uint32_t)(extra(seed, g, a.slot, 0) & mask)); break;
      case 6: case 7: store_un<uint64_t>(out, extra(seed, g, a.slot, 0) & mask); break;
      case 8: store_un<float>(out, (float)(unit(extra(seed, g, a.slot, 0)) * 1000.0)); break;
      case 9: store_un<double>(out, unit(extra(seed, g, a.slot, 0)) * 1000.0); break;
      case 10: for (int c = 0; c < 3; ++c) store_un<uint8_t>(out + c, (uint8_t)extra(seed, g, a.slot, c)); break;
      case 11: for (int c = 0; c < 3; ++c) store_un<uint16_t>(out + 2 * c, (uint16_t)extra(seed, g, a.slot, c)); break;
      case 12: for (int c = 0; c < 3; ++c) store_un<float>(out + 4 * c, (float)(unit(extra(seed, g, a.slot, c)) * 1000.0)); break;
      case 13: for (int c = 0; c < 3; ++c) store_un<uint32_t>(out + 4 * c, (uint32_t)extra(seed, g, a.slot, c)); break;
      case 14: for (int c = 0; c < 3; ++c) store_un<double>(out + 8 * c, unit(extra(seed, g, a.slot, c)) * 1000.0); break;
      default:
        for (uint32_t j = 0; j < a.size; ++j) store_un<uint8_t>(out + j, (uint8_t)(extra(seed, g, a.slot, j / 8) >> (8 * (j % 8))));
    }
  }
}

}  // namespace

namespace pstk {
void launch_synth(const SynthAttr& a, uint64_t n, uint64_t seed, uint64_t first_index, hipStream_t stream) {
  if (n == 0) return;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)device_cus() * 16);
  hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(kBlock), 0, stream, a, n, seed, first_index);
}
}  // namespace pstk
