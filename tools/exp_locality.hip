// Experiment (round 3): what do a window-local gather and a bucket scatter of 24-byte points cost on this memory system?
// Decides whether a payload-carrying first (most significant digit) pass pays for the voxel grid and the kNN reorder.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_locality.hip -o /tmp/exp_locality && /tmp/exp_locality
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __host__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

struct P3 { double x, y, z; };

__global__ void make_window_index(uint32_t* idx, uint32_t n, uint32_t w) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = (i & ~(w - 1)) + (mix(i * 2654435761u + 17) & (w - 1));
    idx[i] = j < n ? j : i;
}

__global__ void __launch_bounds__(256) gather_kernel(P3* __restrict__ out, const P3* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[idx[i]];
}

// the library's reorder_kernel<UNROLL> (normals.hip): grid-stride, a 16-byte + an 8-byte load per point
template <int UNROLL>
__global__ __launch_bounds__(256) void reorder_like_kernel(const double* __restrict__ xyz, const uint32_t* __restrict__ idx, uint64_t n, double* __restrict__ sorted_xyz) {
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    const uint64_t step = (uint64_t)gridDim.x * 256 * UNROLL;
    for (uint64_t j0 = (uint64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; j0 < n; j0 += step) {
        uint64_t i[UNROLL]; d2u xy[UNROLL]; double z[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const uint64_t j = j0 + (uint64_t)u * 256; i[u] = j < n ? idx[j] : 0; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { xy[u] = *reinterpret_cast<const d2u*>(xyz + 3 * i[u]); z[u] = xyz[3 * i[u] + 2]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t j = j0 + (uint64_t)u * 256;
            if (j < n) { *reinterpret_cast<d2u*>(sorted_xyz + 3 * j) = xy[u]; sorted_xyz[3 * j + 2] = z[u]; }
        }
    }
}

__global__ void __launch_bounds__(256) scatter_kernel(P3* __restrict__ out, const P3* __restrict__ in, const uint32_t* __restrict__ dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[dst[i]] = in[i];
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 100000000u;
    P3 *a, *b; uint32_t* idx;
    CK(hipMalloc(&a, sizeof(P3) * (size_t)n)); CK(hipMalloc(&b, sizeof(P3) * (size_t)n)); CK(hipMalloc(&idx, 4 * (size_t)n));
    CK(hipMemset(a, 1, sizeof(P3) * (size_t)n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = (n + 255) / 256;
    for (uint32_t lg = 10; lg <= 28; lg += 2) {
        uint32_t w = 1u << lg;
        make_window_index<<<blocks, 256>>>(idx, n, w);
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0));
            gather_kernel<<<blocks, 256>>>(b, a, idx, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("gather  window 2^%-2u points (%8.1f MB): %7.3f ms\n", lg, w * 24.0 / 1e6, best);
        if (lg == 26) {
            for (unsigned g : {2048u, 8192u, 32768u, 0u}) {
                const unsigned grid = g ? g : (n + 511) / 512;
                float b2 = 1e9f;
                for (int r = 0; r < 4; ++r) {
                    CK(hipEventRecord(e0));
                    reorder_like_kernel<2><<<grid, 256>>>((const double*)a, idx, n, (double*)b);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); b2 = std::min(b2, ms);
                }
                printf("   reorder-like <2>, grid %8u: %7.3f ms\n", grid, b2);
            }
            for (unsigned g : {8192u, 0u}) {
                const unsigned grid = g ? g : (n + 255) / 256;
                float b2 = 1e9f;
                for (int r = 0; r < 4; ++r) {
                    CK(hipEventRecord(e0));
                    reorder_like_kernel<1><<<grid, 256>>>((const double*)a, idx, n, (double*)b);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); b2 = std::min(b2, ms);
                }
                printf("   reorder-like <1>, grid %8u: %7.3f ms\n", grid, b2);
            }
        }
    }
    // stable bucket ranks on the host
    std::vector<uint32_t> dst(n);
    for (uint32_t nb : {16u, 64u, 128u, 256u, 512u, 2048u, 8192u}) {
        std::vector<size_t> count(nb + 1, 0);
        for (uint32_t i = 0; i < n; ++i) count[mix(i ^ 0x9e3779b9u) % nb + 1]++;
        for (uint32_t d = 0; d < nb; ++d) count[d + 1] += count[d];
        for (uint32_t i = 0; i < n; ++i) dst[i] = (uint32_t)count[mix(i ^ 0x9e3779b9u) % nb]++;
        CK(hipMemcpy(idx, dst.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0));
            scatter_kernel<<<blocks, 256>>>(b, a, idx, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("scatter into %5u buckets (direct 24-byte stores): %7.3f ms\n", nb, best);
    }
    return 0;
}
