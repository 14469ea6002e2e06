// Local scale of a point cloud, measured without an index: for 512 sampled points, a histogram of their squared distances to
// a subsample of the cloud (one bin per octave of d^2 -- the f32 exponent --, 80 bins below the bounding box's diagonal: 2^40 in distance).  From each histogram: the radius at which the
// subsample holds T points around that query, the local dimension D (how fast the count grows from T to 4 T), and from those the radius at
// which the FULL cloud holds M points, r_M = r_T (M / (T f))^(1/D) with f = points per subsample point.  The MEDIAN over the queries is the
// cell edge the kNN search grids with.
//
// Why not the bounding box's volume: it is right only for clouds that fill their box.  A surface in a 3-D box is 2-4 times off, two scans a
// long way apart are off by orders of magnitude -- every point of a cluster then lands in one cell and the search degenerates into a brute
// force (2.2 s for 2 x 10^6 points in two balls 800 diameters apart).  A probe of an index built with the wrong edge saturates in exactly
// those cases; distances to nearest neighbours do not care what the box looks like.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.hpp"
#include "normals_host.hpp"

namespace {

constexpr int kBins = 80;          // one per octave of the squared distance: a factor 2^40 in distance below the diagonal (at two per octave a
                                   // cluster 10^-9 of the box across fell into bin 0 whole)
constexpr int kTile = 768;         // candidates staged per step
constexpr int kQPerBlock = 256;    // one query per thread

// cand: packed xyz of the subsample.  Query j is candidate j * q_stride.  Block (x, y): queries [256 x, 256 x + 256) against the candidate
// slice [y * slice, (y + 1) * slice).  Per-thread histograms live in LDS (bin-major: no bank conflicts), merged with global atomics.
__global__ __launch_bounds__(kQPerBlock) void knn_scale_kernel(const double* __restrict__ cand, uint32_t n_c, uint32_t n_q, uint32_t q_stride, uint32_t slice,
                                                               int bin_off, unsigned int* __restrict__ hist) {
  __shared__ double tile[3 * kTile];
  // per-thread histograms, 16-bit counters (a slice has at most 4096 candidates) packed in pairs and advanced with ds_add_u32: an LDS atomic
  // that returns nothing has no read -> add -> write chain through registers (the u16 read-modify-write loop was waiting on itself: 2.3 ms for
  // 512 queries x 2^20 candidates at two waves per SIMD)
  __shared__ uint32_t H[(kBins / 2) * kQPerBlock];
  static_assert(kBins % 2 == 0, "bins are packed in pairs");
  const uint32_t t = threadIdx.x;
  const uint32_t qi = blockIdx.x * kQPerBlock + t;
  const bool has_q = qi < n_q;
  const uint64_t qc = has_q ? (uint64_t)qi * q_stride : 0;
  const double qx = cand[3 * qc], qy = cand[3 * qc + 1], qz = cand[3 * qc + 2];
#pragma unroll
  for (int b = 0; b < kBins / 2; ++b) H[b * kQPerBlock + t] = 0;
  const uint32_t c0 = blockIdx.y * slice, c1 = min(n_c, c0 + slice);
  for (uint32_t base = c0; base < c1; base += kTile) {
    const uint32_t cnt = min((uint32_t)kTile, c1 - base);
    __syncthreads();
    for (uint32_t e = t; e < 3 * cnt; e += kQPerBlock) tile[e] = cand[3ull * base + e];
    __syncthreads();
#pragma unroll 4
    for (uint32_t c = 0; c < cnt; ++c) {  // every lane reads the same candidate: LDS broadcast
      const double dx = tile[3 * c] - qx, dy = tile[3 * c + 1] - qy, dz = tile[3 * c + 2] - qz;
      const float d2 = (float)__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));  // (an estimate: fused multiply-adds are welcome here)
      // bits >> 23 = the biased exponent: bin edges at powers of two.  0 (the query itself, duplicates) and everything below the range go
      // to bin 0; NaN / inf (non-finite coordinates) come out above the range and are not counted.
      const int b = (int)(__float_as_uint(d2) >> 23) - bin_off;
      if (b < kBins) {
        const int bb = b < 0 ? 0 : b;
        atomicAdd(&H[(bb >> 1) * kQPerBlock + t], 1u << (16 * (bb & 1)));
      }
    }
  }
  if (has_q) {
#pragma unroll 4
    for (int b = 0; b < kBins; ++b) {
      const unsigned int v = (H[(b >> 1) * kQPerBlock + t] >> (16 * (b & 1))) & 0xFFFFu;
      if (v) atomicAdd(&hist[(uint64_t)qi * kBins + b], v);
    }
  }
}

float bin_upper_edge(int b, int bin_off) {  // d^2 values below this fall into bins 0 .. b
  const uint32_t bits = (uint32_t)(b + bin_off + 1) << 23;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

}  // namespace

namespace pstk {

size_t knn_scale_scratch_bytes() { return (size_t)512 * kBins * sizeof(unsigned int); }

// cand: n_c packed points (a subsample: every f-th point of the cloud, f = `thinning`), diag2 = squared diagonal of the bounding box.
// Returns false when nothing could be measured (HIP failure, fewer than 64 usable points).  h_m: median radius at which the full cloud holds
// m_target points; dim: median local dimension in [1, 3].
bool knn_scale_estimate(const double* cand, uint32_t n_c, double thinning, double diag2, double m_target, unsigned int* scratch, hipStream_t stream,
                        double& h_m, double& dim, uint32_t max_queries) {
  if (n_c < 64 || !(diag2 > 0.0)) return false;
  const uint32_t n_q = std::min<uint32_t>(std::min<uint32_t>(512, max_queries), n_c), q_stride = n_c / n_q;  // (2048 queries: 2.4 ms for 2^20 candidates, the same median)
  const float top = (float)diag2;
  uint32_t top_bits;
  std::memcpy(&top_bits, &top, 4);
  const int bin_off = (int)(top_bits >> 23) - (kBins - 1);  // the diagonal falls into the last bin
  if (hipMemsetAsync(scratch, 0, (size_t)n_q * kBins * sizeof(unsigned int), stream) != hipSuccess) return false;
  // candidates per block: at most 4096 (16-bit counters), and few enough that the launch fills the chip -- the quick estimate (2^17 candidates,
  // 256 queries) ran on 32 workgroups: 0.52 ms with 224 CUs idle; now 256 slices of 512
  uint32_t slice = 4096;
  {
    const uint32_t q_blocks = (n_q + kQPerBlock - 1) / kQPerBlock;
    while (slice > 512 && (uint64_t)q_blocks * ((n_c + slice - 1) / slice) < 512) slice >>= 1;
  }
  hipLaunchKernelGGL(knn_scale_kernel, dim3((n_q + kQPerBlock - 1) / kQPerBlock, (n_c + slice - 1) / slice), dim3(kQPerBlock), 0, stream, cand, n_c, n_q, q_stride,
                     slice, bin_off, scratch);
  std::vector<unsigned int> h((size_t)n_q * kBins);
  if (hipMemcpyAsync(h.data(), scratch, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    return false;
  // T points of the subsample (the query itself included) -- small enough to stay local, large enough to beat the counting noise
  const double T = 12.0;
  std::vector<double> radii, dims;
  radii.reserve(n_q); dims.reserve(n_q);
  auto crossing = [&](const unsigned int* hq, double target, double& d2_at) {  // squared distance at which the cumulative count reaches target
    double cum = 0.0;
    for (int b = 0; b < kBins; ++b) {
      const double next = cum + (double)hq[b];
      if (next >= target) {
        const double hi = (double)bin_upper_edge(b, bin_off);
        const double lo = b == 0 ? hi * 0.5 : (double)bin_upper_edge(b - 1, bin_off);
        // counts grow like a power of the distance inside a bin: interpolate in log-log
        const double c0 = std::fmax(cum, 0.5), c1 = next;
        const double w = c1 > c0 ? std::log(target / c0) / std::log(c1 / c0) : 1.0;
        d2_at = lo * std::pow(hi / lo, std::fmin(1.0, std::fmax(0.0, w)));
        return true;
      }
      cum = next;
    }
    return false;
  };
  for (uint32_t q = 0; q < n_q; ++q) {
    const unsigned int* hq = &h[(size_t)q * kBins];
    double d2_t, d2_4t;
    if (!crossing(hq, T, d2_t) || !crossing(hq, 4.0 * T, d2_4t) || !(d2_4t > d2_t) || !(d2_t > 0.0)) continue;
    double D = 2.0 * std::log(4.0) / std::log(d2_4t / d2_t);
    D = std::fmin(3.0, std::fmax(1.0, D));
    radii.push_back(std::sqrt(d2_t) * std::pow(m_target / (T * thinning), 1.0 / D));
    dims.push_back(D);
  }
  if (radii.size() < 16) return false;
  std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
  std::nth_element(dims.begin(), dims.begin() + dims.size() / 2, dims.end());
  h_m = radii[radii.size() / 2];
  dim = dims[dims.size() / 2];
  return h_m > 0.0 && std::isfinite(h_m);
}

}  // namespace pstk
