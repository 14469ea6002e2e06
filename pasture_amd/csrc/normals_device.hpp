// Device code shared by the kNN normal-estimation kernels (normals.hip: search over global memory; normals_tile.hip: search over an
// LDS-staged box of grid cells): the uniform grid, the register-resident k-best list and the plane fit of
// pasture-algorithms/src/normal_estimation.rs:198-467.
#pragma once
#include "device_common.hpp"

namespace pstn {

using namespace pstd;

constexpr uint64_t kInvalidKey = ~0ull;
constexpr uint32_t kNoIndex = 0xFFFFFFFFu;

struct GridParams {
  double org[3];   // grid origin (min corner of the finite points)
  double inv_h;    // 1 / cell edge along y and z
  double h;        // cell edge along y and z
  double inv_hx;   // 1 / cell edge along x
  double hx;       // cell edge along x = h / rx: with the dense directory the points of a grid row are sorted by x at this granularity
  uint32_t rx;     // fine x cells per cell edge h (1 = cubic cells)
  uint32_t dim[3]; // cells per axis (<= 2^21)
  uint32_t dense;  // 1: keys are row-major cell numbers (x fastest) with a dense cell_start directory; 0: Morton keys + hash table
  uint32_t rotated;  // 1: cells are assigned in a rotated frame (u, v, w) = rot * (x, y, z) -- the cloud's principal axes --, org / dim refer to it
  double rot[9];     // row-major orthonormal matrix (distances are ALWAYS computed from the original coordinates)
  double rot_c[3];   // the point the rotation turns about (near the cloud's centre: rotated coordinates are then of the size of the cloud, and
                     // their rounding error -- a few ulps of that size -- stays far below the slack of every bound that uses them)
};
// The frame the grid lives in.  A rotation keeps every coordinate difference within the Euclidean distance, so all bounds of the searches
// (a point within distance d of a query lies within d of it along every grid axis) hold in it as they do along the cloud's own axes.
__device__ __forceinline__ void grid_frame(const GridParams& g, double x, double y, double z, double& u, double& v, double& w) {
  if (g.rotated) {
    const double dx = x - g.rot_c[0], dy = y - g.rot_c[1], dz = z - g.rot_c[2];
    u = g.rot[0] * dx + g.rot[1] * dy + g.rot[2] * dz;
    v = g.rot[3] * dx + g.rot[4] * dy + g.rot[5] * dz;
    w = g.rot[6] * dx + g.rot[7] * dy + g.rot[8] * dz;
  } else { u = x; v = y; w = z; }
}
__device__ __forceinline__ double grid_edge(const GridParams& g, int axis) { return axis == 0 ? g.hx : g.h; }

__device__ __forceinline__ bool finite3(double x, double y, double z) {
  return __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
}
__device__ __forceinline__ uint32_t cell_coord(double v, double org, double inv_h, uint32_t dim) {
  double c = __builtin_floor((v - org) * inv_h);
  if (!(c > 0.0)) c = 0.0;
  const double top = (double)(dim - 1);
  if (c > top) c = top;
  return (uint32_t)c;
}

// ---- k-best list, sorted ascending, fully in registers --------------------------------------------------------
template <int K>
struct KBest {
  double d[K];
  uint32_t i[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < K; ++t) { d[t] = __builtin_inf(); i[t] = kNoIndex; }
  }
  // Insert (dist, index) keeping ascending order; an equal distance goes AFTER the existing ones (first found wins).
  // Distances: new[t] = min(d[t], max(d[t-1], dist)) — two f64 ops per slot instead of compare + 64-bit selects (the search is
  // VALU-bound: ~145 candidates per query, every accepted one walks all K slots).  Indices follow the same three cases through
  // keep[t] = d[t] <= dist (monotone in t because the list is sorted).  Distances are never NaN here.
  __device__ __forceinline__ void insert(double dist, uint32_t index) {
    if (!(dist < d[K - 1])) return;
    bool keep_prev = true;  // "d[-1] <= dist"
    double d_prev = -__builtin_inf();
    uint32_t i_prev = index;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const double dt = d[t];
      const uint32_t it = i[t];
      const bool keep = dt <= dist;
      d[t] = __builtin_fmin(dt, __builtin_fmax(d_prev, dist));
      i[t] = keep ? it : (keep_prev ? index : i_prev);
      keep_prev = keep;
      d_prev = dt;
      i_prev = it;
    }
  }
  __device__ __forceinline__ double kth(uint32_t k) const {  // d[k-1] without dynamic register indexing
    double v = d[K - 1];
#pragma unroll
    for (int t = 0; t < K; ++t) v = (uint32_t)t == k - 1 ? d[t] : v;
    return v;
  }
};

// ---- plane fit, normal_estimation.rs:198-467, on neighbours visited in ascending-distance order ------------------
struct Fit { double nx, ny, nz, curvature; int ok; };

// cos(theta) and sin(theta) for theta = atan2(s, b) / 3 with s >= 0 -- the angle of the reference's trigonometric cubic solver
// (normal_estimation.rs:364-371: `theta = f64::atan2(f64::sqrt(-q), half_beta) * one_third`, then cos / sin of it).  The device libm's atan2, cos
// and sin are general-purpose (quadrant logic, huge-argument reduction, table constants): about 450 vector instructions per query, a twelfth of
// the box search's budget.  Here the ranges are known -- phi = atan2(s, b) in [0, pi], theta in [0, pi / 3] -- so: ONE division
// (u = num / den, or (num - den) / (num + den) above tan(pi / 8)), an odd polynomial for atan on |u| <= tan(pi / 8) (degree 11 in u^2, fitted
// at Chebyshev nodes against 50-digit references: 2.2e-16 relative), and the Taylor polynomials of cos and sin on [0, pi / 3] (nine terms:
// 1.1e-16 absolute / 2.1e-16 relative) -- each within an ulp or two of the correctly rounded value, like the functions they replace; the
// results enter the eigenvalue exactly as before.  PST_FIT_LIBM_TRIG (compile time) restores the libm calls for an A/B.
// The polynomials' coefficients live in constant memory and reach the arithmetic as SCALAR operands (s_load into an SGPR pair, one per fused
// multiply-add): written as literals, the compiler hoisted all thirty of them out of the box kernel's chunk loop into 60 vector registers and
// spilled those -- 5.6 GB of scratch write-back per 10^8-point launch for constants (rocprofv3 WRITE_SIZE of the box kernel 9.4 -> 15.4 GB when the
// polynomials replaced libm), and scratch reloads inside the fit.
#ifndef PST_FIT_LITERAL_CONSTANTS
static __device__ __constant__ double kThirdAngleTable[30] = {
    // atan(u) / u - 1 in u^2, degree 11 (highest first)
    0.011133722158619984, -0.029802426579855917, 0.043605084067176156, -0.05187003409810214, 0.058727581474367595, -0.06665857298205219,
    0.07692262470384811, -0.09090907471736968, 0.11111111076292801, -0.14285714285314552, 0.1999999999999803, -0.3333333333333335,
    // cos: 1 + w (-1/2! + w (1/4! - ... + w / 18!)), highest first
    1.0 / 6402373705728000.0, -1.0 / 20922789888000.0, 1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5,
    // sin: theta + theta w (-1/3! + w (1/5! - ... - w / 19!)), highest first
    -1.0 / 121645100408832000.0, 1.0 / 355687428096000.0, -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0,
    -1.0 / 6.0};
// (read through an index the optimiser cannot see through -- an `s_mov_b32 sN, 0` it takes for a run-time value --: a table it can read at compile
//  time is folded back into literals and hoisted again)
__device__ __forceinline__ double third_angle_coefficient(int i) {
  uint32_t z;
  asm("s_mov_b32 %0, 0" : "=s"(z));
  return kThirdAngleTable[(uint32_t)i + z];
}
#define PST_TA(i) third_angle_coefficient(i)
#else
#define PST_TA(i) kThirdAngleLiterals[i]
constexpr double kThirdAngleLiterals[30] = {
    0.011133722158619984, -0.029802426579855917, 0.043605084067176156, -0.05187003409810214, 0.058727581474367595, -0.06665857298205219,
    0.07692262470384811, -0.09090907471736968, 0.11111111076292801, -0.14285714285314552, 0.1999999999999803, -0.3333333333333335,
    1.0 / 6402373705728000.0, -1.0 / 20922789888000.0, 1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5,
    -1.0 / 121645100408832000.0, 1.0 / 355687428096000.0, -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0,
    -1.0 / 6.0};
#endif
__device__ __forceinline__ void cos_sin_third_angle(double s, double b, double& ct, double& st) {
#ifdef PST_FIT_LIBM_TRIG
  const double theta = ::atan2(s, b) * (1.0 / 3.0);
  ct = ::cos(theta); st = ::sin(theta);
#else
  const double ab = __builtin_fabs(b);
  const bool swap = s > ab;                                 // the larger one is the denominator: t = num / den in [0, 1]
  const double num = swap ? ab : s, den = swap ? s : ab;
  const bool hi = num > 0.41421356237309503 * den;          // t > tan(pi / 8): atan(t) = pi / 4 + atan((t - 1) / (t + 1))
  const double un = hi ? num - den : num, ud = hi ? num + den : den;
  const double u = ud > 0.0 ? un / ud : 0.0;                // (s = b = 0: phi = 0 or pi by the sign of b, below)
  const double z = u * u;
  double p = PST_TA(0);
#pragma unroll
  for (int i = 1; i < 12; ++i) p = __builtin_fma(p, z, PST_TA(i));
  double a = __builtin_fma(u * z, p, u);                    // atan(u)
  if (hi) a += 0.7853981633974483;                          // + pi / 4
  double phi = swap ? 1.5707963267948966 - a : a;           // atan2(s, |b|)
  if (__builtin_signbit(b)) phi = 3.141592653589793 - phi;  // atan2(s, b), b < 0 (and atan2(0, -0) = pi)
  const double theta = phi * (1.0 / 3.0);
  const double w = theta * theta;
  double c = PST_TA(12);
#pragma unroll
  for (int i = 13; i < 21; ++i) c = __builtin_fma(c, w, PST_TA(i));
  ct = __builtin_fma(-w, c, 1.0);
  double q = PST_TA(21);
#pragma unroll
  for (int i = 22; i < 30; ++i) q = __builtin_fma(q, w, PST_TA(i));
  st = __builtin_fma(theta * w, q, theta);
#endif
}

// Everything of the fit behind the covariance matrix (upper triangle, NOT divided by the count): eigen_3x3 :429-453 with solve_polynomial
// :328-392, get_largest_eigen_vector :395-426, solve_plane_parameter :456-467.
//
// `ill` (optional): set when the reference's solver AMPLIFIES last-bit differences of the covariance beyond the parity window, i.e. when a
// covariance that is not the reference's own sequence of operations (plane_fit_pivot) must not be trusted for this neighbourhood:
//   * the two SMALLEST roots of the characteristic cubic nearly coincide (prolate / collinear neighbourhoods: half_beta >= 0 and
//     -q <= 1e-3 |alpha/3|^3, i.e. |sin 3 theta| <= 0.03): the trigonometric form takes the square root of the discriminant q, an error
//     delta in q becomes delta / (2 sqrt(-q)) in the roots -- beyond the threshold a factor > 16 on the ~1e-15 relative difference between two
//     valid summation orders, against a window of 1e-13 x scale.  The threshold is a price list (4 10^6 uniform points, k = 16): 1e-4 flags
//     0.004 % of the queries, 1e-2 0.4 % -- and each flagged query costs an exact search (10^8 points: +2.6 % of the call at 1e-2,
//     profiles/r05_abab.txt).  (The other double root, theta = pi / 3, is harmless: the smallest root is the
//     simple one there and its derivative with respect to theta vanishes.)
//   * all three roots nearly coincide (isotropic neighbourhoods, e.g. a lattice point with its six face neighbours): |alpha/3| <= 1e-5 (k2/3)^2;
//     rho = sqrt(-alpha/3) then turns a relative 1e-16 into 1e-8.
//   * the matrix is nearly of rank one (collinear points): every cross product of two rows is a difference of nearly equal products, and the
//     normal -- the largest of them -- carries a relative error of ~1e-16 / (lambda_1 / lambda_2); flagged when its norm in the scaled
//     matrix is below 1e-4.
//   * two of the three cross products have norms within 1e-6 of each other: "the first maximum wins" (:395-426) is then decided by last bits,
//     and on exactly planar neighbourhoods the candidates are parallel but may point in opposite directions (found by the round-5 structured
//     volume test on a quantised plane: normal = -oracle's).
// A flagged query is handed to the exact search behind the box search, whose fit adds in the reference's order of operations (knn_tile2_kernel).
//
// FAST (the one-pass fit of the box search only; every exact search keeps the reference's sequence): the parts of the solver whose rounding does NOT
// feed the amplifying steps are taken in a cheaper form -- the nine entries are scaled by ONE reciprocal (1 / scale, then six products) instead of six
// IEEE divisions, and the three cross products are ranked by their SQUARED norms (three f64 square roots less).  Both change the scaled entries / the
// ranking by a few 1e-16 relative, the size of the difference between the two summation orders this instance already carries; a ranking that close is
// flagged by the tie test below (in squared norms: 2e-6) and handed to the exact search, like every other neighbourhood where last bits decide.
// 542 -> 430 vector instructions for the fit (round 5).
template <bool FAST = false>
__device__ __forceinline__ Fit fit_from_covariance(double c00, double c01, double c02, double c11, double c12, double c22, bool* ill = nullptr) {
  Fit f{0, 0, 0, 0, 1};
  bool flagged = false;
  const double c10 = c01, c20 = c02, c21 = c12;
  // eigen_3x3 :429-453
  double scale = __builtin_fabs(c00);  // covariance_matrix.abs().max(), column-major order
  {
    const double a[8] = {c10, c20, c01, c11, c21, c02, c12, c22};
#pragma unroll
    for (int q = 0; q < 8; ++q) { const double v = __builtin_fabs(a[q]); if (v > scale) scale = v; }
  }
  double s00, s01, s02, s11, s12, s22;
  if constexpr (FAST) {
    const double inv = 1.0 / scale;
    s00 = c00 * inv; s01 = c01 * inv; s02 = c02 * inv; s11 = c11 * inv; s12 = c12 * inv; s22 = c22 * inv;
  } else {
    s00 = c00 / scale; s01 = c01 / scale; s02 = c02 / scale; s11 = c11 / scale; s12 = c12 / scale; s22 = c22 / scale;
  }
  const double s10 = s01, s20 = s02, s21 = s12;
  // solve_polynomial on the UNSCALED matrix :328-392
  double ev0, ev1, ev2;
  {
    const double k0 = c00 * c11 * c22 + 2.0 * c01 * c02 * c12 - c00 * c12 * c12 - c11 * c02 * c02 - c22 * c01 * c01;
    const double k1 = c00 * c11 - c01 * c01 + c00 * c22 - c02 * c02 + c11 * c22 - c12 * c12;
    const double k2 = c00 + c11 + c22;
    auto quadratic = [&]() {  // :308-325
      ev0 = 0.0;
      double delta = k2 * k2 - 4.0 * k1;
      if (delta < 0.0) delta = 0.0;
      const double sd = __builtin_sqrt(delta);
      ev2 = 0.5 * (k2 + sd);
      ev1 = 0.5 * (k2 - sd);
    };
    if (__builtin_fabs(k0) < 2.220446049250313e-16) {
      quadratic();
    } else {
      const double one_third = 1.0 / 3.0;
      const double sqrt_3 = __builtin_sqrt(3.0);
      const double k2_third = k2 * one_third;
      double alpha_third = (k1 - k2 * k2_third) * one_third;
      if (alpha_third > 0.0) alpha_third = 0.0;
      const double half_beta = 0.5 * (k0 + k2_third * (2.0 * k2_third * k2_third - k1));
      double q = half_beta * half_beta + alpha_third * alpha_third * alpha_third;
      if (q > 0.0) q = 0.0;
      const double a3 = -alpha_third;
      flagged = (half_beta >= 0.0 && -q <= 1e-3 * (a3 * a3 * a3)) || a3 <= 1e-5 * (k2_third * k2_third);
      const double rho = __builtin_sqrt(-alpha_third);
      double ct, st;
      cos_sin_third_angle(__builtin_sqrt(-q), half_beta, ct, st);
      double a = k2_third + 2.0 * rho * ct;
      double b = k2_third - rho * (ct + sqrt_3 * st);
      double c = k2_third - rho * (ct - sqrt_3 * st);
      // sort ascending (:384-386)
      if (b < a) { const double t = a; a = b; b = t; }
      if (c < b) { const double t = b; b = c; c = t; }
      if (b < a) { const double t = a; a = b; b = t; }
      ev0 = a; ev1 = b; ev2 = c;
      if (ev0 <= 0.0) quadratic();
    }
    (void)ev1; (void)ev2;
  }
  const double eigen_value = ev0 * scale;  // "undo scale" :443 (sic)
  // :446-449 subtracts ev0 from a COPY of the diagonal: no effect on the scaled matrix
  // get_largest_eigen_vector :395-426: rows r0 x r1, r0 x r2, r1 x r2; first maximum of the L2 norm wins
  const double a0 = s01 * s12 - s02 * s11, a1 = s02 * s10 - s00 * s12, a2 = s00 * s11 - s01 * s10;
  const double b0 = s01 * s22 - s02 * s21, b1 = s02 * s20 - s00 * s22, b2 = s00 * s21 - s01 * s20;
  const double d0 = s11 * s22 - s12 * s21, d1 = s12 * s20 - s10 * s22, d2 = s10 * s21 - s11 * s20;
  double na = a0 * a0 + a1 * a1 + a2 * a2, nb = b0 * b0 + b1 * b1 + b2 * b2, nd = d0 * d0 + d1 * d1 + d2 * d2;  // FAST: squared norms throughout
  if constexpr (!FAST) { na = __builtin_sqrt(na); nb = __builtin_sqrt(nb); nd = __builtin_sqrt(nd); }
  f.nx = a0; f.ny = a1; f.nz = a2;
  double best = na;
  if (nb > best) { f.nx = b0; f.ny = b1; f.nz = b2; best = nb; }
  if (nd > best) { f.nx = d0; f.ny = d1; f.nz = d2; best = nd; }
  if (ill) {
    // ... and the winner among the three cross products must be the reference's: with two norms within 1e-6 of each other (exact ties on
    // lattices and quantised planes, where the candidates are parallel and may point in OPPOSITE directions) last bits decide it
    const double lo = __builtin_fmin(na, __builtin_fmin(nb, nd)), mid = (na + nb + nd) - best - lo;
    *ill = flagged || !(best >= (FAST ? 1e-8 : 1e-4)) || (best - mid) <= (FAST ? 2e-6 : 1e-6) * best;
  }
  // solve_plane_parameter :456-467
  const double eigen_sum = c00 + c11 + c22;
  f.curvature = eigen_sum != 0.0 ? __builtin_fabs(eigen_value / eigen_sum) : 0.0;
  return f;
}

// KMAX > 0: the neighbour list lives in registers (get(t) selects among KMAX of them): the loops over t are unrolled so that t is a
// compile-time constant and the selection folds away; the order of the floating-point sums is unchanged.
// FINITE: the caller guarantees finite coordinates (the grid search only ever sees the finite points): the NaN / finiteness tests per
// neighbour, which then cannot change anything, are not compiled in.
template <int KMAX = 0, bool FINITE = false, typename GetPoint>
__device__ __forceinline__ Fit plane_fit(uint32_t m, GetPoint&& get) {
  Fit f{0, 0, 0, 0, 1};
  auto for_each = [&](auto&& body) __attribute__((always_inline)) {
    if constexpr (KMAX > 0) {
#pragma unroll
      for (int t = 0; t < KMAX; ++t) if ((uint32_t)t < m) body((uint32_t)t);
    } else {
      for (uint32_t t = 0; t < m; ++t) body(t);
    }
  };
  // is_dense :133-140 (any NaN coordinate => the "not dense" path that skips non-FINITE points) and compute_centroid :198-237 in ONE
  // pass over the neighbours (each pass re-gathers 16 points): both candidate sums are accumulated in point order -- over all points
  // (the dense path) and over the finite ones (the other path) -- and the one `dense` selects is used, so every sum is the same sequence
  // of additions as in the reference.
  bool dense = true;
  double ax = 0, ay = 0, az = 0, fx = 0, fy = 0, fz = 0;
  long long cnt = 0;
  for_each([&](uint32_t t) __attribute__((always_inline)) {
    double x, y, z; get(t, x, y, z);
    ax += x; ay += y; az += z;
    if constexpr (!FINITE) {
      if (x != x || y != y || z != z) dense = false;
      if (finite3(x, y, z)) { fx += x; fy += y; fz += z; cnt += 1; }
    }
  });
  const double sx = dense ? ax : fx, sy = dense ? ay : fy, sz = dense ? az : fz;
  const double div = dense ? (double)m : (double)cnt;
  const double cx = sx / div, cy = sy / div, cz = sz / div;
  // compute_covariance_matrix :240-305 (upper triangle, NOT divided by the count)
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  long long used = 0;
  for_each([&](uint32_t t) __attribute__((always_inline)) {
    double x, y, z; get(t, x, y, z);
    if (FINITE || dense || finite3(x, y, z)) {
      double d0 = x - cx, d1 = y - cy, d2 = z - cz;
      c11 += d1 * d1; c12 += d1 * d2; c22 += d2 * d2;
      const double dx = d0;
      d0 *= dx; d1 *= dx; d2 *= dx;
      c00 += d0; c01 += d1; c02 += d2;
      used += 1;
    }
  });
  if ((dense ? (long long)m : used) < 3) { f.ok = 0; return f; }  // Err(...) :293-295 -> unwrap panic :471
  return fit_from_covariance(c00, c01, c02, c11, c12, c22);
}

// The same fit from ONE pass over the neighbours: sums about a PIVOT near the neighbourhood (the query itself), u_t = p_t - pivot,
//   S = sum u_t,  M = sum u_t u_t^T,  covariance = M - S S^T / m   (= sum (p_t - centroid)(p_t - centroid)^T, :240-305, in exact arithmetic).
// Why: the reference's two passes (centroid, then the moments of p_t - centroid) need every neighbour TWICE -- gathered twice, or held in
// 6 k registers per lane, which at k = 16 is 96 of a 128-register budget and went through scratch memory (33.5 GB of write-back per 10^8
// points, round 3).  This form holds 12 running sums and whatever loads are in flight.  Accuracy: |u_t| is at most the k-th neighbour
// distance r, so every product and sum carries an absolute rounding error of ~eps m r^2 -- the same size as the errors of the reference's
// own sum of (p_t - c)^2 terms, which are of that magnitude too; the subtraction M - S S^T / m cancels at most a factor
// (sigma^2 + mu^2) / sigma^2 with |mu| <= r (the pivot lies inside the neighbourhood), not the catastrophic |p|^2 / sigma^2 of raw moments.
// The sums are NOT the reference's sequence of operations (12 instead of 18 f64 operations per neighbour, fused multiply-adds): results agree
// to ~1e-14 relative, inside the north star's 1e-9; PST_KNN_FIT=seq selects the reference-order instance for bit comparison.
// Finite coordinates only (the grid searches never see another kind).
__device__ __forceinline__ Fit pivot_fit_finish(uint32_t m, double sx, double sy, double sz, double mxx, double mxy, double mxz, double myy, double myz, double mzz, bool* ill);
template <int KMAX, typename GetPoint>
__device__ __forceinline__ Fit plane_fit_pivot(uint32_t m, double px, double py, double pz, GetPoint&& get, bool* ill = nullptr) {
  double sx = 0, sy = 0, sz = 0, mxx = 0, mxy = 0, mxz = 0, myy = 0, myz = 0, mzz = 0;
#pragma unroll
  for (int t = 0; t < KMAX; ++t) {
    if ((uint32_t)t < m) {
      double x, y, z; get((uint32_t)t, x, y, z);
      const double ux = x - px, uy = y - py, uz = z - pz;
      sx += ux; sy += uy; sz += uz;
      mxx = __builtin_fma(ux, ux, mxx); mxy = __builtin_fma(ux, uy, mxy); mxz = __builtin_fma(ux, uz, mxz);
      myy = __builtin_fma(uy, uy, myy); myz = __builtin_fma(uy, uz, myz); mzz = __builtin_fma(uz, uz, mzz);
    }
  }
  return pivot_fit_finish(m, sx, sy, sz, mxx, mxy, mxz, myy, myz, mzz, ill);
}
// the fit behind the twelve pivot sums (shared by plane_fit_pivot and the box search's batched gather, knn_tile2_kernel)
__device__ __forceinline__ Fit pivot_fit_finish(uint32_t m, double sx, double sy, double sz, double mxx, double mxy, double mxz, double myy, double myz, double mzz, bool* ill) {
  if (m < 3) { Fit f{0, 0, 0, 0, 0}; if (ill) *ill = false; return f; }  // Err(...) :293-295 -> unwrap panic :471
  const double inv = 1.0 / (double)m;
  const double tx = sx * inv, ty = sy * inv, tz = sz * inv;  // centroid - pivot
#ifdef PST_FIT_NO_FAST  // (A/B builds: tools/build_variant_lib.sh)
  constexpr bool kFast = false;
#else
  constexpr bool kFast = true;
#endif
  return fit_from_covariance<kFast>(__builtin_fma(-sx, tx, mxx), __builtin_fma(-sx, ty, mxy), __builtin_fma(-sx, tz, mxz),
                             __builtin_fma(-sy, ty, myy), __builtin_fma(-sy, tz, myz), __builtin_fma(-sz, tz, mzz), ill);
}

// The fit as a WAVE-LEVEL operation (BASELINE.json configs[4]: "per-point 3x3 covariance wavefront reduction") for kernels that spend a whole
// wave or workgroup on ONE query (knn_select_kernel: the exact search of far points against all points): lane t fetches neighbour t -- ONE
// parallel gather instead of 2 m dependent loads by a single lane, which left 63 lanes idle --, and the centroid and the six moments are
// accumulated across the lanes with v_readlane IN POINT ORDER (every lane runs the same wave-uniform chain), i.e. in the reference's order
// of operations: the result is bit-identical to the lane-sequential plane_fit.  A shuffle TREE (six xor stages per sum) was built first and
// is what the name suggests; the differential fuzz rejected it: on ill-conditioned neighbourhoods (a 10^4 x 1 x 10^-2 slab at coordinates of
// 5 10^6) the reference's trigonometric cubic solver amplifies last-bit differences of the covariance into curvature differences far beyond
// 1e-9 -- 808 of 2049 in one case --, so the one place where a whole wave serves one query keeps the reference's order.  The box search
// (one query per LANE) does not use a cross-lane form at all: it is bound by its vector-instruction count, and a lane-parallel fit costs
// 9 sums x 6 stages x 3 instructions per query against 12 m / 64.
__device__ __forceinline__ double wave_readlane_f64(double v, uint32_t lane) {  // lane is wave-uniform
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, (int)lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)lane);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
// m <= 64 neighbours, neighbour t in lane t of the calling wave (all 64 lanes call; finite coordinates)
__device__ __forceinline__ Fit plane_fit_wave(uint32_t m, double x, double y, double z) {
  return plane_fit<0, true>(m, [&](uint32_t t, double& px, double& py, double& pz) __attribute__((always_inline)) {
    const uint32_t tu = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    px = wave_readlane_f64(x, tu); py = wave_readlane_f64(y, tu); pz = wave_readlane_f64(z, tu);
  });
}

// Where the results of a search kernel go.  Two forms:
//   * rec != null: ONE aligned 32-byte record {normal xyz, curvature} per point at the point's ORIGINAL index (a full-sector store);
//     split_results_kernel (normals.hip) then streams the records into the caller's outputs;
//   * rec == null (direct): the kernel writes the caller's outputs itself -- f64 arrays and / or the NORMAL (Vec3f32, Rust `as`
//     narrowing) and Curvature (F64) attributes at their strides.  12- and 8-byte random stores cost 40 + 32 bytes of HBM writes per
//     point against 32 for the record, but there is no second pass over 32 n bytes and no 32 n-byte scratch array; behind the
//     instruction-bound box search the extra write traffic is hidden.
struct RecOut {
  double* rec;              // [n][4] f64: nx, ny, nz, curvature, indexed by ORIGINAL point index; null = direct
  const uint32_t* sidx;     // sorted position -> original index
  long long* knn;           // [n][k] int64 (-1 = none), original indices, or null
  uint32_t* knn_u32;        // [n][k] uint32 (0xFFFFFFFF = none) or null
  int* error_count;         // neighbourhoods with fewer than 3 usable points
  // direct form (used when rec == null; any of them may be null / 0)
  double* normals_f64;      // [n][3]
  double* curvature_f64;    // [n]
  uint64_t normal_attr;     // device address of the NORMAL (Vec3f32) attribute of point 0
  uint64_t normal_stride;
  uint64_t curv_attr;       // device address of the Curvature (F64) attribute of point 0
  uint64_t curv_stride;
};

__device__ __forceinline__ void write_record(const RecOut& o, uint64_t orig, const Fit& f) {
  if (!f.ok) { atomicAdd(o.error_count, 1); return; }
  if (o.rec) {
    double* r = o.rec + 4 * orig;
    typedef double d2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<d2*>(r) = d2{f.nx, f.ny};
    *reinterpret_cast<d2*>(r + 2) = d2{f.nz, f.curvature};
    return;
  }
  // whole values in as few store instructions as possible: a scattered store costs the memory pipeline one pass per lane whatever its width
  struct __attribute__((packed, aligned(1))) F3 { float x, y, z; };
  struct __attribute__((packed, aligned(8))) D3 { double x, y, z; };
  if (o.normals_f64) *reinterpret_cast<D3*>(o.normals_f64 + 3 * orig) = D3{f.nx, f.ny, f.nz};
  if (o.curvature_f64) o.curvature_f64[orig] = f.curvature;
  if (o.normal_attr) {  // f64 -> f32 narrowing of the normal = Rust `as` (RNE, overflow -> inf)
    const F3 v{(float)f.nx, (float)f.ny, (float)f.nz};
    __builtin_memcpy(as_global(o.normal_attr) + orig * o.normal_stride, &v, sizeof(F3));  // one 12-byte store (global_store_dwordx3)
  }
  if (o.curv_attr) store_un<double>(as_global(o.curv_attr) + orig * o.curv_stride, f.curvature);
}
__device__ __forceinline__ void write_knn(const RecOut& o, uint64_t orig, uint32_t k, uint32_t t, uint32_t neighbour_orig) {  // kNoIndex = none
  if (o.knn) o.knn[orig * k + t] = neighbour_orig == kNoIndex ? -1ll : (long long)neighbour_orig;
  if (o.knn_u32) o.knn_u32[orig * k + t] = neighbour_orig;
}

__device__ __forceinline__ bool shell_done(const GridParams& g, double qx, double qy, double qz, int cx, int cy, int cz, int r, double kth) {
  double margin = __builtin_inf();
  const double qa[3] = {qx, qy, qz};
  const int ca[3] = {cx, cy, cz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int ra = a == 0 ? r * (int)g.rx : r;  // shell r spans r * rx fine cells along x: the same distance as r cells along y and z
    const double ha = grid_edge(g, a);
    if (ca[a] - ra > 0) margin = __builtin_fmin(margin, qa[a] - (g.org[a] + (double)(ca[a] - ra) * ha));
    if (ca[a] + ra < (int)g.dim[a] - 1) margin = __builtin_fmin(margin, (g.org[a] + (double)(ca[a] + ra + 1) * ha) - qa[a]);
  }
  if (margin == __builtin_inf()) return true;  // the cube covers the whole grid
  margin = margin * (g.rotated ? 1.0 - 1e-9 : 1.0 - 1e-12) - 1e-300;  // (rotated coordinates carry a rounding error of ~1e-13 of the cloud's size)
  return margin > 0.0 && kth <= margin * margin;
}

}  // namespace pstn
