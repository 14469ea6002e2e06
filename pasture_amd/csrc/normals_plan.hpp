// Host-only planning code of the kNN call (normals.hip): the tuning switches, read ONCE, and the decisions that are pure functions of
// measured statistics -- kept free of HIP so that tests/cpp/test_knn_plan.cpp can compile it with g++ and assert on the chosen path.
// Reference for what is being computed: pasture-algorithms/src/normal_estimation.rs:79-130 (a kd-tree there; a uniform grid here, whose
// cell edge, box and frame are what these functions decide).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace pstk {

// Every PST_KNN_* switch ("Tuning switches").  Loaded at first use and again by pst_reload_tuning() -- never on the call path:
// std::getenv is not thread-safe against setenv, and the call path of two threads may run concurrently.
struct KnnTuning {
  double cell = 0.0;           // PST_KNN_CELL: forced cell edge (> 0)
  double per_cell = 0.0;       // PST_KNN_PER_CELL: forced points per cell (> 0)
  double tau_m = 0.0;          // PST_KNN_TAU_M: points the ball of radius h should hold (default 1.75 k)
  int rx = 0;                  // PST_KNN_RX: fine x cells per h (1..8)
  long cell_budget = 20;       // PST_KNN_CELL_BUDGET: dense-directory cells per point
  bool debug = false, trace = false;
  bool no_scale = false, no_trim = false, no_rotate = false, no_tile = false, force_tile = false;
  int dense = -1;              // PST_KNN_DENSE: 0 = never the dense directory in the global-memory search
  bool direct_out = true;      // PST_KNN_DIRECT=0: 32-byte records + split pass
  bool box_list = true;        // PST_KNN_BOX_LIST=0: one workgroup per box, empty ones included
  bool rounds = true;          // PST_KNN_ROUNDS=0: boxes are not shortened to whole rounds
  bool occupancy_all = false;  // PST_KNN_OCC_ALL=1: the occupancy / trimmed-box pass looks at every point, not at a 2^23-point thinning
  bool side_stream = true;     // PST_KNN_SIDE_STREAM=0: the permutation of the points runs on the caller's stream, before the directory is built
  char variant = '\0';         // PST_KNN_VAR: '1', 'B', 'D', 'G'
  bool sort_fallback = true;   // PST_KNN_SORT_FALLBACK=0: the exact fallback search takes its queries in the order the box kernel's workgroups finished
  int reorder_unroll = 2;      // PST_REORDER_UNROLL: points per lane in flight in the permutation kernel (1 / 2 / 4)
  int fit = -1;                // PST_KNN_FIT=seq|pivot|rows (rows: the cross-lane covariance, sixteen lanes per query -- a measured alternative, normals_tile.hip FIT 2): the box search's plane fit in the reference's order of operations (two passes) / in one pass about the query; default (-1): by cloud --
                               // one pass for clouds that fill their box, the reference's order for surfaces and strips (near-planar neighbourhoods: see normals_device.hpp)
  bool fit_guard = true;       // PST_KNN_FIT_GUARD=0: the one-pass fit never falls back to the reference's order for ill-conditioned neighbourhoods (tests that the guard is what keeps them in the window)
  unsigned tile[3] = {0, 0, 0};  // PST_KNN_TILE=bx,by,bz
  unsigned ablate = 0;         // PST_KNN_ABLATE (tuning only)
  unsigned flush_at = 48;      // PST_KNN_FLUSH_AT
  long long scratch_max = (long long)16 << 30;  // PST_SCRATCH_MAX_BYTES (default 16 GiB: a 10^8-point surface keeps 13.6 GB, 7.5 of them its directory)

  static KnnTuning from_env() {
    KnnTuning t;
    auto num = [](const char* n) -> double { const char* e = std::getenv(n); return e && *e ? std::atof(e) : 0.0; };
    auto set = [](const char* n) { return std::getenv(n) != nullptr; };
    auto off = [](const char* n) { const char* e = std::getenv(n); return e && *e && std::atoi(e) == 0; };
    t.cell = num("PST_KNN_CELL");
    t.per_cell = num("PST_KNN_PER_CELL");
    t.tau_m = num("PST_KNN_TAU_M");
    { const int v = (int)num("PST_KNN_RX"); t.rx = v >= 1 && v <= 8 ? v : 0; }
    { const long v = (long)num("PST_KNN_CELL_BUDGET"); t.cell_budget = v > 0 ? v : 20; }
    t.debug = set("PST_KNN_DEBUG"); t.trace = set("PST_KNN_TRACE");
    t.no_scale = set("PST_KNN_NO_SCALE"); t.no_trim = set("PST_KNN_NO_TRIM"); t.no_rotate = set("PST_KNN_NO_ROTATE");
    t.no_tile = set("PST_KNN_NO_TILE"); t.force_tile = set("PST_KNN_FORCE_TILE");
    if (const char* e = std::getenv("PST_KNN_DENSE")) t.dense = *e == '0' ? 0 : 1;
    t.direct_out = !off("PST_KNN_DIRECT"); t.box_list = !off("PST_KNN_BOX_LIST"); t.rounds = !off("PST_KNN_ROUNDS"); t.side_stream = !off("PST_KNN_SIDE_STREAM"); t.occupancy_all = num("PST_KNN_OCC_ALL") != 0.0;
    if (const char* e = std::getenv("PST_KNN_VAR")) t.variant = *e;
    if (const char* e = std::getenv("PST_KNN_FIT")) t.fit = e[0] == 's' ? 1 : (e[0] == 'p' ? 0 : (e[0] == 'r' ? 2 : -1));
    t.sort_fallback = !off("PST_KNN_SORT_FALLBACK");
    t.fit_guard = !off("PST_KNN_FIT_GUARD");
    if (const char* e = std::getenv("PST_REORDER_UNROLL")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) t.reorder_unroll = v; }
    if (const char* e = std::getenv("PST_KNN_TILE")) {
      unsigned x = 0, y = 0, z = 0;
      if (std::sscanf(e, "%u,%u,%u", &x, &y, &z) == 3 && x && y && z) { t.tile[0] = x; t.tile[1] = y; t.tile[2] = z; }
    }
    t.ablate = (unsigned)num("PST_KNN_ABLATE");
    if (const char* e = std::getenv("PST_KNN_FLUSH_AT")) t.flush_at = (unsigned)std::atoi(e);
    if (const char* e = std::getenv("PST_SCRATCH_MAX_BYTES")) { const long long v = std::atoll(e); if (v >= 0) t.scratch_max = v; }
    return t;
  }
  bool forced_scale() const { return cell > 0.0 || per_cell > 0.0; }
};
inline KnnTuning& knn_tuning_storage() { static KnnTuning t = KnnTuning::from_env(); return t; }
inline const KnnTuning& knn_tuning() { return knn_tuning_storage(); }
inline void knn_reload_tuning() { knn_tuning_storage() = KnnTuning::from_env(); }

// ---- decisions -----------------------------------------------------------------------------------------------------------------------

constexpr double kBallVolume = 4.18879020478639;  // 4/3 pi
constexpr int kShellCap = 6;       // shells a global-memory search walks before it hands a query to a coarser grid
constexpr uint32_t kCrowd = 4096;  // on coarser levels: a range longer than this is not walked by one lane

// the box a grid is laid over: extents, the largest one, volume over the non-flat axes
struct BoxStats {
  double ext[3], maxext, vol;
  int dims_used;
  static BoxStats of(const double mn[3], const double mx[3]) {
    BoxStats b{};
    for (int c = 0; c < 3; ++c) b.ext[c] = mx[c] - mn[c];
    b.maxext = std::fmax(b.ext[0], std::fmax(b.ext[1], b.ext[2]));
    if (!(b.maxext > 0.0)) b.maxext = 1.0;
    b.vol = 1.0; b.dims_used = 0;  // flat axes (extent < 1e-9 of the largest) are thickness-free: density is per area / per length then
    for (int c = 0; c < 3; ++c) if (b.ext[c] > b.maxext * 1e-9) { b.vol *= b.ext[c]; b.dims_used += 1; }
    return b;
  }
  // cell edge at which a cubic cell holds `per_cell` of the n points, by the box's volume
  double edge_for(double per_cell, uint64_t n) const { return dims_used ? std::pow(vol * per_cell / (double)n, 1.0 / dims_used) : maxext; }
};

// GATE: a quick scale estimate against the radius the box's volume predicts.  concentrated = most of the points sit in a small part of the box
inline bool cloud_is_concentrated(double h_gate, double h_box) { return h_gate > 0.0 && h_gate < h_box / 1.5; }
// the cloud is what its (trimmed) box says: the box is full and the measured scale agrees with the volume's within 30 %
inline bool cloud_fills_box(double occupancy, double h_gate, double h_box) {
  return occupancy >= 0.9 && !(h_gate > 0.0 && std::fabs(h_gate / h_box - 1.0) > 0.3);
}
// up to 4 million points the gate's subsample is thinned by at most 32: as good as the full estimate
inline bool gate_is_enough(double h_gate, uint64_t n) { return h_gate > 0.0 && n <= (32ull << 17); }

// TRIMMED BOX from point counts per slice of every axis (kAxisBins slices): the smallest slice ranges that hold all but `cut_frac` of the
// points at either end, one slice added on each side.  Returns the volume ratio trimmed / untrimmed.
template <uint32_t BINS>
inline double trimmed_box(const uint32_t* hist /* [3][BINS] */, const double mn[3], const double mx[3], const double slices_per_unit[3], double cut_frac,
                          double tmn[3], double tmx[3]) {
  double shrink = 1.0;
  for (int c = 0; c < 3; ++c) {
    tmn[c] = mn[c]; tmx[c] = mx[c];
    if (!(slices_per_unit[c] > 0)) continue;
    const uint32_t* hc = hist + c * BINS;
    uint64_t total = 0;
    for (uint32_t i = 0; i < BINS; ++i) total += hc[i];
    const uint64_t cut = (uint64_t)((double)total * cut_frac);
    uint32_t lo = 0, hi = BINS - 1;
    for (uint64_t acc = 0; lo < hi && acc + hc[lo] <= cut; ++lo) acc += hc[lo];
    for (uint64_t acc = 0; hi > lo && acc + hc[hi] <= cut; --hi) acc += hc[hi];
    lo = lo > 0 ? lo - 1 : 0; hi = hi + 1 < BINS ? hi + 1 : BINS - 1;
    tmn[c] = std::fmax(mn[c], mn[c] + (double)lo / slices_per_unit[c]);
    tmx[c] = std::fmin(mx[c], mn[c] + (double)(hi + 1) / slices_per_unit[c]);
    shrink *= (tmx[c] - tmn[c]) / (mx[c] - mn[c]);
  }
  return shrink;
}
// a trimmed box replaces the current one when it is at least 8 times smaller (first look) / 40 % smaller (second and third look)
inline bool take_trimmed_box(int pass, double shrink) { return pass < 3 && shrink <= (pass == 0 ? 0.125 : 0.6); }

// PRINCIPAL AXES: looked at for clouds that fill less than half of their box; taken when the box along them is at most 0.35 of the volume;
// not even measured when the axes are within 6 degrees of the coordinate axes (align = smallest of the rows' largest components)
inline bool consider_rotation(double occupancy, uint64_t n, const KnnTuning& t) { return occupancy < 0.5 && n >= (1u << 16) && !t.no_rotate; }
inline bool axes_are_coordinate_axes(double align) { return align >= 0.995; }
inline bool take_rotated_box(double v_rot, double v_now) { return v_rot <= 0.35 * v_now; }

// BOX SEARCH: which clouds try it, with how many fine x cells per h, and whether the directory fits its budget
inline bool try_box_search(uint32_t k, double occupancy, uint64_t n, const KnnTuning& t) {
  return k <= 64 && !t.no_tile && (occupancy >= 0.5 || n >= (1u << 20) || t.force_tile);
}
// 4 for clouds that fill their box; 2 for surfaces (the measured dimension where there is one: a sheet that fills a thin box is still a sheet)
inline uint32_t fine_cells_per_h(double h_est, double d_est, double occupancy, const KnnTuning& t) {
  if (t.rx) return (uint32_t)t.rx;
  return (h_est > 0.0 ? d_est < 2.5 : occupancy < 0.5) ? 2u : 4u;
}
inline uint64_t directory_budget(uint64_t n, const KnnTuning& t) { return std::max<uint64_t>((uint64_t)t.cell_budget * n, 1u << 20); }
inline bool directory_fits(uint64_t cells, uint64_t budget) { return cells <= budget && cells < 0xFFFFFFF0ull; }
// DENSITY PROBE of a built index: N(r) ~ r^D through (h/2, m_half) and (h, m_full); the radius that holds m_target points
struct ProbeFit { double dim, h_new; };
inline ProbeFit probe_fit(double h, double m_half, double m_full, double m_target) {
  const double D = std::fmin(3.0, std::fmax(1.0, std::log2(std::fmax(m_full, 1.0) / std::fmax(m_half, 1.0))));
  return ProbeFit{D, h * std::pow(m_target / std::fmax(m_full, 1.0), 1.0 / D)};
}
inline bool probe_accepts(int round, double h, double h_new, const KnnTuning& t) {
  return round == 2 || std::fabs(h_new / h - 1.0) <= 0.10 || t.forced_scale();
}

// GLOBAL-MEMORY SEARCH: cubic cells of ~k/12 points (dense directory) or ~k/3 (hash table), from the measured scale where there is one:
// a cubic cell of edge h holds about what a ball of radius c_D h does, and a cloud of dimension D occupies 3^D of the 27 cells of the
// first shell, so the cell's share is scaled by 3^(3-D)
struct FallbackEdges { double dense, hash; };
inline FallbackEdges fallback_edges(const BoxStats& b, uint64_t n, uint32_t k, double h_est, double d_est, double m_target, const KnnTuning& t) {
  FallbackEdges e{b.edge_for(t.per_cell > 0 ? t.per_cell : std::fmax(0.5, (double)k / 12.0), n), b.edge_for(t.per_cell > 0 ? t.per_cell : std::fmax(1.0, (double)k / 3.0), n)};
  if (h_est > 0.0 && !t.forced_scale()) {
    const double c_d = 0.44 + 0.06 * d_est, shells = std::pow(3.0, 3.0 - d_est);
    e.dense = h_est / c_d * std::pow(std::fmax(0.5, (double)k / 12.0) * shells / m_target, 1.0 / d_est);
    e.hash = h_est / c_d * std::pow(std::fmax(1.0, (double)k / 3.0) * shells / m_target, 1.0 / d_est);
  }
  return e;
}
inline bool dense_directory_ok(uint64_t cells, uint64_t n) { return cells <= std::max<uint64_t>(4 * n, 1u << 20) && cells < 0xFFFFFFF0ull; }

// OPEN QUERIES after a capped search: all points, or another (six times coarser) level?  The all-points search does ~2e12 pairs per
// second; an index costs ~2 ns per point with a dense directory and ~4.5 ns with the hash table, and may leave queries open.
inline bool search_all_points(int level, uint64_t n_open, uint64_t nf, bool up_dense) {
  const double cost_all = (double)n_open * (double)nf / 2e12, cost_level = (double)nf * (up_dense ? 2e-9 : 4.5e-9);
  return level > 8 || cost_all <= 2.0 * cost_level;
}

// ROUNDS: the queries of a box are handed to the workgroup's waves in chunks of 64, so a box of Q queries keeps its LDS for
// ceil(Q / threads) rounds.  If the typical box (mean + two standard deviations of a Poisson count) needs r rounds and fills less than
// 85 % of them, the box is shortened along x to what r - 1 rounds hold -- unless that costs more than a quarter of its length.
// Returns the new bx, or bx itself.
inline uint32_t box_length_for_whole_rounds(uint32_t bx, double queries_per_box, uint32_t threads) {
  const double nt = (double)threads, rounds = std::ceil((queries_per_box + 2.0 * std::sqrt(queries_per_box)) / nt);
  if (!(rounds >= 2.0) || queries_per_box / (rounds * nt) >= 0.85) return bx;
  const double target = (rounds - 1.0) * nt - 2.0 * std::sqrt((rounds - 1.0) * nt);
  const uint32_t bx2 = (uint32_t)std::floor((double)bx * target / queries_per_box);
  return bx2 >= 1 && 4 * bx2 >= 3 * bx && bx2 < bx ? bx2 : bx;
}

// which instance of the box kernel (normals_tile.hip): 'D' = 512 threads / 3000 staged points for clouds that fill their box, 'G' = 256 /
// 1536 for the others, the first form ('1') for k > 16
inline char box_kernel_for(uint32_t k, bool volume_like, const KnnTuning& t) {
  if (k > 16) return '1';
  switch (t.variant) { case '1': case 'B': case 'D': case 'G': return t.variant; default: return volume_like ? 'D' : 'G'; }
}

}  // namespace pstk
