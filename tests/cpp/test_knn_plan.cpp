// CPU unit test of the kNN call's planning code (pasture_amd/csrc/normals_plan.hpp): the decisions are pure functions of measured statistics,
// so they are fed the statistics the GPU measured for the cloud kinds of the differential fuzz (recorded with PST_KNN_DEBUG=1 on an MI355X:
// tools/fuzz_knn_sparse.py, bench.py --workload normals_knn16 / normals_knn16_sheet) and the chosen path is asserted.
// Built and run by tests/test_knn_plan.py: g++ -std=c++17 -Ipasture_amd/csrc tests/cpp/test_knn_plan.cpp
#include <cassert>
#include <cstdio>
#include <cstring>
#include <vector>

#include "normals_plan.hpp"

using namespace pstk;

static int g_fail = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); ++g_fail; } } while (0)

struct Cloud {
  const char* name;
  uint64_t n;
  uint32_t k;
  double mn[3], mx[3];   // the (trimmed / rotated) box the grid is laid over
  double occupancy;      // of the 32^3 coarse cells of that box
  double h_gate, d_gate; // quick scale estimate (0 = not run)
  // expectations
  bool fills, box_search;
  uint32_t rx;
  char kernel;
};

int main() {
  // no PST_KNN_* variable set: the defaults
  const KnnTuning t = KnnTuning{};
  CHECK(t.cell_budget == 20 && t.flush_at == 48 && t.direct_out && t.box_list && t.rounds && !t.forced_scale() && t.scratch_max == ((long long)16 << 30));

  const Cloud clouds[] = {
      // 10^8 uniform points in 1000 x 1000 x 100 (bench normals_knn16): full box, gate agrees with the volume
      {"uniform volume", 100000000ull, 16, {0, 0, 0}, {1000, 1000, 100}, 1.0, 1.8838, 2.9, true, true, 4, 'D'},
      // the LiDAR-like sheet after trimming (normals_knn16_sheet): 23 % of the coarse cells, dimension 2
      {"sheet", 100000000ull, 16, {0, 0, 39.3}, {1000, 1000, 60.5}, 0.226, 0.2825, 1.96, false, true, 2, 'G'},
      // 20 000 points in a cube: too few for the box search of a sparse cloud, but it fills its box
      {"small volume", 20000ull, 16, {0, 0, 0}, {1000, 1000, 100}, 0.99, 0.0, 3.0, true, true, 4, 'D'},
      // a surface of 30 000 points: occupancy 0.06, below 2^20 points: the global-memory search
      {"small surface", 30000ull, 16, {0, 0, -20}, {500, 500, 20}, 0.06, 6.1, 2.0, false, false, 2, 'G'},
      // a dense core inside a thin halo: every coarse cell occupied, but the measured radius is far below the volume's
      {"core with halo", 10000000ull, 16, {-50000, -50000, -50000}, {50000, 50000, 50000}, 1.0, 2.0, 3.0, false, true, 4, 'G'},
      // k = 40 on a filled box: the first form of the box kernel
      {"k = 40", 10000000ull, 40, {0, 0, 0}, {1000, 1000, 100}, 1.0, 4.5, 3.0, true, true, 4, '1'},
      // a diagonal flight strip of 10^8 points in its rotated frame (8000 x 300 x 20): thin, filled along its own axes
      {"strip, rotated frame", 100000000ull, 16, {-4000, -150, -10}, {4000, 150, 10}, 0.62, 0.31, 2.1, false, true, 2, 'G'},
  };
  for (const Cloud& c : clouds) {
    const BoxStats b = BoxStats::of(c.mn, c.mx);
    const double m_target = 1.75 * c.k, h_box = b.edge_for(m_target / kBallVolume, c.n);
    const bool fills = cloud_fills_box(c.occupancy, c.h_gate, h_box);
    if (fills != c.fills) { std::printf("%s: fills = %d, expected %d (h_box %g)\n", c.name, (int)fills, (int)c.fills, h_box); ++g_fail; }
    const bool tile = try_box_search(c.k, c.occupancy, c.n, t);
    if (tile != c.box_search) { std::printf("%s: box search = %d, expected %d\n", c.name, (int)tile, (int)c.box_search); ++g_fail; }
    const double h_est = fills ? 0.0 : c.h_gate;
    const uint32_t rx = fine_cells_per_h(h_est, c.d_gate, c.occupancy, t);
    if (rx != c.rx) { std::printf("%s: rx = %u, expected %u\n", c.name, rx, c.rx); ++g_fail; }
    const char kern = box_kernel_for(c.k, fills, t);
    if (kern != c.kernel) { std::printf("%s: kernel %c, expected %c\n", c.name, kern, c.kernel); ++g_fail; }
  }

  // the uniform bench cloud: the volume's cell edge is the measured one (1.8838), its directory fits the budget with rx = 4
  {
    const double mn[3] = {0, 0, 0}, mx[3] = {1000, 1000, 100};
    const BoxStats b = BoxStats::of(mn, mx);
    const double h = b.edge_for(28.0 / kBallVolume, 100000000ull);
    CHECK(std::fabs(h - 1.8838) < 1e-3);
    const uint64_t cells = (uint64_t)(std::floor(1000 / (h / 4)) + 1) * (uint64_t)(std::floor(1000 / h) + 1) * (uint64_t)(std::floor(100 / h) + 1);
    CHECK(cells == 60903576ull);  // dim 2124 x 531 x 54, as the GPU run prints
    CHECK(directory_fits(cells, directory_budget(100000000ull, t)));
    const ProbeFit pf = probe_fit(h, 4.5, 28.8, 28.0);  // the probe of that index: no re-grid
    CHECK(pf.dim > 2.6 && pf.dim < 2.75 && probe_accepts(0, h, pf.h_new, t));
    CHECK(!cloud_is_concentrated(1.8838, h));
  }
  // the sheet: 1.88e9 cells at rx = 2 fit 20 cells per point; a 9-fold density contrast asks for one re-grid
  {
    CHECK(directory_fits(1879740000ull, directory_budget(100000000ull, t)));
    CHECK(!directory_fits(1879740000ull * 2, directory_budget(100000000ull, t)));  // rx = 4 would not
    const ProbeFit pf = probe_fit(1.0, 9.0, 71.0, 28.0);
    CHECK(!probe_accepts(0, 1.0, pf.h_new, t) && probe_accepts(2, 1.0, pf.h_new, t));
    CHECK(cloud_is_concentrated(2.0, 100.0) && !cloud_is_concentrated(0.0, 100.0));
  }
  // trimmed box: 64 outliers of 10^7 points stretch the box 40-fold along every axis; the 0.05 % cut finds the core
  {
    constexpr uint32_t B = 256;
    std::vector<uint32_t> hist(3 * B, 0);
    for (int c = 0; c < 3; ++c) { hist[c * B + 0] = 11; hist[c * B + B - 1] = 10; for (uint32_t i = 125; i < 131; ++i) hist[c * B + i] = 1666663; }
    const double mn[3] = {-20000, -20000, -20000}, mx[3] = {20000, 20000, 20000}, spu[3] = {B / 40000.0, B / 40000.0, B / 40000.0};
    double tmn[3], tmx[3];
    const double shrink = trimmed_box<B>(hist.data(), mn, mx, spu, 0.0005, tmn, tmx);
    CHECK(shrink < 1e-4 && take_trimmed_box(0, shrink));
    CHECK(tmn[0] > -700 && tmn[0] < -600 && tmx[0] > 600 && tmx[0] < 700);  // slices 124 .. 131: the core plus one slice either side
    CHECK(!take_trimmed_box(0, 0.2) && take_trimmed_box(1, 0.5) && !take_trimmed_box(3, 0.01));
  }
  // principal axes
  {
    CHECK(consider_rotation(0.3, 1u << 20, t) && !consider_rotation(0.6, 1u << 20, t) && !consider_rotation(0.3, 1000, t));
    CHECK(axes_are_coordinate_axes(0.999) && !axes_are_coordinate_axes(0.77));
    CHECK(take_rotated_box(0.1, 1.0) && !take_rotated_box(0.5, 1.0));
  }
  // global-memory search: cell edges from the measured scale of a surface (dimension 2): a cell holds k / 12 * 3 points
  {
    const double mn[3] = {0, 0, 0}, mx[3] = {500, 500, 40};
    const BoxStats b = BoxStats::of(mn, mx);
    const FallbackEdges by_volume = fallback_edges(b, 30000, 16, 0.0, 3.0, 28.0, t), measured = fallback_edges(b, 30000, 16, 6.1, 2.0, 28.0, t);
    CHECK(by_volume.dense > 0 && by_volume.hash > by_volume.dense);
    CHECK(measured.dense < by_volume.dense && measured.hash > measured.dense);
    CHECK(dense_directory_ok(100000, 30000) && !dense_directory_ok(2000000, 30000));
  }
  // open queries: 64 outliers against 10^7 points are searched against all points; a million open queries get another level
  {
    CHECK(search_all_points(1, 64, 10000000ull, true));
    CHECK(!search_all_points(1, 1000000, 10000000ull, true));
    CHECK(search_all_points(9, 1000000, 10000000ull, true));
  }
  // rounds: 527 queries on 256 threads (8.2 of 12 wave slots) -> 21 cells become 17; 912 on 512 threads stay; 354 on 256 stay (10 < 3/4 of 18)
  {
    CHECK(box_length_for_whole_rounds(21, 527.0, 256) == 18 || box_length_for_whole_rounds(21, 527.0, 256) == 17);
    CHECK(box_length_for_whole_rounds(19, 912.0, 512) == 19);
    CHECK(box_length_for_whole_rounds(18, 354.0, 256) == 18);
  }
  // switches
  {
    KnnTuning f = t;
    f.variant = 'B';
    CHECK(box_kernel_for(16, true, f) == 'B' && box_kernel_for(24, true, f) == '1');
    f.rx = 1;
    CHECK(fine_cells_per_h(0.0, 3.0, 1.0, f) == 1);
    f.no_tile = true;
    CHECK(!try_box_search(16, 1.0, 100000000ull, f));
    f.no_tile = false; f.force_tile = true;
    CHECK(try_box_search(16, 0.06, 30000, f));
    f.cell = 2.5;
    CHECK(f.forced_scale() && probe_accepts(0, 1.0, 5.0, f));
  }
  std::printf(g_fail ? "knn plan: %d FAILED\n" : "knn plan: all checks passed\n", g_fail);
  return g_fail != 0;
}
