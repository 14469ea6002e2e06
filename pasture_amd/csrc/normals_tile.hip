// K4 (dense-directory regime) — kNN search over an LDS-staged box of grid cells, gfx950.
//
// compute_normals (pasture-algorithms/src/normal_estimation.rs:79-130) needs, per point, the k nearest neighbours in ascending
// distance (:103-108) and a plane fit over them (:198-467).
//
// Grid (normals.hip): cell edge h along y and z = R0, the radius of the sphere expected to hold ~1.75 k points; along x the cells are rx
// (4) times finer.  Sorted points are row-major by cell, so the fine cells x0..x1 of one grid row are ONE contiguous range of sorted
// points, ordered by x at granularity h / rx.  tau0 = h^2 is an a-priori bound on the squared k-th distance: a query that finds k points
// inside tau0 needs only the 3 x 3 rows around its own row, and in each of them only the x interval the ball of that radius cuts out.
//
// A workgroup owns a BOX of bx * by * bz query cells.  It stages the points of the box plus its halo -- (by + 2)(bz + 2) row segments of
// bx + 2 (rx + 1) fine cells, each a coalesced copy -- into LDS ONCE, with a 16-bit local copy of the cell directory, and every query
// of the box finds its candidates there:
//   * no dependent global loads in the search (the first version of this kernel issued one per candidate and lane: 30x the algorithmic
//     HBM traffic and an L2 / HBM round trip per candidate);
//   * 9 row segments per query, each TRIMMED to the ball of the a-priori bound tau0 in f32 with upward slack (~97 candidates instead of the
//     ~170 of a 5 x 5 x 5 block of cubic cells); all nine ranges are worked out up front into a register table that a lane shifts down;
//   * four candidates per step: their 12 coordinate reads (one array per coordinate: neighbouring slots never share a bank) are
//     issued together, one LDS round trip per four distance tests;
//   * a candidate is only QUEUED (its 16-bit slot, per-lane LDS queue) when it beats tau0 and the lane's (k+1)-th best key as of the last
//     insertion round; the sorted insertion -- which a wave pays for whenever ANY lane inserts -- runs in rounds when queues fill,
//     software-pipelined, on keys that carry the slot in their low 11 bits: two f64 operations per list entry;
//   * no per-lane predication in the two inner loops (masked arithmetic and +inf insertions instead of execution-mask changes);
//   * the plane fit reads its neighbours from LDS; the result leaves as one aligned 32-byte record at the point's original index
//     (a full-sector store) and split_results_kernel streams the records into the caller's outputs.
//
// A query that does not find k points inside tau0 (sparse neighbourhoods), whose packed keys are ambiguous (see the end of the
// search), and every query of a box whose halo does not fit the LDS budget (locally dense clouds) is appended to a fallback list and
// searched by knn_grid_kernel (normals.hip) over global memory: the box kernel is the fast path, not the only path.
// f64 throughout, -ffp-contract=off; no MFMA (no contraction here).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"
#include "normals_device.hpp"
#include "normals_host.hpp"
#include "normals_plan.hpp"

using namespace pstn;

namespace {

// step / candidate statistics of the search (a build with -DPST_KNN_STATS prints them per launch; off: the counters cost registers)
// -DPST_KNN_MARKS (analysis builds only, tools/knn_phase_budget.py): comment lines in the assembly that delimit the phases of knn_tile2_kernel
#ifdef PST_KNN_MARKS
#define PST_KNN_MARK(name) asm volatile("; PSTMARK " name)
#else
#define PST_KNN_MARK(name)
#endif
// (knn_tile2_kernel's counters are added by lane ONE: a trailing `if (lane == 0) <global store>` at the chunk loop's latch, next to the hand-out's
//  `if (lane == 0) atomicAdd(&s_next, ...)` at its head, makes this toolchain emit a loop that hands the same chunks out again and again -- the
//  fallback list overflows and the kernel faults; found in round 5 with three probe builds, tools/README.md)
#ifdef PST_KNN_STATS
#define PST_KNN_STAT(...) __VA_ARGS__
#else
#define PST_KNN_STAT(...)
#endif

constexpr int kHalo = 1;          // halo rows on each side of a query row (the cell edge h is the a-priori bound on the k-th distance)
constexpr int kMaxRows = 144;     // halo rows per box: (by + 2)(bz + 2)
constexpr int kMaxQRows = 64;     // query rows per box: by * bz (one lane each in the prefix sum)
constexpr int kMaxRowCells = 31;  // halo cells per row: bx + 2 (rx + 1) (directory rows have one more entry, one lane each)
constexpr int kMaxDir = 3072;     // directory entries per box: rows * (cells per row + 1)
constexpr int kQueue = 16;        // queued candidates per lane
constexpr int kBatch = 4;         // candidates tested per lane and step (their 12 LDS reads are issued together)
constexpr int kSegs = 9;          // row segments per query: the 3 x 3 rows around its own, each trimmed to the ball along x

struct TileArgs {
  const double* sxyz;          // sorted positions
  const uint32_t* cell_start;  // dense directory
  GridParams g;
  uint32_t bx, by, bz;         // box size in cells
  uint32_t nbx, nby, n_boxes;  // boxes along x, along y, in total
  uint32_t k, nf;
  RecOut out;
  uint32_t* fb_list;           // queries left to the global-memory search ...
  uint32_t* fb_count;          // ... and how many
  double tau0;                 // a-priori bound on the squared k-th distance (+inf = none), see launch_knn_tile
  unsigned long long* dbg;     // -DPST_KNN_STATS builds only: [0] scan steps, [1] insertion steps, [2] query waves, [3] candidates tested, [4] queued
  double tau0_below;           // the largest double below tau0
  const uint32_t* box_list;    // the boxes to search (null: all n_boxes of them, box = workgroup id); n_boxes = its length then
  const uint32_t* n_boxes_dev; // stream-ordered replay of a plan: the list's length lives on the device (n_boxes = the capacity the grid was sized for)
  uint32_t flush_at;           // PST_KNN_FLUSH_AT, default 48
  uint32_t fit_guard;          // 1 (default): a lane whose neighbourhood is ill-conditioned repeats the one-pass fit in the reference's order (PST_KNN_FIT_GUARD=0: never)
  uint32_t ablate;             // tuning only (PST_KNN_ABLATE): 1 = no insertion, 2 = no plane fit, 4 = no scan, 8 = nothing queued, 16 = no copy, 32 = no records, 64 = empty kernel
};

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t& total) {
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  total = (uint32_t)__shfl((int)inc, 63, 64);
  return inc - v;
}

// ---- k-best list with the LDS slot of the candidate packed into the low bits of its f64 squared distance ---------------------------
// One register pair per entry and two f64 operations per entry and insertion (v_max_f64 + v_min_f64) instead of five with a separate
// index array: the sorted insertion is what a query wave spends most of its VALU time on.  11 bits of slot (CAP <= 2048) leave 41
// mantissa bits; the kernel verifies afterwards that the dropped bits could not have changed the order (see the end of the search).
constexpr uint32_t kSlotBits = 11, kSlotMask = (1u << kSlotBits) - 1u;
__device__ __forceinline__ double pack_key(double d, uint32_t slot) {
  // a squared distance that overflowed to +inf becomes a NaN pattern here and then fails every `<`: the candidate is ignored, the
  // query cannot reach k neighbours inside tau0 through it, and the completion test sends it to the exact search
  uint64_t b = __builtin_bit_cast(uint64_t, d);
  b = (b & ~(uint64_t)kSlotMask) | slot;
  return __builtin_bit_cast(double, b);
}
__device__ __forceinline__ uint32_t key_slot(double key) { return (uint32_t)__builtin_bit_cast(uint64_t, key) & kSlotMask; }
__device__ __forceinline__ bool same_masked(double a, double b) {
  return ((__builtin_bit_cast(uint64_t, a) ^ __builtin_bit_cast(uint64_t, b)) >> kSlotBits) == 0;
}
__device__ __forceinline__ double key_upper(double key) {  // >= the exact distance the key was made from
  return __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, key) | (uint64_t)kSlotMask);
}
template <int K>
struct KBestPacked {
  double key[K + 1];  // ascending; one entry more than needed: the (k+1)-th key decides whether the k-th is unambiguous
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t <= K; ++t) key[t] = __builtin_inf();
  }
  __device__ __forceinline__ void insert(double x) {  // new[t] = min(old[t], max(old[t-1], x))
    if (!(x < key[K])) return;
    // v_max_f64 / v_min_f64 written out: __builtin_fmin / fmax make hipcc canonicalise every operand first (one extra v_max_f64 per
    // entry; keys come out of integer bit operations, which it cannot prove to be canonical) -- half again as many f64 operations.
    // From the top down, so that entry t-1 still holds its old value when entry t is rewritten: in place, no register copies.
#pragma unroll
    for (int t = K; t >= 1; --t) {
      double hi;
      asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(key[t - 1]), "v"(x));
      asm("v_min_f64 %0, %1, %2" : "=v"(key[t]) : "v"(key[t]), "v"(hi));
    }
    asm("v_min_f64 %0, %1, %2" : "=v"(key[0]) : "v"(key[0]), "v"(x));
  }
  __device__ __forceinline__ void insert_always(double x) {  // the same without the early exit (x = +inf is a no-op): no execution-mask traffic
#pragma unroll
    for (int t = K; t >= 1; --t) {
      double hi;
      asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(key[t - 1]), "v"(x));
      asm("v_min_f64 %0, %1, %2" : "=v"(key[t]) : "v"(key[t]), "v"(hi));
    }
    asm("v_min_f64 %0, %1, %2" : "=v"(key[0]) : "v"(key[0]), "v"(x));
  }
  __device__ __forceinline__ double kth(uint32_t k) const {  // key[k-1] without dynamic register indexing
    double v = key[K - 1];
#pragma unroll
    for (int t = 0; t < K; ++t) v = (uint32_t)t == k - 1 ? key[t] : v;
    return v;
  }
};

// The halo of a box is addressed WITHOUT clipping: rows and cells outside the grid exist in the local directory as empty ranges, so
// the search loop has no boundary tests: halo row r = (z - (Z0-2)) * HY + (y - (Y0-2)), halo cell c = x - (X0-2), and the candidates of
// a query in halo cell (lx, ly, lz) in the row (dy, dz) away are the LDS slots [ldir[B + D], ldir[B + D + 5]) with
// B = (lz * HY + ly) * NC1 + lx - XH (per lane) and D = (dz * HY + dy) * NC1 (per segment: a compile-time pair (dy, dz)).
template <int K, int THREADS, int CAP, bool WITH_KNN>
__global__ __launch_bounds__(THREADS, (K <= 16 ? 3 : 2) * THREADS / 256) void knn_tile_kernel(const TileArgs a) {
  // staged points, one array per coordinate (8-byte stride: neighbouring slots never share a bank); + kBatch: a batch may read past a range
  constexpr int CS = CAP + kBatch;
  __shared__ double P3[3 * CS];  // x of slot s = P3[s], y = P3[CS + s], z = P3[2 * CS + s]
  __shared__ uint16_t ldir[kMaxDir];            // ldir[r * NC1 + c] = first LDS slot of halo cell c of halo row r
  __shared__ uint32_t g0[kMaxRows];             // first sorted point of every halo row
  __shared__ uint32_t rbase[kMaxRows + 1];      // first LDS slot of every halo row (exclusive prefix sum of the row lengths)
  __shared__ uint32_t qpre[kMaxQRows + 1];      // exclusive prefix sum of the query counts of the query rows
  __shared__ uint16_t qbuf[kQueue * THREADS];
  __shared__ uint32_t s_next;
  constexpr int NW = THREADS / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t wg = xcd_block_id();
  if (wg >= a.n_boxes || (a.n_boxes_dev && wg >= *a.n_boxes_dev)) return;
  if (a.ablate & 64u) return;
  const uint32_t box = a.box_list ? a.box_list[wg] : wg;
  const GridParams& g = a.g;
  const int dim0 = (int)g.dim[0], dim1 = (int)g.dim[1], dim2 = (int)g.dim[2];
  const int bxi = (int)(box % a.nbx), byi = (int)((box / a.nbx) % a.nby), bzi = (int)(box / (a.nbx * a.nby));
  const int X0 = bxi * (int)a.bx, Y0 = byi * (int)a.by, Z0 = bzi * (int)a.bz;  // first query cell
  const int XH = (int)g.rx + 1;  // halo cells along x: the ball of radius < h reaches at most rx fine cells + the query's own partial cell
  const int HX = (int)a.bx + 2 * XH, HY = (int)a.by + 2 * kHalo, HZ = (int)a.bz + 2 * kHalo;
  const int NC1 = HX + 1, NR = HY * HZ, nqy = (int)a.by, nqr = (int)(a.by * a.bz);
  auto row_in_grid = [&](int r, uint64_t& cell0) {
    const int y = Y0 - kHalo + r % HY, z = Z0 - kHalo + r / HY;
    cell0 = ((uint64_t)z * (uint64_t)dim1 + (uint64_t)y) * (uint64_t)dim0;
    return y >= 0 && y < dim1 && z >= 0 && z < dim2;
  };
  auto clamp_x = [&](int x) { return (uint32_t)(x < 0 ? 0 : (x > dim0 ? dim0 : x)); };

  // ---- A: the directory entries of every halo row (half a wave per row, one lane per cell boundary): ONE global round trip; the raw
  //         32-bit entries are parked in the coordinate arrays, which are not in use yet --------------------------------------------
  uint32_t* raw = reinterpret_cast<uint32_t*>(P3);  // [NR][32]
  static_assert(sizeof(P3) >= kMaxRows * 32 * 4, "raw directory does not fit the coordinate arrays");
  for (int r = (int)(tid >> 5); r < NR; r += THREADS / 32) {
    const int c = (int)(tid & 31u);
    if (c < NC1) {
      uint64_t c0;
      raw[r * 32 + c] = row_in_grid(r, c0) ? a.cell_start[c0 + clamp_x(X0 - XH + c)] : 0u;
    }
  }
  if (tid == 0) s_next = 0;
#pragma unroll
  for (int i = 0; i < kQueue; ++i) qbuf[i * THREADS + tid] = 0;  // stale queue entries are read (never used): they must be valid slots
  __syncthreads();  // (1) raw directory in LDS
  // Every wave works out the row prefix sums for itself (NR <= 144 = 3 rows per lane): the total decides the path without another
  // barrier, and the wave can request its share of the points before the local directory is written.  Wave 0 publishes the sums.
  auto qrow_halo = [&](int qr) { return (kHalo + qr / nqy) * HY + kHalo + qr % nqy; };
  uint32_t total;
  {
    uint32_t v[3], sum = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int r = 3 * (int)lane + u; v[u] = r < NR ? raw[r * 32 + HX] - raw[r * 32] : 0u; sum += v[u]; }
    uint32_t run = wave_excl_scan(sum, lane, total);
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int r = 3 * (int)lane + u; if (r < NR) { rbase[r] = run; g0[r] = raw[r * 32]; } run += v[u]; }
      if (lane == 0) rbase[NR] = total;
      // D: queries of the box = the points of its bx cells in each of its by * bz rows
      uint32_t cnt = 0;
      if ((int)lane < nqr) { const int r = qrow_halo((int)lane); cnt = raw[r * 32 + XH + (int)a.bx] - raw[r * 32 + XH]; }
      uint32_t tot;
      const uint32_t ex = wave_excl_scan(cnt, lane, tot);
      if ((int)lane < nqr) qpre[lane] = ex;
      if (lane == 0) qpre[nqr] = tot;
    }
  }
  if (total == 0) return;
  if (total > (uint32_t)CAP) {
    // the halo does not fit: every query of the box goes to the global-memory search
    for (int qr = (int)wave; qr < nqr; qr += NW) {
      const int r = qrow_halo(qr);
      const uint32_t s0 = raw[r * 32 + XH], s1 = raw[r * 32 + XH + (int)a.bx];
      for (uint32_t j = s0 + lane; j < s1; j += 64) a.fb_list[atomicAdd(a.fb_count, 1u)] = j;
    }
    return;
  }
  // ---- C: coalesced copy of the row segments, xyz de-interleaved; kRows rows x kChunks 64-double chunks in flight per wave.  The loads
  //         of the first pass (all rows of a typical box: NW * kRows = 32) are requested HERE, off the raw directory, and land while the
  //         local directory is written (16-byte loads were measured slower: 9.0 against 7.0 ms for staging + output alone) -------------
  constexpr int kRows = 8, kChunks = 4;
  double cv[kRows][kChunks];
  uint32_t len3[kRows];
  const bool copy_on = !(a.ablate & 16u);
#pragma unroll
  for (int u = 0; u < kRows; ++u) {
    const int r = (int)wave * kRows + u;
    const uint32_t first = r < NR ? raw[r * 32] : 0u;
    len3[u] = r < NR && copy_on ? 3u * (raw[r * 32 + HX] - first) : 0u;
    const double* src = a.sxyz + 3ull * first;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) cv[u][c] = lane + 64u * c < len3[u] ? src[lane + 64u * c] : 0.0;
  }
  __syncthreads();  // (2) row sums published
  // ---- B: local 16-bit directory ----------------------------------------------------------------------------------------------------
  for (int r = (int)(tid >> 5); r < NR; r += THREADS / 32) {
    const int c = (int)(tid & 31u);
    if (c < NC1) ldir[r * NC1 + c] = (uint16_t)(raw[r * 32 + c] - g0[r] + rbase[r]);
  }
  __syncthreads();  // (3) the raw directory is dead: the coordinate arrays can be filled
  {
    auto put = [&](uint32_t base, uint32_t e, double v) __attribute__((always_inline)) {
      const uint32_t pt = e / 3u, c = e - 3u * pt;
      P3[c * CS + base + pt] = v;
    };
    auto store_rows = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int r = r0 + u;
        if (r >= NR) continue;
        const uint32_t base = rbase[r];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) if (lane + 64u * c < len3[u]) put(base, lane + 64u * c, cv[u][c]);
        if (len3[u] > 64u * kChunks) {
          const double* src = a.sxyz + 3ull * g0[r];
          for (uint32_t e = lane + 64u * kChunks; e < len3[u]; e += 64) put(base, e, src[e]);
        }
      }
    };
    store_rows((int)wave * kRows);
    for (int r0 = (NW + (int)wave) * kRows; r0 < NR && copy_on; r0 += NW * kRows) {  // boxes with more than NW * kRows rows
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int r = r0 + u;
        len3[u] = r < NR ? 3u * (rbase[r + 1] - rbase[r]) : 0u;
        const double* src = a.sxyz + 3ull * (r < NR ? g0[r] : 0u);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) cv[u][c] = lane + 64u * c < len3[u] ? src[lane + 64u * c] : 0.0;
      }
      store_rows(r0);
    }
  }
  __syncthreads();  // (4) points staged
  const uint32_t Q = qpre[nqr];
  const uint32_t m = a.nf < a.k ? a.nf : a.k;

  for (;;) {
    uint32_t c0 = 0;
    if (lane == 0) c0 = atomicAdd(&s_next, 64u);
    c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
    if (c0 >= Q) break;
    const uint32_t q = c0 + lane;
    const bool active = q < Q;
    PST_KNN_STAT(if (lane == 0) atomicAdd(a.dbg + 2, 1ull);)
    int qr = 0;
    {
      int lo = 0, hi = nqr;  // largest qr with qpre[qr] <= q
      const uint32_t qq = active ? q : 0u;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qpre[mid] <= qq) lo = mid; else hi = mid; }
      qr = lo;
    }
    const int hr = qrow_halo(qr);
    const uint32_t slot = active ? (uint32_t)ldir[hr * NC1 + XH] + (q - qpre[qr]) : 0u;
    const double qx = P3[slot], qy = P3[CS + slot], qz = P3[2 * CS + slot];
    const uint32_t j = g0[hr] + (slot - rbase[hr]);  // index among the sorted points
    const uint32_t orig = active ? a.out.sidx[j] : 0u;  // requested now: the round trip hides behind the search
    // the row (y, z) is the query row; the cell along x comes from the coordinate (same arithmetic as keys_kernel)
    double qu, qv, qw;  // the query in the grid's frame (cells, trims); distances use (qx, qy, qz)
    grid_frame(g, qx, qy, qz, qu, qv, qw);
    const int cx = (int)cell_coord(qu, g.org[0], g.inv_hx, g.dim[0]), cy = Y0 + qr % nqy, cz = Z0 + qr / nqy;
    const int B = hr * NC1 + (active ? cx - X0 + XH : XH);  // directory entry of the query's own cell (idle lanes: any valid one)
    // A query OUTSIDE the grid's box (the box may be a trimmed one: normals.hip) was clamped into a boundary cell.  The trims below bound the
    // distance to a row of cells by the distance to its slab -- true for the points of the row, and for points clamped INTO it as seen from
    // inside the box (they lie beyond their cell, never nearer), but not as seen from outside: a query half a cell beyond a face may have a
    // neighbour right next to it that was clamped into the same boundary row, and the slab is half a cell away.  Such a query goes to the
    // global-memory search, which uses no slab bound.  (Exactly ON an upper face -- the box's own last points -- is inside: slab distance 0.)
    auto beyond = [](double v, double org, double inv, uint32_t dim) { const double t = (v - org) * inv; return t < 0.0 || t > (double)dim; };
    const bool outside = beyond(qu, g.org[0], g.inv_hx, g.dim[0]) || beyond(qv, g.org[1], g.inv_h, g.dim[1]) || beyond(qw, g.org[2], g.inv_h, g.dim[2]);

    KBestPacked<K> best;
    best.init();
    uint32_t qn = 0;  // queued candidates of this lane
    // A candidate is queued when its squared distance is below the a-priori bound AND not above the (k+1)-th best key as of the last
    // insertion round with its slot bits set (key_upper: every distance whose packed key could still be smaller).  The plain f64
    // distance is compared: packing happens in the insertion round only.
    double lim = a.tau0_below;
    auto key_of = [&](double x, double y, double z, uint32_t p) __attribute__((always_inline)) {
      const double dx = x - qx, dy = y - qy, dz = z - qz;
      return pack_key(dx * dx + dy * dy + dz * dz, p);
    };
    // Batched insertion, software-pipelined: the slot of entry i+2 and the coordinates of entry i+1 are requested before entry i is
    // inserted, so the LDS round trips hide behind the insertion's VALU work.
    auto flush = [&]() __attribute__((always_inline)) {
      // nothing here is predicated per lane: a lane without an entry reads a stale (valid) slot and inserts +inf, which changes nothing
      uint32_t p0 = qbuf[tid];
      uint32_t p1 = qbuf[THREADS + tid];
      double x0 = P3[p0], y0 = P3[CS + p0], z0 = P3[2 * CS + p0];
#pragma unroll 2
      for (uint32_t i = 0; i < (uint32_t)kQueue; ++i) {
        const bool has = i < qn;
        if (!__any(has)) break;
        PST_KNN_STAT(if (lane == 0) atomicAdd(a.dbg + 1, 1ull);)
        const uint32_t p2 = qbuf[(i + 2 < (uint32_t)kQueue ? i + 2 : (uint32_t)kQueue - 1u) * THREADS + tid];
        const double x1 = P3[p1], y1 = P3[CS + p1], z1 = P3[2 * CS + p1];
        const double key = key_of(x0, y0, z0, p0);
        if (!(a.ablate & 1u)) best.insert_always(has ? key : __builtin_inf());
        p0 = p1; p1 = p2; x0 = x1; y0 = y1; z0 = z1;
      }
      qn = 0;
      const double ku = best.key[K] < __builtin_inf() ? key_upper(best.key[K]) : lim;  // (+inf with the slot bits set would be a NaN)
      asm("v_min_f64 %0, %1, %2" : "=v"(lim) : "v"(ku), "v"(lim));
    };
    // Scan.  Every lane walks its own 9 row segments (the 3 x 3 rows around its cell), own row first, then the four rows that share a
    // face with it, then the diagonals.  BALL TRIMMING per segment, in units of h and f32 with upward slack: a row whose (y, z) slab is
    // farther from the query than the a-priori bound tau0 < h^2 is dropped, the others are cut to the fine x cells that intersect the
    // ball.  All nine ranges are worked out up front, by all lanes together with compile-time row offsets (their 18 directory reads in
    // one round trip), and kept as a register table that a lane shifts down when a range is used up: ~13 instructions per advance.
    // (Trimming each segment when the lane reached it, with the running k-th distance as the bound, tested 90.7 instead of 96.6
    // candidates per query, but the wave ran that ~55-instruction advance for some lane in nearly every scan step.)
    uint32_t ent[kSegs];
    {
      const float fx = (float)((qu - g.org[0]) * g.inv_hx - (double)cx), fy = (float)((qv - g.org[1]) * g.inv_h - (double)cy),
                  fz = (float)((qw - g.org[2]) * g.inv_h - (double)cz);
      const float bound = (float)a.tau0 * ((float)(g.inv_h * g.inv_h) * 1.00001f), rxf = (float)g.rx * 1.00001f, xh = (float)XH;
      const bool on = active && !outside && !(a.ablate & 4u);
#pragma unroll
      for (int s = 0; s < kSegs; ++s) {
        constexpr int kDy[kSegs] = {0, -1, 1, 0, 0, -1, 1, -1, 1}, kDz[kSegs] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        const int dy = kDy[s], dz = kDz[s];
        // distance (in units of h) from the query to the slab [dy, dy + 1) x [dz, dz + 1) of rows, relative to its own row
        const float ty = (float)dy - fy, tz = (float)dz - fz;
        const float gy = fmaxf(0.0f, fmaxf(ty, -ty - 1.0f)), gz = fmaxf(0.0f, fmaxf(tz, -tz - 1.0f));
        const float r2 = bound - (gy * gy + gz * gz);
        // half-width of the ball in this row, in fine x cells: the raw v_sqrt_f32 (1 ulp; sqrtf expands into a 20-instruction
        // correctly-rounded sequence) under the 1e-5 relative slack of rxf and an absolute one for arguments near zero
        const float ext = __builtin_amdgcn_sqrtf(fmaxf(r2, 0.0f)) * rxf + 1e-6f;
        const float lo_f = fmaxf(floorf(fx - ext), -xh), hi_f = fminf(floorf(fx + ext), xh);
        const int Bd = B + (dz * HY + dy) * NC1;  // (dz * HY + dy) * NC1 is the same for the whole workgroup
        const uint32_t s0 = ldir[Bd + (int)lo_f], s1 = ldir[Bd + (int)hi_f + 1];  // fine cells [lo, hi] of the row, relative to the query's cell
        ent[s] = (on && r2 > 0.0f) ? (s0 | (s1 << 16)) : 0u;
      }
    }
    uint32_t left = kSegs;
    uint32_t p = 0, pe = 0;
    for (;;) {
      if (p >= pe && left != 0u) {  // ONE shift per step: a lane that drew an empty range idles this step and draws again in the next
        const uint32_t e = ent[0];
#pragma unroll
        for (int s = 0; s + 1 < kSegs; ++s) ent[s] = ent[s + 1];
        left -= 1;
        p = e & 0xFFFFu; pe = e >> 16;
      }
      // A lane whose queue could overflow WAITS (it tests nothing this step) instead of forcing the whole wave into a half-empty
      // insertion round: the queues are emptied only when (almost) no lane can go on scanning, i.e. when nearly every lane holds a full
      // queue or has finished (measured at 26 queued candidates per query: 71 insertion steps per query wave with flush-on-first-full).
      const bool room = qn <= (uint32_t)(kQueue - kBatch);
      const uint32_t rem = (p < pe && room) ? pe - p : 0u;
      // With lanes waiting, the wave scans on only while more than flush_at lanes can still scan: waiting for the last stragglers costs
      // more scan steps than their few queue entries save in the insertion round (same-box sweep of flush_at, box search per 10^8
      // points: 0: 45.2 ms, 4: 43.5, 8: 43.0, 16: 42.5, 32: 42.6, 48: 43.5, 56: 44.5).
      const uint64_t can = __ballot(rem != 0u || (p >= pe && left != 0u)), waiting = __ballot(p < pe && !room);
      if (can != 0 && !(waiting != 0 && (uint32_t)__builtin_popcountll(can) <= a.flush_at)) {
        PST_KNN_STAT(if (lane == 0) atomicAdd(a.dbg, 1ull); atomicAdd(a.dbg + 3, (unsigned long long)(rem < (uint32_t)kBatch ? rem : (uint32_t)kBatch));)
        if (rem != 0u) {  // ONE predicate per step; inside, nothing branches and the execution mask stays put
          double cx_[kBatch], cy_[kBatch], cz_[kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; ++u) { cx_[u] = P3[p + u]; cy_[u] = P3[CS + p + u]; cz_[u] = P3[2 * CS + p + u]; }  // (past the range: read, then ignored)
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            const double dx = cx_[u] - qx, dy = cy_[u] - qy, dz = cz_[u] - qz;
            const double d2 = dx * dx + dy * dy + dz * dz;
            // the slot is written in any case (a scanning lane has room for kBatch entries) and kept if the candidate passes
            qbuf[qn * THREADS + tid] = (uint16_t)(p + (uint32_t)u);
            const bool pass = ((uint32_t)u < rem) & (d2 <= lim) & !(a.ablate & 8u);
            PST_KNN_STAT(if (pass) atomicAdd(a.dbg + 4, 1ull);)
            qn += pass ? 1u : 0u;
          }
        }
        p += rem < (uint32_t)kBatch ? rem : (uint32_t)kBatch;
      } else {
        if (__any(qn != 0u)) flush();  // ONE inlined copy of the insertion code
        if (!__any(p < pe || left != 0u)) break;
      }
    }
    // Packed keys order candidates by (distance with its low 11 bits dropped, slot).  That IS the exact ascending-distance order, ties
    // broken by slot, unless two of the best k+1 keys agree in every kept bit: then the pair is compared exactly -- an exact tie is
    // already in its final order; a near-tie inside the list, or any agreement across the k / k+1 boundary (a candidate that was
    // dropped might belong in front), sends the query to the exact search.
    bool exact = true;
    {
      const uint32_t kk = a.k;
#pragma unroll
      for (int t = 0; t < K; ++t) {
        if ((uint32_t)t < kk && same_masked(best.key[t], best.key[t + 1])) {
          if ((uint32_t)t + 1 == kk) exact = false;
          else {
            const uint32_t s0 = key_slot(best.key[t]), s1 = key_slot(best.key[t + 1]);
            const double dx0 = P3[s0] - qx, dy0 = P3[CS + s0] - qy, dz0 = P3[2 * CS + s0] - qz, dx1 = P3[s1] - qx, dy1 = P3[CS + s1] - qy, dz1 = P3[2 * CS + s1] - qz;
            if (dx0 * dx0 + dy0 * dy0 + dz0 * dz0 != dx1 * dx1 + dy1 * dy1 + dz1 * dz1) exact = false;
          }
        }
      }
    }
    // k candidates inside tau0 < h^2: everything at most that far from the query lies in the 3 x 3 rows (each at least h deep beyond
    // the query's cell) and inside the trims, which were cut with bounds that only shrank afterwards -- the list is complete
    const bool done = a.ablate ? true : !outside && exact && best.kth(a.k) < a.tau0;
    if (active && !done) a.fb_list[atomicAdd(a.fb_count, 1u)] = j;
    if (active && done && !(a.ablate & 32u)) {
      if constexpr (WITH_KNN) {
        for (uint32_t t = 0; t < a.k; ++t) {
          uint32_t v = kNoIndex;
          uint32_t pl = 0;
#pragma unroll
          for (int u = 0; u < K; ++u) if ((uint32_t)u == t) pl = key_slot(best.key[u]);
          if (t < m) {
            int lo = 0, hi = NR;  // halo row of LDS slot pl: the last row that starts at or before it
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rbase[mid] <= pl) lo = mid; else hi = mid; }
            v = a.out.sidx[g0[lo] + (pl - rbase[lo])];
          }
          write_knn(a.out, orig, a.k, t, v);
        }
      }
      Fit f{0, 0, 0, 0, 1};
      if (!(a.ablate & 2u)) f = plane_fit<K, true>(m, [&](uint32_t t, double& x, double& y, double& z) __attribute__((always_inline)) {
        uint32_t pl = 0;
#pragma unroll
        for (int u = 0; u < K; ++u) if ((uint32_t)u == t) pl = key_slot(best.key[u]);
        x = P3[pl]; y = P3[CS + pl]; z = P3[2 * CS + pl];
      });
      write_record(a.out, orig, f);
    }
  }
}

// =====================================================================================================================================
// Box search, second form (round 3): the SCAN and the k-best list run in f32 on box-relative coordinates; f64 only proves the result.
//
// The first form above spends ~24 vector instructions per candidate: f64 differences and products for every one of the ~97 candidates of
// a query, and two f64 operations per list entry for every one of the ~27 that enter the list.  Here
//   * every staged point also gets a BOX-RELATIVE f32 copy r = (float)((x - c) * s): c = the box's centre, s chosen so that the a-priori
//     bound tau0 on the squared k-th distance maps to 2^21 (coordinates in units of h / 1448).  The scan tests candidates in PACKED f32
//     (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two candidates per instruction) against a limit with upward slack;
//   * list keys are 32-bit: the squared f32 distance truncated to a 21-bit integer (v_cvt_u32_f32) above the 11-bit LDS slot.  Sorted
//     insertion is ONE v_med3_u32 per entry -- new[t] = med3(old[t-1], old[t], x) is the clamp of x into [old[t-1], old[t]] -- instead of
//     v_max_f64 + v_min_f64, and the list takes K + 1 registers instead of 2 (K + 1);
//   * candidates past the end of a row segment (a step tests four) are real staged points: they are tested like the others instead of
//     being masked out (the slots behind the last staged point hold a far-away sentinel);
//   * the EXACT f64 squared distances (the same sequence of operations as the reference) of the k best are computed once, at the end:
//       - they must ascend strictly in list order, ties by slot (the order the first form produces); a candidate that was tested twice
//         (an overrun into another segment of the same query) shows up as two equal neighbours and fails this test;
//       - completeness: every candidate that is NOT among the first k entries has a key >= key[k], so its f32 distance is at least
//         fix(key[k]) and its exact distance at least fix(key[k]) - eps (eps bounds |f32 - exact| for points of the box, see
//         launch_knn_tile2); the k-th exact distance must lie below that, and below tau0 (3 x 3 rows cover that radius);
//     a query that fails either test joins the fallback list like the sparse ones (measured: ~0.1 % of a uniform cloud).
// The result is therefore the same neighbour list in the same order as the first form's, or the query is searched by the exact kernel.
// P3LDS = false: the f64 coordinates are not staged at all; the end phase reads them from the sorted array in global memory (L2: the box
// was just copied from there) through a slot -> sorted-index map, which leaves 16 instead of 36 bytes of LDS per staged point.
typedef float f2v __attribute__((ext_vector_type(2)));

template <int K>
struct KBestU32 {
  uint32_t key[K + 1];  // ascending; 0xFFFFFFFF = empty
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t <= K; ++t) key[t] = 0xFFFFFFFFu;
  }
  __device__ __forceinline__ void insert(uint32_t x) {  // x = 0xFFFFFFFF is a no-op; from the top down: in place
#pragma unroll
    for (int t = K; t >= 1; --t) asm("v_med3_u32 %0, %1, %0, %2" : "+v"(key[t]) : "v"(key[t - 1]), "v"(x));
    asm("v_min_u32 %0, %0, %1" : "+v"(key[0]) : "v"(x));
  }
  __device__ __forceinline__ uint32_t at(uint32_t i) const {  // key[i] without dynamic register indexing
    uint32_t v = key[K];
#pragma unroll
    for (int t = 0; t < K; ++t) v = (uint32_t)t == i ? key[t] : v;
    return v;
  }
};

// ---- FIT 2: the nine pivot moments of a query, reduced across the sixteen lanes of a DPP row --------------------------------------------------
// v += row_ror(v, 8), 4, 2, 1: after four rotations every lane of the row holds the sum of all sixteen (each lane in its own order of additions;
// the owner keeps its own).  A 64-bit value moves as two v_mov_b32 dpp (gfx950 has no 64-bit DPP rotation): 3 instructions per stage and sum.
template <int CTRL>
__device__ __forceinline__ double row_ror_f64(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
__device__ __forceinline__ double row16_sum(double v) {
  v += row_ror_f64<0x128>(v);  // row_ror:8
  v += row_ror_f64<0x124>(v);  // row_ror:4
  v += row_ror_f64<0x122>(v);  // row_ror:2
  v += row_ror_f64<0x121>(v);  // row_ror:1
  return v;
}

struct Tile2Args {
  TileArgs t;
  double s;        // f32 coordinates are (x - c) * s
  double s2;       // s * s: exact squared distance -> key units
  double eps;      // bound on |f32 squared distance - exact squared distance * s2| for two points of a box (key units)
  float lim0;      // scan limit before the first insertion round: tau0 * s2 + eps, rounded up
  float kx, kyz;   // f32 coordinate -> fine x cells / rows: inv_hx / s, inv_h / s
  uint32_t gap;    // bins (F = key >> slot bits) two list entries must be apart for their exact order to be certain
  uint32_t f_max;  // the k-th entry's bin must lie below this one: upper edge + eps < tau0 * s2
};

// FIT: 2 = the covariance as a CROSS-LANE reduction (BASELINE.json's "per-point 3x3 covariance wavefront reduction", taken literally: sixteen lanes per
//          query, one neighbour each, the nine sums folded across the row with DPP rotations) -- built to be MEASURED against FIT 1 (round-4 review,
//          row NS-1; PST_KNN_FIT=rows, k <= 16, no neighbour lists, unrotated grids); see rows16_moments below for the count and the result;
//      1 = the plane fit in ONE pass about the query (plane_fit_pivot: 12 running sums, every neighbour fetched once and not kept);
//      0 = the reference's order of operations (plane_fit: centroid, then moments -- the 16 neighbours are held across the two passes, which
//          at 128 registers means scratch memory; kept for bit comparison and the A/B, PST_KNN_FIT=seq)
template <int K, int THREADS, int CAP, bool P3LDS, int BATCH, int WPE, bool WITH_KNN, bool ROT, int FIT>
__global__ __launch_bounds__(THREADS, WPE) void knn_tile2_kernel(const Tile2Args aa) {
  const TileArgs& a = aa.t;
  constexpr int CS = CAP + BATCH;
  constexpr uint32_t SLOT_BITS = CAP <= 2048 ? 11u : 12u, SLOT_MASK = (1u << SLOT_BITS) - 1u;
  static_assert(CAP + BATCH <= (1 << SLOT_BITS), "slots must fit the key");
  constexpr int kQ = 16;  // queue entries per lane
  // staged points: f32 box-relative copy (scan) + f64 originals (proof and plane fit), or + the map to the sorted array; one region,
  // which holds the raw 32-bit directory entries ([NR][32]) until the points arrive
  constexpr int PT_BYTES = P3LDS ? 36 * CS : 16 * CS;
  static_assert(PT_BYTES >= kMaxRows * 32 * 4, "raw directory does not fit the point region");
  __shared__ __attribute__((aligned(16))) uint8_t pts[PT_BYTES];
  double* P3s = reinterpret_cast<double*>(pts);                                   // [3][CS] (P3LDS)
  float* R3 = reinterpret_cast<float*>(pts + (P3LDS ? 24 * CS : 0));              // [3][CS]
  uint32_t* Jmap = reinterpret_cast<uint32_t*>(pts + 12 * CS);                    // [CS]    (!P3LDS)
  uint32_t* raw = reinterpret_cast<uint32_t*>(pts);
  __shared__ uint16_t ldir[kMaxDir];
  __shared__ uint32_t g0[kMaxRows];
  __shared__ uint32_t rbase[kMaxRows + 1];
  __shared__ uint32_t qpre[kMaxQRows + 1];
  __shared__ uint16_t qbuf[kQ * THREADS];
  __shared__ uint32_t s_next;
  constexpr int NW = THREADS / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t wg = xcd_block_id();
  if (wg >= a.n_boxes || (a.n_boxes_dev && wg >= *a.n_boxes_dev)) return;
  if (a.ablate & 64u) return;
  const uint32_t box = a.box_list ? a.box_list[wg] : wg;
  PST_KNN_STAT(const long long t_start = clock64(); long long t_flush = 0;)
  const GridParams& g = a.g;
  const int dim0 = (int)g.dim[0], dim1 = (int)g.dim[1], dim2 = (int)g.dim[2];
  const int bxi = (int)(box % a.nbx), byi = (int)((box / a.nbx) % a.nby), bzi = (int)(box / (a.nbx * a.nby));
  const int X0 = bxi * (int)a.bx, Y0 = byi * (int)a.by, Z0 = bzi * (int)a.bz;
  const int XH = (int)g.rx + 1;
  const int HX = (int)a.bx + 2 * XH, HY = (int)a.by + 2 * kHalo, HZ = (int)a.bz + 2 * kHalo;
  const int NC1 = HX + 1, NR = HY * HZ, nqy = (int)a.by, nqr = (int)(a.by * a.bz);
  auto row_in_grid = [&](int r, uint64_t& cell0) {
    const int y = Y0 - kHalo + r % HY, z = Z0 - kHalo + r / HY;
    cell0 = ((uint64_t)z * (uint64_t)dim1 + (uint64_t)y) * (uint64_t)dim0;
    return y >= 0 && y < dim1 && z >= 0 && z < dim2;
  };
  auto clamp_x = [&](int x) { return (uint32_t)(x < 0 ? 0 : (x > dim0 ? dim0 : x)); };
  // the box's centre in the cloud's own coordinates: every f32 coordinate is relative to it
  double ctr[3];
  {
    const double cu = g.org[0] + ((double)X0 + 0.5 * (double)a.bx) * g.hx, cv = g.org[1] + ((double)Y0 + 0.5 * (double)a.by) * g.h,
                 cw = g.org[2] + ((double)Z0 + 0.5 * (double)a.bz) * g.h;
    if constexpr (ROT) {  // (u, v, w) = rot (x - rot_c)  =>  x = rot_c + rot^T (u, v, w)   (ROT: a separate instance -- the nine matrix entries and the
                          //  centre are 24 scalar registers the kernel does not have to spare, see launch_knn_tile)
      ctr[0] = g.rot_c[0] + g.rot[0] * cu + g.rot[3] * cv + g.rot[6] * cw;
      ctr[1] = g.rot_c[1] + g.rot[1] * cu + g.rot[4] * cv + g.rot[7] * cw;
      ctr[2] = g.rot_c[2] + g.rot[2] * cu + g.rot[5] * cv + g.rot[8] * cw;
    } else { ctr[0] = cu; ctr[1] = cv; ctr[2] = cw; }
  }

  // ---- A: raw directory entries of every halo row --------------------------------------------------------------------------------
  for (int r = (int)(tid >> 5); r < NR; r += THREADS / 32) {
    const int c = (int)(tid & 31u);
    if (c < NC1) {
      uint64_t c0;
      raw[r * 32 + c] = row_in_grid(r, c0) ? a.cell_start[c0 + clamp_x(X0 - XH + c)] : 0u;
    }
  }
  if (tid == 0) s_next = 0;
#pragma unroll
  for (int i = 0; i < kQ; ++i) qbuf[i * THREADS + tid] = 0;
  __syncthreads();  // (1)
  auto qrow_halo = [&](int qr) { return (kHalo + qr / nqy) * HY + kHalo + qr % nqy; };
  uint32_t total;
  {
    uint32_t v[3], sum = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int r = 3 * (int)lane + u; v[u] = r < NR ? raw[r * 32 + HX] - raw[r * 32] : 0u; sum += v[u]; }
    uint32_t run = wave_excl_scan(sum, lane, total);
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int r = 3 * (int)lane + u; if (r < NR) { rbase[r] = run; g0[r] = raw[r * 32]; } run += v[u]; }
      if (lane == 0) rbase[NR] = total;
      uint32_t cnt = 0;
      if ((int)lane < nqr) { const int r = qrow_halo((int)lane); cnt = raw[r * 32 + XH + (int)a.bx] - raw[r * 32 + XH]; }
      uint32_t tot;
      const uint32_t ex = wave_excl_scan(cnt, lane, tot);
      if ((int)lane < nqr) qpre[lane] = ex;
      if (lane == 0) qpre[nqr] = tot;
    }
  }
  if (total == 0) return;
  if (total > (uint32_t)CAP) {
    for (int qr = (int)wave; qr < nqr; qr += NW) {
      const int r = qrow_halo(qr);
      const uint32_t s0 = raw[r * 32 + XH], s1 = raw[r * 32 + XH + (int)a.bx];
      for (uint32_t j = s0 + lane; j < s1; j += 64) a.fb_list[atomicAdd(a.fb_count, 1u)] = j;
    }
    return;
  }
  // ---- C: coalesced copy of the row segments; the loads of the first pass are requested HERE, off the raw directory, and land while the
  //         local directory is written -------------------------------------------------------------------------------------------
  constexpr int kRows = THREADS >= 1024 ? 4 : (THREADS >= 512 ? 4 : 8), kChunks = 4;
  double cv[kRows][kChunks];
  uint32_t len3[kRows];
  const bool copy_on = !(a.ablate & 16u);
#pragma unroll
  for (int u = 0; u < kRows; ++u) {
    const int r = (int)wave * kRows + u;
    const uint32_t first = r < NR ? raw[r * 32] : 0u;
    len3[u] = r < NR && copy_on ? 3u * (raw[r * 32 + HX] - first) : 0u;
    const double* src = a.sxyz + 3ull * first;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) cv[u][c] = lane + 64u * c < len3[u] ? src[lane + 64u * c] : 0.0;
  }
  __syncthreads();  // (2) row sums published
  // ---- B: local 16-bit directory --------------------------------------------------------------------------------------------------
  for (int r = (int)(tid >> 5); r < NR; r += THREADS / 32) {
    const int c = (int)(tid & 31u);
    if (c < NC1) ldir[r * NC1 + c] = (uint16_t)(raw[r * 32 + c] - g0[r] + rbase[r]);
  }
  __syncthreads();  // (3) the raw directory is dead: the point region can be filled
  {
    auto put = [&](uint32_t base, uint32_t gfirst, uint32_t e, double v) __attribute__((always_inline)) {
      const uint32_t pt = e / 3u, c = e - 3u * pt;
      const double cc = c == 0 ? ctr[0] : (c == 1 ? ctr[1] : ctr[2]);
      R3[c * CS + base + pt] = (float)((v - cc) * aa.s);
      if constexpr (P3LDS) P3s[c * CS + base + pt] = v;
      else if (c == 0) Jmap[base + pt] = gfirst + pt;
    };
    auto store_rows = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int r = r0 + u;
        if (r >= NR) continue;
        const uint32_t base = rbase[r], gf = g0[r];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) if (lane + 64u * c < len3[u]) put(base, gf, lane + 64u * c, cv[u][c]);
        if (len3[u] > 64u * kChunks) {
          const double* src = a.sxyz + 3ull * gf;
          for (uint32_t e = lane + 64u * kChunks; e < len3[u]; e += 64) put(base, gf, e, src[e]);
        }
      }
    };
    store_rows((int)wave * kRows);
    for (int r0 = (NW + (int)wave) * kRows; r0 < NR && copy_on; r0 += NW * kRows) {  // boxes with more than NW * kRows rows
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int r = r0 + u;
        len3[u] = r < NR ? 3u * (rbase[r + 1] - rbase[r]) : 0u;
        const double* src = a.sxyz + 3ull * (r < NR ? g0[r] : 0u);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) cv[u][c] = lane + 64u * c < len3[u] ? src[lane + 64u * c] : 0.0;
      }
      store_rows(r0);
    }
    // the slots behind the last staged point: a step may read them; far away in f32 (their squared distance overflows to +inf)
    if (tid < (uint32_t)BATCH) {
      R3[total + tid] = 3.0e38f; R3[CS + total + tid] = 3.0e38f; R3[2 * CS + total + tid] = 3.0e38f;
      if constexpr (!P3LDS) Jmap[total + tid] = 0u;
    }
  }
  __syncthreads();  // (4) points staged
  PST_KNN_MARK("staged");
  PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg + 8, (unsigned long long)(clock64() - t_start));)
  const float* Rx = R3;
  const float* Ry = R3 + CS;
  const float* Rz = R3 + 2 * CS;
  const uint32_t Q = qpre[nqr];
  const uint32_t m = a.nf < a.k ? a.nf : a.k;
  auto exact_xyz = [&](uint32_t sl, double& x, double& y, double& z) __attribute__((always_inline)) {
    if constexpr (P3LDS) { x = P3s[sl]; y = P3s[CS + sl]; z = P3s[2 * CS + sl]; }
    else { const double* pp = a.sxyz + 3ull * Jmap[sl]; x = pp[0]; y = pp[1]; z = pp[2]; }
  };

  for (;;) {
    uint32_t c0 = 0;
    if (lane == 0) c0 = atomicAdd(&s_next, 64u);
    c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
    if (c0 >= Q) break;
    PST_KNN_MARK("chunk_begin");
    const uint32_t q = c0 + lane;
    const bool active = q < Q;
    PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg + 2, 1ull);)
    PST_KNN_STAT(const long long t_chunk = clock64(); t_flush = 0;)
    int qr = 0;
    {
      int lo = 0, hi = nqr;
      const uint32_t qq = active ? q : 0u;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qpre[mid] <= qq) lo = mid; else hi = mid; }
      qr = lo;
    }
    const int hr = qrow_halo(qr);
    const uint32_t slot = active ? (uint32_t)ldir[hr * NC1 + XH] + (q - qpre[qr]) : 0u;
    const float rqx = Rx[slot], rqy = Ry[slot], rqz = Rz[slot];
    const uint32_t j = g0[hr] + (slot - rbase[hr]);
    // The query's position in the halo's cell coordinates, from its f32 copy: (u, v, w) = rot (x - c) + (centre of the box in the frame),
    // exact to ~2e-6 cells -- far inside the slack of the trims below.  Only the SUM cell + fraction enters the trims, so a query that
    // sits on a cell border may be booked into either cell.  A query outside the grid's box (clamped into a boundary cell) must go to the
    // exact search (see the first form); "outside" is decided with 1e-4 cells of margin towards the inside.
    int B;
    bool outside;
    float fx, fy, fz;
    {
      float ru = rqx, rv = rqy, rw = rqz;
      if constexpr (ROT) {
        ru = (float)g.rot[0] * rqx + (float)g.rot[1] * rqy + (float)g.rot[2] * rqz;
        rv = (float)g.rot[3] * rqx + (float)g.rot[4] * rqy + (float)g.rot[5] * rqz;
        rw = (float)g.rot[6] * rqx + (float)g.rot[7] * rqy + (float)g.rot[8] * rqz;
      }
      const float px = ru * aa.kx + (0.5f * (float)a.bx + (float)XH);        // halo cells along x
      const float py = rv * aa.kyz + 0.5f * (float)a.by, pz = rw * aa.kyz + 0.5f * (float)a.bz;  // query rows along y, z (0 = the box's first row)
      const float cxf = fminf(fmaxf(floorf(px), (float)XH), (float)(XH + (int)a.bx - 1));
      const int qy_l = qr % nqy, qz_l = qr / nqy;
      fx = fminf(fmaxf(px - cxf, 0.0f), 1.0f);
      fy = fminf(fmaxf(py - (float)qy_l, 0.0f), 1.0f);
      fz = fminf(fmaxf(pz - (float)qz_l, 0.0f), 1.0f);
      B = hr * NC1 + (active ? (int)cxf : XH);
      const float m_in = 1e-4f;
      outside = (X0 == 0 && px - (float)XH < m_in) || (X0 + (int)a.bx >= dim0 && px - (float)(XH + dim0 - X0) > -m_in) ||
                (Y0 == 0 && py < m_in) || (Y0 + (int)a.by >= dim1 && py - (float)(dim1 - Y0) > -m_in) ||
                (Z0 == 0 && pz < m_in) || (Z0 + (int)a.bz >= dim2 && pz - (float)(dim2 - Z0) > -m_in) || !(px == px && py == py && pz == pz);
    }

    KBestU32<K> best;
    best.init();
    // per-lane queue of candidate slots: entry i of lane tid at qbuf[i * THREADS + tid]; qaddr = LDS byte address of the next free entry
    constexpr uint32_t QSTRIDE = 2u * THREADS;
    const uint32_t qbase = (uint32_t)(uintptr_t)(&qbuf[tid]);
    uint32_t qaddr = qbase;
    const uint32_t qroom = qbase + (uint32_t)(kQ - BATCH) * QSTRIDE;  // a lane scans only while BATCH more entries fit
    float lim = aa.lim0;
    auto flush = [&]() __attribute__((always_inline)) {
      // nothing here is predicated per lane: a lane without an entry reads a stale (valid) slot and inserts 0xFFFFFFFF, which changes nothing
      const uint32_t qn = (qaddr - qbase) / QSTRIDE;
      // the longest queue of the wave (scalar: five ballots), so that the loop below is a counted one with scalar control
      uint32_t qmax = 0;
#pragma unroll
      for (int bit = 4; bit >= 0; --bit)
        qmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(qn >= (qmax | (1u << bit))) ? qmax | (1u << bit) : qmax));
      uint32_t p0 = qbuf[tid];
      uint32_t p1 = qbuf[THREADS + tid];
      float x0 = Rx[p0], y0 = Ry[p0], z0 = Rz[p0];
      PST_KNN_MARK("flush_loop_begin");
      for (uint32_t i = 0; i < qmax; ++i) {
        PST_KNN_MARK("flush_iter");
        const bool has = i < qn;
        PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg + 1, 1ull);)
        const uint32_t p2 = qbuf[(i + 2 < (uint32_t)kQ ? i + 2 : (uint32_t)kQ - 1u) * THREADS + tid];
        const float x1 = Rx[p1], y1 = Ry[p1], z1 = Rz[p1];
        // the same three operations, in the same order, as the scan's packed ones: the same value
        const float dx = x0 - rqx, dy = y0 - rqy, dz = z0 - rqz;
        float d2 = dx * dx;
        d2 = __builtin_fmaf(dy, dy, d2);
        d2 = __builtin_fmaf(dz, dz, d2);
        uint32_t fix = (uint32_t)d2;  // v_cvt_u32_f32: toward zero, monotone
        fix = fix < 0x1FFFFEu ? fix : 0x1FFFFEu;
        uint32_t key = SLOT_BITS == 12u ? (((fix >> 1) << SLOT_BITS) | p0) : ((fix << SLOT_BITS) | p0);
        key = has ? key : 0xFFFFFFFFu;
        best.insert(key);
        p0 = p1; p1 = p2; x0 = x1; y0 = y1; z0 = z1;
        PST_KNN_MARK("flush_iter_end");
      }
      PST_KNN_MARK("flush_loop_end");
      qaddr = qbase;
      // every later candidate that could still enter the list has an f32 distance below the upper edge of the last key's bin
      const uint32_t kl = best.key[K];
      const float edge = SLOT_BITS == 12u ? (float)(((kl >> SLOT_BITS) + 1u) << 1) : (float)((kl >> SLOT_BITS) + 1u);
      lim = kl != 0xFFFFFFFFu ? __builtin_fminf(lim, edge) : lim;
    };
    uint32_t ent[kSegs];
    {
      const float bound = (float)a.tau0 * ((float)(g.inv_h * g.inv_h) * 1.00002f) + 2e-5f, rxf = (float)g.rx * 1.00001f, xh = (float)XH;
      const bool on = active && !outside;
#pragma unroll
      for (int sg = 0; sg < kSegs; ++sg) {
        constexpr int kDy[kSegs] = {0, -1, 1, 0, 0, -1, 1, -1, 1}, kDz[kSegs] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        const int dy = kDy[sg], dz = kDz[sg];
        const float ty = (float)dy - fy, tz = (float)dz - fz;
        const float gy = fmaxf(0.0f, fmaxf(ty, -ty - 1.0f)), gz = fmaxf(0.0f, fmaxf(tz, -tz - 1.0f));
        const float r2 = bound - (gy * gy + gz * gz);
        const float ext = __builtin_amdgcn_sqrtf(fmaxf(r2, 0.0f)) * rxf + 8e-6f;  // (+ the f32 position's own error)
        const float lo_f = fmaxf(floorf(fx - ext), -xh), hi_f = fminf(floorf(fx + ext), xh);
        const int Bd = B + (dz * HY + dy) * NC1;
        const uint32_t s0 = ldir[Bd + (int)lo_f], s1 = ldir[Bd + (int)hi_f + 1];
        ent[sg] = (on && r2 > 0.0f) ? (s0 | (s1 << 16)) : 0u;
      }
    }
    const f2v qx2 = {rqx, rqx}, qy2 = {rqy, rqy}, qz2 = {rqz, rqz};
    PST_KNN_STAT(const long long t_scan = clock64(); if (lane == 1) atomicAdd(a.dbg + 9, (unsigned long long)(t_scan - t_chunk));)
    uint32_t left = kSegs;
    uint32_t p = 0, pe = 0;
    PST_KNN_MARK("scan_loop_begin");
    for (;;) {
      PST_KNN_MARK("step_head");
      // ONE shift of the segment table per step, written as selects (a branch made the compiler copy the table at the loop's latch):
      // a lane that drew an empty range idles this step and draws again in the next
      {
        const bool take = p >= pe && left != 0u;
        const uint32_t e = ent[0];
#pragma unroll
        for (int sg = 0; sg + 1 < kSegs; ++sg) ent[sg] = take ? ent[sg + 1] : ent[sg];
        p = take ? (e & 0xFFFFu) : p;
        pe = take ? (e >> 16) : pe;
        left = take ? left - 1u : left;
      }
      // A lane whose queue could overflow WAITS; the queues are emptied when at most flush_at lanes can go on scanning
      const bool room = qaddr <= qroom;
      const bool scan = p < pe && room;
      const uint64_t can = __builtin_amdgcn_ballot_w64(scan || (p >= pe && left != 0u)), waiting = __builtin_amdgcn_ballot_w64(p < pe && !room);
      if (can != 0 && !(waiting != 0 && (uint32_t)__builtin_popcountll(can) <= a.flush_at)) {
        PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg, 1ull); atomicAdd(a.dbg + 3, (unsigned long long)(scan ? BATCH : 0));)
        if (scan) {  // ONE predicate per step; inside, nothing branches and the execution mask stays put
          PST_KNN_MARK("step_body");
          f2v cxs[BATCH / 2], cys[BATCH / 2], czs[BATCH / 2];
#pragma unroll
          for (int u = 0; u < BATCH / 2; ++u) {
            cxs[u] = f2v{Rx[p + 2 * u], Rx[p + 2 * u + 1]};
            cys[u] = f2v{Ry[p + 2 * u], Ry[p + 2 * u + 1]};
            czs[u] = f2v{Rz[p + 2 * u], Rz[p + 2 * u + 1]};
          }
#pragma unroll
          for (int u = 0; u < BATCH / 2; ++u) {
            const f2v dx = cxs[u] - qx2, dy = cys[u] - qy2, dz = czs[u] - qz2;
            f2v d2 = dx * dx;
            d2 = __builtin_elementwise_fma(dy, dy, d2);
            d2 = __builtin_elementwise_fma(dz, dz, d2);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              // the slot is written in any case (a scanning lane has room for BATCH entries) and kept if the candidate passes; candidates
              // behind the end of the segment are staged points like any other
              *reinterpret_cast<uint16_t __attribute__((address_space(3)))*>(qaddr) = (uint16_t)(p + (uint32_t)(2 * u + hh));
              const bool pass = d2[hh] <= lim;
              PST_KNN_STAT(if (pass) atomicAdd(a.dbg + 4, 1ull);)
              qaddr += pass ? QSTRIDE : 0u;
            }
          }
          const uint32_t pn = p + (uint32_t)BATCH;
          p = pn < pe ? pn : pe;
          PST_KNN_MARK("step_body_end");
        }
      } else {
        PST_KNN_STAT(const long long t_f0 = clock64();)
        PST_KNN_STAT(if (lane == 1 && __builtin_amdgcn_ballot_w64(qaddr != qbase)) atomicAdd(a.dbg + 7, 1ull);)
        if (__builtin_amdgcn_ballot_w64(qaddr != qbase)) flush();  // ONE inlined copy of the insertion code
        PST_KNN_STAT(t_flush += clock64() - t_f0;)
        if (!__builtin_amdgcn_ballot_w64(p < pe || left != 0u)) break;
      }
    }
    PST_KNN_STAT(const long long t_proof = clock64(); if (lane == 1) { atomicAdd(a.dbg + 10, (unsigned long long)(t_proof - t_scan - t_flush)); atomicAdd(a.dbg + 11, (unsigned long long)t_flush); })
    // ---- the proof ------------------------------------------------------------------------------------------------------------
    // Keys ascend.  F = key >> SLOT_BITS is the candidate's f32 squared distance in bins of `unit` key units, and the exact squared
    // distance (times s2) lies within eps of the f32 one.  Two entries whose bins are at least `gap` = floor(1 + 2 eps / unit) + 1
    // apart are therefore in their exact order, strictly; and everything that is NOT among the first k entries has a key >= key[k].
    //   * all k adjacent pairs (0,1) .. (k-1,k) at least `gap` apart: the first k entries ARE the k nearest, in ascending exact order;
    //   * a closer pair INSIDE the list (about 2e-4 of the pairs) is settled with the exact f64 distances of its two points (ties by
    //     slot, as the first form does) -- unless the next pair is close too (three entries in one window: exact search);
    //   * a close pair at the k / k+1 boundary cannot be settled here (unseen candidates may sit in the same window): exact search.
    // The k-th exact distance must also lie below tau0 (the 3 x 3 rows cover that radius): its bin's upper edge + eps < tau0 * s2.
    bool ok;
    uint32_t nb[K];
    PST_KNN_MARK("proof_begin");
    {
      const uint32_t kk = a.k;
      uint32_t amb = 0;  // bit t: the pair (t, t + 1) is closer than `gap`
#pragma unroll
      for (int t = 0; t < K; ++t) {
        if ((uint32_t)t < kk) {
          const uint32_t d = (best.key[t + 1] >> SLOT_BITS) - (best.key[t] >> SLOT_BITS);  // (an empty entry: F = all ones, never close)
          amb |= d < aa.gap ? 1u << t : 0u;
        }
      }
      // key[kk - 1] without an indexed read: the compiler turns a chain of `t == index` selects back into an array in scratch memory (17 dwords
      // stored per query for one indexed load: 13.6 GB of write-back per 10^8 points); `t < kk` selects, ascending, end on the same entry
      uint32_t k_last = best.key[0];
#pragma unroll
      for (int t = 1; t < K; ++t) k_last = (uint32_t)t < kk ? best.key[t] : k_last;
      ok = k_last != 0xFFFFFFFFu && (k_last >> SLOT_BITS) < aa.f_max;   // k entries, the k-th inside tau0
      ok = ok && !(amb & (amb >> 1)) && !((amb >> (kk - 1u)) & 1u);      // no chain of close pairs, none at the boundary
      amb = ok ? amb : 0u;
      // the neighbours' LDS slots, in list order (a separate array: the key registers are not written outside the insertion rounds)
#pragma unroll
      for (int t = 0; t < K; ++t) nb[t] = best.key[t] & SLOT_MASK;
      if (a.ablate) {  // (tuning runs: every query counts as done, whatever its list holds -- empty entries must still be staged slots)
#pragma unroll
        for (int t = 0; t < K; ++t) nb[t] = nb[t] < total ? nb[t] : 0u;
      }
      if (__builtin_amdgcn_ballot_w64(amb != 0u)) {  // rare: about one wave in five settles one pair
        double qx, qy, qz;
        exact_xyz(slot, qx, qy, qz);
#pragma unroll
        for (int t = 0; t + 1 < K; ++t) {
          if (__builtin_amdgcn_ballot_w64((amb >> t) & 1u)) {
            const bool mine = (amb >> t) & 1u;
            const uint32_t s0 = mine ? nb[t] : 0u, s1 = mine ? nb[t + 1] : 0u;
            double x0, y0, z0, x1, y1, z1;
            exact_xyz(s0, x0, y0, z0);
            exact_xyz(s1, x1, y1, z1);
            const double ax = x0 - qx, ay = y0 - qy, az = z0 - qz, bx = x1 - qx, by = y1 - qy, bz = z1 - qz;
            const double e0 = ax * ax + ay * ay + az * az, e1 = bx * bx + by * by + bz * bz;
            const bool swap = mine && (e1 < e0 || (e1 == e0 && s1 < s0));
            nb[t] = swap ? s1 : nb[t];
            nb[t + 1] = swap ? s0 : nb[t + 1];
            if (mine && e1 == e0 && s1 == s0) ok = false;  // the same point twice: a step ran over the end of a segment into another one of this query
          }
        }
      }
    }
    PST_KNN_MARK("proof_end");
    const bool done = a.ablate ? true : !outside && ok;
    PST_KNN_STAT(if (active && !outside && !ok) atomicAdd(a.dbg + 5, 1ull);)
    if (active && !done) a.fb_list[atomicAdd(a.fb_count, 1u)] = j;
    // (the original index is requested here, together with the neighbours' coordinates: a load left in flight across the scan loop is
    //  waited for at the loop's head, and this one comes from HBM)
    const uint32_t orig = active && done ? a.out.sidx[j] : 0u;
    PST_KNN_STAT(const long long t_fit = clock64(); if (lane == 1) atomicAdd(a.dbg + 12, (unsigned long long)(t_fit - t_proof));)
    // FIT 2: sixteen rounds; in round r the sixteen lanes of row g serve the query of lane 16 g + r (same row: the owner keeps its own lane's
    // totals), lane t of the row fetching neighbour t.  The lists cross the lanes through the candidate queue's LDS (free after the search).
    // Per round: 1 LDS read + 1 ds_bpermute (the owner's slot) + 2 gathers of 24 bytes + 3 subtractions + 6 products + 6 selects (lanes t >= m)
    // + 9 sums x 4 stages x 3 instructions + 18 selects (the owner keeps) ~ 150 wave instructions, x 16 rounds = 2400 per 64 queries, against
    // ~320 for the sixteen iterations of the one-lane loop it replaces (12 f64 operations + the gather per neighbour, all 64 lanes busy).
    double rows_s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (FIT == 2) {
      static_assert(K <= 16 && kQ >= K, "one DPP row per query");
#pragma unroll
      for (int t = 0; t < K; ++t) qbuf[t * THREADS + tid] = (uint16_t)nb[t];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t tl = lane & 15u, row0 = tid & ~15u;
      const bool use = tl < m;
#pragma unroll 1
      for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t owner = row0 + r;  // thread index of the query this row serves
        const uint32_t ns = qbuf[(tl < (uint32_t)K ? tl : 0u) * THREADS + owner];
        const uint32_t qs = (uint32_t)__shfl((int)slot, (int)((lane & ~15u) + r), 64);
        double qx, qy, qz, x, y, z;
        exact_xyz(qs, qx, qy, qz);
        exact_xyz(use && ns < total ? ns : qs, x, y, z);  // (an unfinished query's list holds empty entries: any staged point will do, its sums are not used)
        const double ux = use ? x - qx : 0.0, uy = use ? y - qy : 0.0, uz = use ? z - qz : 0.0;
        const double v[9] = {row16_sum(ux), row16_sum(uy), row16_sum(uz), row16_sum(ux * ux), row16_sum(ux * uy), row16_sum(ux * uz),
                             row16_sum(uy * uy), row16_sum(uy * uz), row16_sum(uz * uz)};
        const bool mine = tl == r;
#pragma unroll
        for (int c = 0; c < 9; ++c) rows_s[c] = mine ? v[c] : rows_s[c];
      }
    }
    if (active && done && !(a.ablate & 32u)) {
      if constexpr (WITH_KNN) {
        for (uint32_t t = 0; t < a.k; ++t) {
          uint32_t v = kNoIndex;
          uint32_t pl = 0;
#pragma unroll
          for (int u = 0; u < K; ++u) pl = (uint32_t)u == t ? nb[u] : pl;
          if (t < m) {
            if constexpr (P3LDS) {
              int lo = 0, hi = NR;
              while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rbase[mid] <= pl) lo = mid; else hi = mid; }
              v = a.out.sidx[g0[lo] + (pl - rbase[lo])];
            } else {
              v = a.out.sidx[Jmap[pl]];
            }
          }
          write_knn(a.out, orig, a.k, t, v);
        }
      }
      PST_KNN_MARK("fit_begin");
      Fit f{0, 0, 0, 0, 1};
      bool handed = false;
      if constexpr (FIT == 2) {
        bool ill = false;
        if (m < 3) f.ok = 0;
        else if (!(a.ablate & 2u)) {
          const double inv = 1.0 / (double)m;
          const double tx = rows_s[0] * inv, ty = rows_s[1] * inv, tz = rows_s[2] * inv;
          f = fit_from_covariance<true>(__builtin_fma(-rows_s[0], tx, rows_s[3]), __builtin_fma(-rows_s[0], ty, rows_s[4]), __builtin_fma(-rows_s[0], tz, rows_s[5]),
                                        __builtin_fma(-rows_s[1], ty, rows_s[6]), __builtin_fma(-rows_s[1], tz, rows_s[7]), __builtin_fma(-rows_s[2], tz, rows_s[8]), &ill);
        }
        ill = ill && a.fit_guard;
        if (ill) { a.fb_list[atomicAdd(a.fb_count, 1u)] = j; handed = true; }
      } else if constexpr (FIT == 1) {
        double qx, qy, qz;
        exact_xyz(slot, qx, qy, qz);
        bool ill = false;
        // The sixteen gathers in TWO batches of eight instead of sixteen dependent round trips.  plane_fit_pivot's loop tests `t < m` per neighbour,
        // every iteration is a basic block of its own and no load is lifted over a branch: slot -> sorted index (LDS) -> 24 bytes (L2) -> wait ->
        // twelve operations, sixteen times in a row (37 % of a wave's cycles for 13 % of its instructions, profiles/r05_knn_phases.txt).  Here all
        // sorted indices are read at once and eight neighbours' coordinates are in flight together; entries t >= m name the query itself, whose
        // u = 0 adds nothing to any sum: no branch, no select.  The sums are plane_fit_pivot's operations in plane_fit_pivot's order: bit-identical
        // results.  Only worth it since the kernel holds no scratch (normals_device.hpp kThirdAngleTable): with the solver's literals spilled the
        // 48 extra live registers spilled too and the same change LOST 0.8 %; without, +3.5 % (27.74 -> 26.80 ms, 6 of 6 pairs; four in flight and
        // sixteen in flight spill again).  -DPST_KNN_FIT_BATCH=0 restores the loop for the A/B.
#ifndef PST_KNN_FIT_BATCH
#define PST_KNN_FIT_BATCH 8
#endif
#if PST_KNN_FIT_BATCH > 0
        if constexpr (!P3LDS) {
          if (!(a.ablate & 2u)) {
            uint32_t jm[K];
#pragma unroll
            for (int u = 0; u < K; ++u) jm[u] = Jmap[(uint32_t)u < m ? nb[u] : slot];
            double sx = 0, sy = 0, sz = 0, mxx = 0, mxy = 0, mxz = 0, myy = 0, myz = 0, mzz = 0;
            constexpr int H = K < PST_KNN_FIT_BATCH ? K : PST_KNN_FIT_BATCH;
#pragma unroll
            for (int h = 0; h < K; h += H) {
              if (h > 0) {
#pragma unroll
                for (int u = 0; u < H; ++u) asm volatile("" : "+v"(jm[h + u]), "+v"(sx), "+v"(mzz));
              }
              double x[H], y[H], z[H];
#pragma unroll
              for (int u = 0; u < H; ++u) { const double* pp = a.sxyz + 3ull * jm[h + u]; x[u] = pp[0]; y[u] = pp[1]; z[u] = pp[2]; }
#pragma unroll
              for (int u = 0; u < H; ++u) {
                const double ux = x[u] - qx, uy = y[u] - qy, uz = z[u] - qz;
                sx += ux; sy += uy; sz += uz;
                mxx = __builtin_fma(ux, ux, mxx); mxy = __builtin_fma(ux, uy, mxy); mxz = __builtin_fma(ux, uz, mxz);
                myy = __builtin_fma(uy, uy, myy); myz = __builtin_fma(uy, uz, myz); mzz = __builtin_fma(uz, uz, mzz);
              }
            }
            f = pivot_fit_finish(m, sx, sy, sz, mxx, mxy, mxz, myy, myz, mzz, &ill);
          }
        } else
#endif
        if (!(a.ablate & 2u)) f = plane_fit_pivot<K>(m, qx, qy, qz, [&](uint32_t t, double& x, double& y, double& z) __attribute__((always_inline)) {
          uint32_t pl = 0;
#pragma unroll
          for (int u = 0; u < K; ++u) if ((uint32_t)u == t) pl = nb[u];
          exact_xyz(pl, x, y, z);
        }, &ill);
        // Conditioning is a property of the NEIGHBOURHOOD, not of the cloud (round-4 review): where the reference's solver amplifies last-bit
        // differences of the covariance (fit_from_covariance: `ill`), the one-pass result is not written; the query is handed to the exact
        // search behind this kernel, whose fit adds in the reference's own order of operations -- centroid, then moments.  Rare in
        // volume-filling clouds (4 queries in 1000 at k = 16); on wires, lattices and quantised planes inside such clouds it is what keeps
        // curvature and normal inside the parity window.  (Repeating the fit HERE, in the lane, cost 5.5 % of the call at 10^8 points: one ill lane
        // makes its whole wave walk two dependent gathers of k neighbours; handed back, the same queries cost the exact search a quarter more
        // work -- profiles/r05_abab.txt.)
        ill = ill && a.fit_guard;
        PST_KNN_STAT(if (ill) atomicAdd(a.dbg + 6, 1ull);)
        if (ill) { a.fb_list[atomicAdd(a.fb_count, 1u)] = j; handed = true; }
      } else if constexpr (!P3LDS && K <= 16) {
        // The reference's order of operations (centroid, then the moments about it: plane_fit<K, true>, bit for bit) WITHOUT holding the neighbours
        // across the two passes.  Rounds 3-5 fetched every neighbour once and kept 6 k registers per lane -- 96 of the 128 this kernel may use at
        // k = 16: 60-108 bytes of scratch per lane in every instance of this fit (the surfaces' and strips' instance: WRITE_SIZE 9.7 GB against
        // the one-pass instance's 7.0, profiles/r06_normals_knn16_sheet_rocprof.txt).  Round 6: both passes gather, eight neighbours in flight
        // (the batches of the one-pass fit above); the second gather of a neighbour hits the cache the first one filled.  Entries t >= m name
        // the query itself (a valid address) and are not added: m is a kernel argument, the tests below are wave-uniform.
        if (!(a.ablate & 2u)) {
          uint32_t jm[K];
#pragma unroll
          for (int u = 0; u < K; ++u) jm[u] = Jmap[(uint32_t)u < m ? nb[u] : slot];
          // the first KEEP neighbours stay in registers across the two passes, the others are gathered twice, four in flight
          constexpr int KEEP = K < 8 ? K : 8, H = 4;
          double kx[KEEP], ky[KEEP], kz[KEEP];
#pragma unroll
          for (int u = 0; u < KEEP; ++u) { const double* pp = a.sxyz + 3ull * jm[u]; kx[u] = pp[0]; ky[u] = pp[1]; kz[u] = pp[2]; }
          double ax = 0, ay = 0, az = 0;
#pragma unroll
          for (int u = 0; u < KEEP; ++u) if ((uint32_t)u < m) { ax += kx[u]; ay += ky[u]; az += kz[u]; }
#pragma unroll
          for (int h = KEEP; h < K; h += H) {
#pragma unroll
            for (int u = 0; u < H; ++u) asm volatile("" : "+v"(jm[h + u]), "+v"(ax));
            double x[H], y[H], z[H];
#pragma unroll
            for (int u = 0; u < H; ++u) { const double* pp = a.sxyz + 3ull * jm[h + u]; x[u] = pp[0]; y[u] = pp[1]; z[u] = pp[2]; }
#pragma unroll
            for (int u = 0; u < H; ++u) if ((uint32_t)(h + u) < m) { ax += x[u]; ay += y[u]; az += z[u]; }
          }
          const double div = (double)m;
          const double cx = ax / div, cy = ay / div, cz = az / div;
          double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
          auto moment = [&](double x, double y, double z) __attribute__((always_inline)) {  // compute_covariance_matrix :240-305, plane_fit's operations in plane_fit's order
            double d0 = x - cx, d1 = y - cy, d2 = z - cz;
            c11 += d1 * d1; c12 += d1 * d2; c22 += d2 * d2;
            const double dx = d0;
            d0 *= dx; d1 *= dx; d2 *= dx;
            c00 += d0; c01 += d1; c02 += d2;
          };
#pragma unroll
          for (int u = 0; u < KEEP; ++u) if ((uint32_t)u < m) moment(kx[u], ky[u], kz[u]);
#pragma unroll
          for (int h = KEEP; h < K; h += H) {
#pragma unroll
            for (int u = 0; u < H; ++u) asm volatile("" : "+v"(jm[h + u]), "+v"(c00), "+v"(c22));
            double x[H], y[H], z[H];
#pragma unroll
            for (int u = 0; u < H; ++u) { const double* pp = a.sxyz + 3ull * jm[h + u]; x[u] = pp[0]; y[u] = pp[1]; z[u] = pp[2]; }
#pragma unroll
            for (int u = 0; u < H; ++u) if ((uint32_t)(h + u) < m) moment(x[u], y[u], z[u]);
          }
          if (m < 3) f.ok = 0;  // Err(...) :293-295 -> unwrap panic :471
          else f = fit_from_covariance(c00, c01, c02, c11, c12, c22);
        }
      } else {
        if (!(a.ablate & 2u)) f = plane_fit<K, true>(m, [&](uint32_t t, double& x, double& y, double& z) __attribute__((always_inline)) {
          uint32_t pl = 0;
#pragma unroll
          for (int u = 0; u < K; ++u) if ((uint32_t)u == t) pl = nb[u];
          exact_xyz(pl, x, y, z);
        });
      }
      PST_KNN_MARK("fit_end");
      if (!handed) write_record(a.out, orig, f);
      PST_KNN_MARK("results_end");
    }
    PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg + 13, (unsigned long long)(clock64() - t_fit));)
  }
  PST_KNN_STAT(if (lane == 1) atomicAdd(a.dbg + 14, (unsigned long long)(clock64() - t_start));)
}

// ---- density probe: how many points does a ball of radius h (and h / 2) around a typical point hold? ---------------------------------
// The grid's cell edge h should be the radius R0 of the ball expected to hold M points.  The volume of the bounding box gives that only
// for clouds that fill their box: a surface in a 3-D box (the LiDAR case) has far more points inside R0 than n * ball / box.  The probe
// counts, for a sample of the sorted points, the points within h / 2 and within h (exact: the 3 x 3 rows around a point cover radius h),
// and the host fits a power law through the two means (normals.hip).
__global__ __launch_bounds__(kBlock) void knn_probe_kernel(const double* __restrict__ sxyz, const uint32_t* __restrict__ cell_start, GridParams g, uint32_t nf,
                                                           uint32_t stride, unsigned long long* __restrict__ sums) {
  const uint32_t j = (blockIdx.x * kBlock + threadIdx.x) * stride;
  unsigned long long c_half = 0, c_full = 0, one = 0;
  if (j < nf) {
    const double qx = sxyz[3 * (uint64_t)j], qy = sxyz[3 * (uint64_t)j + 1], qz = sxyz[3 * (uint64_t)j + 2];
    double qu, qv, qw;
    grid_frame(g, qx, qy, qz, qu, qv, qw);
    const int cx = (int)cell_coord(qu, g.org[0], g.inv_hx, g.dim[0]), cy = (int)cell_coord(qv, g.org[1], g.inv_h, g.dim[1]),
              cz = (int)cell_coord(qw, g.org[2], g.inv_h, g.dim[2]);
    const double r2 = g.h * g.h, r2h = 0.25 * r2;
    const int x0 = max(cx - (int)g.rx, 0), x1 = min(cx + (int)g.rx, (int)g.dim[0] - 1);
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy, z = cz + dz;
        if (y < 0 || y >= (int)g.dim[1] || z < 0 || z >= (int)g.dim[2]) continue;
        const uint64_t row = ((uint64_t)z * g.dim[1] + (uint64_t)y) * g.dim[0];
        const uint32_t p0 = cell_start[row + (uint32_t)x0], p1 = cell_start[row + (uint32_t)x1 + 1];
        for (uint32_t p = p0; p < p1; ++p) {
          const double dx = sxyz[3 * (uint64_t)p] - qx, ddy = sxyz[3 * (uint64_t)p + 1] - qy, ddz = sxyz[3 * (uint64_t)p + 2] - qz;
          const double d = dx * dx + ddy * ddy + ddz * ddz;
          c_half += d < r2h;
          c_full += d < r2;
        }
      }
    one = 1;
  }
  // wave sums, one atomic per wave and counter
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    c_half += shfl_xor_any(c_half, off); c_full += shfl_xor_any(c_full, off); one += shfl_xor_any(one, off);
  }
  if ((threadIdx.x & 63u) == 0 && one) { atomicAdd(sums, c_half); atomicAdd(sums + 1, c_full); atomicAdd(sums + 2, one); }
}

// ---- box census: what would a box shape stage?  One thread per box: staged points (halo) and queries, from the directory alone.
// sums: [0] queries, [1] staged points over non-empty boxes, [2] queries of boxes whose halo exceeds the capacity, [3] non-empty boxes
__global__ __launch_bounds__(kBlock) void knn_census_kernel(const uint32_t* __restrict__ cell_start, GridParams g, uint32_t bx, uint32_t by, uint32_t bz,
                                                            uint32_t nbx, uint32_t nby, uint32_t n_boxes, uint32_t cap, unsigned long long* __restrict__ sums,
                                                            uint32_t* __restrict__ list, uint32_t* __restrict__ list_count, const uint32_t* __restrict__ box_q) {
  const uint32_t box = blockIdx.x * kBlock + threadIdx.x;
  unsigned long long q = 0, staged = 0, lost = 0, occupied = 0;
  if (box < n_boxes) {
    const int dim0 = (int)g.dim[0], dim1 = (int)g.dim[1], dim2 = (int)g.dim[2], XH = (int)g.rx + 1;
    const int X0 = (int)(box % nbx) * (int)bx, Y0 = (int)((box / nbx) % nby) * (int)by, Z0 = (int)(box / (nbx * nby)) * (int)bz;
    auto cl = [&](int x) { return (uint32_t)(x < 0 ? 0 : (x > dim0 ? dim0 : x)); };
    // the queries first: a box without one stages nothing, and its halo rows are not looked up (a surface in a 3-D grid: 88 % of the boxes).
    // box_q: they have been counted from the sorted cell numbers (knn_box_queries_kernel) -- looked up in the directory, the box boundaries
    // of every row touch every line of it: 7.5 GB for the LiDAR sheet of 10^8 points, 1.8 ms
    if (box_q) q = box_q[box];
    else
    for (int z = Z0; z < Z0 + (int)bz && z < dim2; ++z)
      for (int y = Y0; y < Y0 + (int)by && y < dim1; ++y) {
        const uint64_t row = ((uint64_t)z * dim1 + (uint64_t)y) * dim0;
        q += cell_start[row + cl(X0 + (int)bx)] - cell_start[row + cl(X0)];
      }
    if (q != 0)
      for (int z = Z0 - kHalo; z < Z0 + (int)bz + kHalo; ++z)
        for (int y = Y0 - kHalo; y < Y0 + (int)by + kHalo; ++y) {
          if (y < 0 || y >= dim1 || z < 0 || z >= dim2) continue;
          const uint64_t row = ((uint64_t)z * dim1 + (uint64_t)y) * dim0;
          staged += cell_start[row + cl(X0 + (int)bx + XH)] - cell_start[row + cl(X0 - XH)];
        }
    if (staged > cap) lost = q;
    occupied = q != 0;
  }
  // Atomics on one address are served one after the other (~10 ns each): the workgroup, not the wave, appends its boxes and adds its sums.
  __shared__ uint32_t wave_occ[kBlock / 64];
  __shared__ uint32_t list_base;
  __shared__ unsigned long long wave_sums[kBlock / 64][4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t m = __builtin_amdgcn_ballot_w64(occupied != 0);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    q += shfl_xor_any(q, off); staged += shfl_xor_any(staged, off); lost += shfl_xor_any(lost, off);
  }
  if (lane == 0) {
    wave_occ[wave] = (uint32_t)__builtin_popcountll(m);
    wave_sums[wave][0] = q; wave_sums[wave][1] = staged; wave_sums[wave][2] = lost;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t[3] = {0, 0, 0};
    uint32_t occ = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { t[0] += wave_sums[w][0]; t[1] += wave_sums[w][1]; t[2] += wave_sums[w][2]; occ += wave_occ[w]; }
    if (t[0]) { atomicAdd(sums, t[0]); atomicAdd(sums + 1, t[1]); atomicAdd(sums + 2, t[2]); atomicAdd(sums + 3, (unsigned long long)occ); }
    list_base = list && occ ? atomicAdd(list_count, occ) : 0u;
  }
  if (list) {  // the boxes that hold a query, ascending within a workgroup: the launch list of the box search
    __syncthreads();
    uint32_t before = list_base;
    for (uint32_t w = 0; w < wave; ++w) before += wave_occ[w];
    if (occupied) list[before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = box;
  }
}

// ---- queries per box from the sorted cell numbers: one lane per point, runs of one box (the points of a row segment are consecutive) are
// counted across the wave and added by their first lane.  Cell number -> (x, y, z) -> box by reciprocal multiplication in f64
// (floor((a + 0.5) * (1 / d)) is exact for a < 2^51).
__global__ __launch_bounds__(kBlock) void knn_box_queries_kernel(const uint32_t* __restrict__ cells, uint32_t nf, uint32_t dim0, uint32_t dim1, double inv_plane, double inv_dim0,
                                                                 uint32_t bx, uint32_t by, uint32_t bz, double inv_bx, double inv_by, double inv_bz, uint32_t nbx, uint32_t nby,
                                                                 uint32_t* __restrict__ box_q) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t box = 0xFFFFFFFFu;
  if (i < nf) {
    const uint32_t c = cells[i];
    const uint32_t cz = (uint32_t)(((double)c + 0.5) * inv_plane);
    const uint32_t rem = c - cz * (dim0 * dim1);
    const uint32_t cy = (uint32_t)(((double)rem + 0.5) * inv_dim0);
    const uint32_t cx = rem - cy * dim0;
    const uint32_t Bx = (uint32_t)(((double)cx + 0.5) * inv_bx), By = (uint32_t)(((double)cy + 0.5) * inv_by), Bz = (uint32_t)(((double)cz + 0.5) * inv_bz);
    box = (Bz * nby + By) * nbx + Bx;
  }
  const uint32_t prev = (uint32_t)__shfl_up((int)box, 1, 64);
  const bool head = box != 0xFFFFFFFFu && (lane == 0 || prev != box);
  const uint64_t heads = __builtin_amdgcn_ballot_w64(head), valid = __builtin_amdgcn_ballot_w64(box != 0xFFFFFFFFu);
  if (head) {
    // the run ends before the next head above this lane, or with the wave's last valid lane
    const uint64_t above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
    const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : (uint32_t)(64 - __builtin_clzll(valid));
    atomicAdd(&box_q[box], end - lane);
  }
}

// ---- the boxes that hold a query, in box order (one thread per box; a wave appends its boxes with ONE atomic): clouds that are not a
// filled box (a surface in a 3-D grid: two thirds of the boxes are empty) launch a workgroup per listed box, not per box
__global__ __launch_bounds__(kBlock) void knn_box_list_kernel(const uint32_t* __restrict__ cell_start, GridParams g, uint32_t bx, uint32_t by, uint32_t bz, uint32_t nbx,
                                                              uint32_t nby, uint32_t n_boxes, uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  const uint32_t box = blockIdx.x * kBlock + threadIdx.x;
  bool has = false;
  if (box < n_boxes) {
    const int dim0 = (int)g.dim[0], dim1 = (int)g.dim[1], dim2 = (int)g.dim[2];
    const int X0 = (int)(box % nbx) * (int)bx, Y0 = (int)((box / nbx) % nby) * (int)by, Z0 = (int)(box / (nbx * nby)) * (int)bz;
    const uint32_t x0 = (uint32_t)X0, x1 = (uint32_t)(X0 + (int)bx > dim0 ? dim0 : X0 + (int)bx);
    for (int z = Z0; z < Z0 + (int)bz && z < dim2 && !has; ++z)
      for (int y = Y0; y < Y0 + (int)by && y < dim1; ++y) {
        const uint64_t row = ((uint64_t)z * dim1 + (uint64_t)y) * dim0;
        if (cell_start[row + x1] != cell_start[row + x0]) { has = true; break; }
      }
  }
  const uint64_t m = __builtin_amdgcn_ballot_w64(has);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u, first = (uint32_t)__builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == first) base = atomicAdd(count, (uint32_t)__builtin_popcountll(m));
  base = (uint32_t)__shfl((int)base, (int)first, 64);
  if (has) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = box;
}

}  // namespace

namespace pstk {

namespace {
// Which box kernel runs (PST_KNN_VAR, read once): "1" = the first form (f64 scan, packed f64 keys); letters = instances of the second form
// (f32 scan, 32-bit keys): threads per workgroup, staged-point capacity, f64 coordinates in LDS or read from global memory, candidates
// per scan step.  The default is the measured optimum.
struct TileVariant { int version; uint32_t threads, cap; bool p3lds; int batch; char tag; };
// volume_like: the cloud fills its (trimmed) box -- boxes are full and large ones pay (512 threads, 3000 staged points, two workgroups per
// CU); otherwise (surfaces, strips: most of a box's cells are empty and the 31-cell / 64-row limits of a box bind before its capacity)
// 256 threads and 1536 points, four workgroups per CU.  Same-box A/B at 10^8 points: uniform cloud 35.4 (D) / 37.0 (G) / 36.8 (B) ms per
// call, LiDAR-like sheet 96 (D) / 80.7 (G) / 86.4 (B); the first form: 41.5 / 88.9.
const TileVariant& tile_variant(uint32_t k, bool volume_like) {
  static const TileVariant v1{1, 256, 1536, true, 4, '1'}, vB{2, 256, 2044, false, 4, 'B'}, vD{2, 512, 3000, false, 4, 'D'}, vG{2, 256, 1536, false, 4, 'G'};
  switch (box_kernel_for(k, volume_like, knn_tuning())) {
    case 'B': return vB;
    case 'D': return vD;
    case 'G': return vG;
    default: return v1;
  }
}

bool tile_fits(const pstn::GridParams& g, uint32_t bx, uint32_t by, uint32_t bz) {
  const uint32_t hx = bx + 2 * (g.rx + 1), hy = by + 2 * kHalo, hz = bz + 2 * kHalo;
  return bx <= g.dim[0] && by <= g.dim[1] && bz <= g.dim[2] && hx <= (uint32_t)kMaxRowCells && by * bz <= (uint32_t)kMaxQRows &&
         hy * hz <= (uint32_t)kMaxRows && (hx + 1) * hy * hz <= (uint32_t)kMaxDir;
}
// least geometric halo amplification among the boxes whose halo has at most `max_halo_cells` fine cells
bool best_box(const pstn::GridParams& g, double max_halo_cells, pstk::TileShape& t) {
  const uint32_t xh = g.rx + 1;
  double best_amp = 1e300;
  bool found = false;
  for (uint32_t bz = 1; bz <= 8; ++bz)
    for (uint32_t by = 1; by <= 8; ++by)
      for (uint32_t bx = 1; bx <= (uint32_t)kMaxRowCells; ++bx) {
        if (!tile_fits(g, bx, by, bz)) continue;
        // the halo is addressed unclipped (rows and cells outside the grid are empty), only its POINTS are clipped by the grid
        const uint32_t px = std::min(bx + 2 * xh, g.dim[0]), py = std::min(by + 2 * kHalo, g.dim[1]), pz = std::min(bz + 2 * kHalo, g.dim[2]);
        if ((double)px * py * pz > max_halo_cells) continue;
        const double amp = (double)(px * py * pz) / (double)(bx * by * bz);
        if (amp < best_amp - 1e-12) { best_amp = amp; t.bx = bx; t.by = by; t.bz = bz; found = true; }
      }
  return found;
}
}  // namespace

// Density probe (knn_probe_kernel): mean number of points within h / 2 and within h of a sampled point.  false on a HIP failure.
bool knn_probe(const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t nf, unsigned long long* scratch3, hipStream_t stream,
               double& mean_half, double& mean_full) {
  const uint32_t samples = 1u << 16;
  const uint32_t stride = std::max<uint32_t>(1u, nf / samples);
  const uint32_t n_s = (nf + stride - 1) / stride;
  if (hipMemsetAsync(scratch3, 0, 32, stream) != hipSuccess) return false;
  hipLaunchKernelGGL(knn_probe_kernel, dim3((n_s + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sxyz, cell_start, g, nf, stride, scratch3);
  unsigned long long h[3] = {};
  if (hipMemcpyAsync(h, scratch3, 24, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return false;
  if (!h[2]) return false;
  mean_half = (double)h[0] / (double)h[2];
  mean_full = (double)h[1] / (double)h[2];
  return true;
}

// Picks the box by MEASUREMENT: candidates from large to small (least geometric halo amplification for a shrinking halo budget); the first
// whose census (knn_census_kernel: staged points and queries of every box, read off the directory) loses at most 2 % of the queries to boxes
// that exceed the LDS capacity wins.  Average density would do for a cloud that fills its bounding box; a surface in a 3-D box puts all
// its points into a few boxes.  bx counts FINE cells along x (edge h / rx), by and bz rows (edge h).
bool knn_tile_shape(const pstn::GridParams& g, uint64_t nf, uint64_t cells, uint32_t k, bool volume_like, const uint32_t* cell_start,
                    unsigned long long* scratch3, hipStream_t stream, TileShape& t, BoxListSink* sink) {
  if (k > 64 || cells == 0 || nf == 0) return false;
  const TileVariant& var = tile_variant(k, volume_like);  // (the second form is instantiated for k <= 16; larger k: the first form)
  const KnnTuning& tune = knn_tuning();
  t.threads = var.threads;
  t.cap = var.cap;
  t.tag = var.tag;
  if (tune.tile[0] && tile_fits(g, tune.tile[0], tune.tile[1], tune.tile[2])) { t.bx = tune.tile[0]; t.by = tune.tile[1]; t.bz = tune.tile[2]; return true; }  // PST_KNN_TILE
  const double rho = (double)nf / (double)cells;  // points per fine cell, averaged over the whole grid: the first guess
  double budget = 0.90 * (double)t.cap / rho;
  TileShape last{};
  double shrink = 1.0;
  const bool debug = tune.debug;
  // census of one shape: h[0] queries, h[1] staged points, h[2] queries lost to boxes over capacity, h[3] non-empty boxes
  // (with a sink: every census also lists the boxes that hold a query, in a buffer of its own; the caller launches from the list of the
  //  shape that wins -- h[3] is its length)
  const uint32_t* last_list = nullptr;
  auto census = [&](const TileShape& c, unsigned long long (&h)[4]) -> int {
    const uint32_t nbx = (g.dim[0] + c.bx - 1) / c.bx, nby = (g.dim[1] + c.by - 1) / c.by, nbz = (g.dim[2] + c.bz - 1) / c.bz;
    const uint64_t n_boxes = (uint64_t)nbx * nby * nbz;
    if (n_boxes >= 0x7FFFFFFFull) return 1;
    if (hipMemsetAsync(scratch3, 0, 32, stream) != hipSuccess) return -1;
    uint32_t* list = nullptr;
    uint32_t* box_q = nullptr;
    // a grid far larger than the cloud (a surface in a 3-D box): the queries of the boxes are counted from the points' sorted cell numbers
    const bool from_points = sink && sink->sorted_cells && cells > 3 * nf && nf < 0xFFFFFFF0ull && (uint64_t)g.dim[0] * g.dim[1] < 0xFFFFFFFFull;
    if (sink) {
      list = sink->alloc((size_t)n_boxes * (from_points ? 8 : 4));
      if (!list || hipMemsetAsync(sink->count_dev, 0, 4, stream) != hipSuccess) return -1;
      if (from_points) {
        box_q = list + n_boxes;
        if (hipMemsetAsync(box_q, 0, (size_t)n_boxes * 4, stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(knn_box_queries_kernel, dim3((unsigned)((nf + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sink->sorted_cells, (uint32_t)nf, g.dim[0], g.dim[1],
                           1.0 / ((double)g.dim[0] * (double)g.dim[1]), 1.0 / (double)g.dim[0], c.bx, c.by, c.bz, 1.0 / (double)c.bx, 1.0 / (double)c.by, 1.0 / (double)c.bz,
                           nbx, nby, box_q);
      }
    }
    last_list = list;
    hipLaunchKernelGGL(knn_census_kernel, dim3((unsigned)((n_boxes + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, cell_start, g, c.bx, c.by, c.bz, nbx, nby,
                       (uint32_t)n_boxes, t.cap, scratch3, list, sink ? sink->count_dev : nullptr, (const uint32_t*)box_q);
    if (hipMemcpyAsync(h, scratch3, 32, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return -1;
    if (debug)
      fprintf(stderr, "[pst knn census] box %ux%ux%u: halo amplification %.2f, %.2f %% of the queries in boxes over capacity, %.0f queries per occupied box\n", c.bx, c.by,
              c.bz, h[0] ? (double)h[1] / (double)h[0] : 0.0, h[0] ? 100.0 * (double)h[2] / (double)h[0] : 0.0, h[3] ? (double)h[0] / (double)h[3] : 0.0);
    return 0;
  };
  for (int attempt = 0; attempt < 14; ++attempt) {
    budget *= shrink;
    TileShape c = t;
    if (!best_box(g, budget, c)) break;
    if (c.bx == last.bx && c.by == last.by && c.bz == last.bz) continue;
    last = c;
    unsigned long long h[4] = {};
    const int rc = census(c, h);
    if (rc < 0) return false;
    if (rc > 0) break;
    if (h[0] && (double)h[2] <= 0.02 * (double)h[0]) {
      t.bx = c.bx; t.by = c.by; t.bz = c.bz;
      if (sink) { sink->list = last_list; sink->n = (uint32_t)h[3]; }
      // ROUNDS.  The queries of a box are handed to the workgroup's waves in chunks of 64, so a box of Q queries keeps its LDS for
      // ceil(Q / threads) rounds, and in the last round most waves have left: 560 queries on 256 threads use 8.75 of 12 wave slots, and
      // no other workgroup can take the idle ones while the box holds its LDS.  If the typical box (mean + two standard deviations of
      // a Poisson count) needs r rounds and fills less than 85 % of them, the box is shortened along x to what r - 1 rounds hold --
      // unless that costs more than a quarter of its length (more halo per query, more workgroups).
      if (tune.rounds && h[3]) {
        const uint32_t bx2 = box_length_for_whole_rounds(t.bx, (double)h[0] / (double)h[3], t.threads);  // (normals_plan.hpp: ROUNDS)
        if (bx2 != t.bx) {
          TileShape c2 = t;
          c2.bx = bx2;
          unsigned long long h2[4] = {};
          if (tile_fits(g, c2.bx, c2.by, c2.bz) && census(c2, h2) == 0 && h2[0] && (double)h2[2] <= 0.02 * (double)h2[0]) {
            t.bx = bx2;
            if (sink) { sink->list = last_list; sink->n = (uint32_t)h2[3]; }
          }
        }
      }
      return true;
    }
    // most queries lost: the cloud is locally several times denser than its grid average (a surface): big steps down; otherwise fine ones
    shrink = h[0] && (double)h[2] > 0.5 * (double)h[0] ? 0.6 : 0.85;
  }
  return false;
}

// The boxes of shape `t` that hold at least one query, ascending, into list[0 .. return value); count_dev = one device word.  Returns
// 0xFFFFFFFF on a HIP failure.  (synchronises the stream)
uint32_t knn_box_list(const TileShape& t, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t* list, uint32_t* count_dev, hipStream_t stream) {
  const uint32_t nbx = (g.dim[0] + t.bx - 1) / t.bx, nby = (g.dim[1] + t.by - 1) / t.by, nbz = (g.dim[2] + t.bz - 1) / t.bz;
  const uint32_t n_boxes = nbx * nby * nbz;
  if (hipMemsetAsync(count_dev, 0, 4, stream) != hipSuccess) return 0xFFFFFFFFu;
  hipLaunchKernelGGL(knn_box_list_kernel, dim3((n_boxes + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cell_start, g, t.bx, t.by, t.bz, nbx, nby, n_boxes, list, count_dev);
  uint32_t n = 0;
  if (hipMemcpyAsync(&n, count_dev, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return 0xFFFFFFFFu;
  return n;
}
// the same list, its length left in *count_dev (stream-ordered replay of a plan: no read-back)
bool knn_box_list_async(const TileShape& t, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t* list, uint32_t* count_dev, hipStream_t stream) {
  const uint32_t nbx = (g.dim[0] + t.bx - 1) / t.bx, nby = (g.dim[1] + t.by - 1) / t.by, nbz = (g.dim[2] + t.bz - 1) / t.bz;
  const uint32_t n_boxes = nbx * nby * nbz;
  if (hipMemsetAsync(count_dev, 0, 4, stream) != hipSuccess) return false;
  hipLaunchKernelGGL(knn_box_list_kernel, dim3((n_boxes + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cell_start, g, t.bx, t.by, t.bz, nbx, nby, n_boxes, list, count_dev);
  return hipGetLastError() == hipSuccess;
}
uint32_t knn_box_count(const TileShape& t, const pstn::GridParams& g) {
  return ((g.dim[0] + t.bx - 1) / t.bx) * ((g.dim[1] + t.by - 1) / t.by) * ((g.dim[2] + t.bz - 1) / t.bz);
}

void launch_knn_tile(const TileShape& t, const double* sxyz, const uint32_t* cell_start, const pstn::GridParams& g, uint32_t k, uint32_t nf,
                     const pstn::RecOut& out, uint32_t* fb_list, uint32_t* fb_count, const uint32_t* box_list, uint32_t n_list, hipStream_t stream,
                     const uint32_t* n_list_dev) {
  TileArgs a{};
  a.sxyz = sxyz; a.cell_start = cell_start; a.g = g;
  a.bx = t.bx; a.by = t.by; a.bz = t.bz;
  a.nbx = (g.dim[0] + t.bx - 1) / t.bx; a.nby = (g.dim[1] + t.by - 1) / t.by;
  const uint32_t nbz = (g.dim[2] + t.bz - 1) / t.bz;
  a.n_boxes = box_list ? n_list : a.nbx * a.nby * nbz;
  a.box_list = box_list;
  a.n_boxes_dev = box_list ? n_list_dev : nullptr;
  a.k = k; a.nf = nf; a.out = out; a.fb_list = fb_list; a.fb_count = fb_count;
  a.ablate = knn_tuning().ablate;
  a.fit_guard = knn_tuning().fit_guard ? 1u : 0u;
  a.flush_at = knn_tuning().flush_at;
  // A-priori bound on the squared k-th distance: the grid's cell edge h was chosen as the radius of the sphere expected to hold about
  // 1.75 k points (normals.hip), and candidates beyond tau0 = h^2 (less a few ulps for the packed keys) are not even queued.  Without
  // it every candidate passes the "closer than the current k-th" test until a lane's list is full, and about k (1 + ln(N / k)) of N
  // candidates pass overall -- each pass is a sorted insertion the whole wave waits for.  The result stays exact: a query that finds
  // fewer than k candidates inside tau0 goes to the exact global-memory search like any other unfinished query (about 1 % of a uniform
  // cloud's queries), and tau0 < h^2 is also what makes 3 x 3 rows of cells enough.
  // (in a rotated frame the cell coordinates carry a rounding error of up to ~1e-9 h at the largest grids: ten times the margin there)
  a.tau0 = g.h * g.h * (1.0 - (g.rotated ? 4e-8 : 4e-9));
  a.tau0_below = std::nextafter(a.tau0, 0.0);
#ifdef PST_KNN_STATS
  static unsigned long long* dbg_dev = nullptr;
  {
    if (!dbg_dev) (void)hipMalloc((void**)&dbg_dev, 128);
    (void)hipMemsetAsync(dbg_dev, 0, 128, stream);
    a.dbg = dbg_dev;
  }
#endif
  const unsigned grid = (a.n_boxes + 7u) & ~7u;
  const bool knn = out.knn != nullptr || out.knn_u32 != nullptr;
  if (t.tag != '1') {
    // Second form.  Key units: f32 coordinates are (x - c) s with s^2 tau0 + eps = 2^21 - 3, so that every squared distance that passes the
    // scan's first limit converts to a 21-bit integer.  eps bounds |f32 squared distance - exact squared distance * s^2| for two points
    // of one box: coordinates are at most L sqrt(U) in magnitude (L = half diagonal of the halo box in units of h, plus one; U = s^2 h^2 ~
    // 2^21), so their f32 rounding error is at most half an ulp there, a coordinate difference (<= sqrt(U) for anything inside tau0) adds
    // half an ulp of its own, and the three products / fused adds of the squared distance 4 * 2^-24 relative.  Times 1.5 for safety.
    Tile2Args b{};
    const double hxh = (double)(t.bx + 2 * (g.rx + 1)) / (2.0 * (double)g.rx), hyh = 0.5 * (double)(t.by + 2 * kHalo), hzh = 0.5 * (double)(t.bz + 2 * kHalo);
    const double L = std::sqrt(hxh * hxh + hyh * hyh + hzh * hzh) + 1.0;
    const double U0 = 2097152.0, sU = std::sqrt(U0);
    auto ulp32 = [](double v) { const float f = (float)v; return (double)(std::nextafter(f, 3.0e38f) - f); };
    const double d_r = 0.5 * ulp32(L * sU * 1.01), d_dx = 2.0 * d_r + 0.5 * ulp32(sU * 1.01);
    const double eps = 1.5 * (2.0 * std::sqrt(3.0) * sU * 1.01 * d_dx + 3.0 * d_dx * d_dx + 4.0 * U0 / 16777216.0);
    const double T = a.tau0;                       // squared length in the cloud's units
    b.s2 = (2097149.0 - eps) / T;                  // tau0 * s2 + eps = 2^21 - 3
    b.s = std::sqrt(b.s2);
    b.s2 = b.s * b.s;
    b.eps = eps;
    b.lim0 = std::nextafter((float)(T * b.s2 + eps), 3.0e38f);
    b.kx = (float)(g.inv_hx / b.s);
    b.kyz = (float)(g.inv_h / b.s);
    {
      const double unit = t.cap > 2048 ? 2.0 : 1.0;  // 12 slot bits leave 20 for the distance: bins of two key units
      b.gap = (uint32_t)std::floor(1.0 + 2.0 * eps / unit) + 1u;
      const double fm = std::floor((T * b.s2 - eps) / unit) - 1.0;  // (F + 1) unit + eps < tau0 s2
      b.f_max = fm > 0.0 ? (uint32_t)fm : 0u;
    }
    b.t = a;
#define PST_TILE2_LAUNCH(KK, TT, CC, P3, BB, WW, FF)                                                                             \
    do {                                                                                                                       \
      if (knn && g.rotated) hipLaunchKernelGGL((knn_tile2_kernel<KK, TT, CC, P3, BB, WW, true, true, FF>), dim3(grid), dim3(TT), 0, stream, b);    \
      else if (knn) hipLaunchKernelGGL((knn_tile2_kernel<KK, TT, CC, P3, BB, WW, true, false, FF>), dim3(grid), dim3(TT), 0, stream, b);          \
      else if (g.rotated) hipLaunchKernelGGL((knn_tile2_kernel<KK, TT, CC, P3, BB, WW, false, true, FF>), dim3(grid), dim3(TT), 0, stream, b);    \
      else hipLaunchKernelGGL((knn_tile2_kernel<KK, TT, CC, P3, BB, WW, false, false, FF>), dim3(grid), dim3(TT), 0, stream, b);                  \
    } while (0)
#define PST_TILE2_K(TT, CC, P3, BB, WW)                                                                                        \
    do {                                                                                                                       \
      if (fit_rows) hipLaunchKernelGGL((knn_tile2_kernel<16, TT, CC, P3, BB, WW, false, false, 2>), dim3(grid), dim3(TT), 0, stream, b);                 \
      else if (fit_seq) { if (k <= 8) PST_TILE2_LAUNCH(8, TT, CC, P3, BB, WW, 0); else PST_TILE2_LAUNCH(16, TT, CC, P3, BB, WW, 0); }   \
      else if (k <= 8) PST_TILE2_LAUNCH(8, TT, CC, P3, BB, WW, 1);                                                             \
      else PST_TILE2_LAUNCH(16, TT, CC, P3, BB, WW, 1);                                                                        \
    } while (0)
    const bool fit_seq = t.fit_seq;
    const bool fit_rows = knn_tuning().fit == 2 && !fit_seq && !knn && !g.rotated;  // (PST_KNN_FIT=rows: the measured alternative, not a default)
    switch (t.tag) {
      case 'D': PST_TILE2_K(512, 3000, false, 4, 4); break;
      case 'G': PST_TILE2_K(256, 1536, false, 4, 4); break;
      default: PST_TILE2_K(256, 2044, false, 4, 3); break;
    }
#undef PST_TILE2_K
#undef PST_TILE2_LAUNCH
#ifdef PST_KNN_STATS
    {
      unsigned long long h[16] = {};
      (void)hipMemcpyAsync(h, a.dbg, 128, hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
      {
        const double tot = (double)(h[14] ? h[14] : 1);
        fprintf(stderr, "[pst knn tile2 %c] wave time (clock samples drain the LDS queue: indicative only): staging %.1f %%, chunk set-up %.1f %%, scan %.1f %%, insertion rounds %.1f %%, proof %.1f %%, fit + results %.1f %%, (other: chunk hand-out, exit) %.1f %%; %.0f kcycles per query wave\n",
                t.tag, 100.0 * h[8] / tot, 100.0 * h[9] / tot, 100.0 * h[10] / tot, 100.0 * h[11] / tot, 100.0 * h[12] / tot, 100.0 * h[13] / tot,
                100.0 * (tot - (double)(h[8] + h[9] + h[10] + h[11] + h[12] + h[13])) / tot, tot / 1000.0 / (double)(h[2] ? h[2] : 1));
      }
      fprintf(stderr, "[pst knn tile2 %c] boxes %u (%.1f query waves per box, %u waves per workgroup); insertion rounds %.2f per query wave\n", t.tag, a.n_boxes,
              (double)h[2] / (double)(a.n_boxes ? a.n_boxes : 1), t.threads / 64u, (double)h[7] / (double)(h[2] ? h[2] : 1));
      fprintf(stderr, "[pst knn tile2 %c] query waves %llu: scan steps %.1f, insertion steps %.1f per wave; slots tested %.1f, queued %.1f per query; %.4f %% failed the proof; %.4f %% handed to the exact search by the fit's conditioning guard; eps %.2f\n",
              t.tag, h[2], (double)h[0] / (double)(h[2] ? h[2] : 1), (double)h[1] / (double)(h[2] ? h[2] : 1), (double)h[3] / (double)nf, (double)h[4] / (double)nf,
              100.0 * (double)h[5] / (double)nf, 100.0 * (double)h[6] / (double)nf, eps);
    }
#endif
    return;
  }
#define PST_TILE_LAUNCH(KK, TT, CC)                                                                                          \
  do {                                                                                                                       \
    if (knn) hipLaunchKernelGGL((knn_tile_kernel<KK, TT, CC, true>), dim3(grid), dim3(TT), 0, stream, a);                   \
    else hipLaunchKernelGGL((knn_tile_kernel<KK, TT, CC, false>), dim3(grid), dim3(TT), 0, stream, a);                      \
  } while (0)
#define PST_TILE_K(TT, CC)                                                                                                   \
  do {                                                                                                                       \
    if (k <= 8) PST_TILE_LAUNCH(8, TT, CC);                                                                                  \
    else if (k <= 16) PST_TILE_LAUNCH(16, TT, CC);                                                                           \
    else if (k <= 24) PST_TILE_LAUNCH(24, TT, CC);                                                                           \
    else if (k <= 32) PST_TILE_LAUNCH(32, TT, CC);                                                                           \
    else PST_TILE_LAUNCH(64, TT, CC);                                                                                        \
  } while (0)
  PST_TILE_K(256, 1536);
#undef PST_TILE_K
#undef PST_TILE_LAUNCH
#ifdef PST_KNN_STATS
  {
    unsigned long long h[8] = {};
    (void)hipMemcpyAsync(h, a.dbg, 64, hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    fprintf(stderr, "[pst knn tile] query waves %llu: scan steps %.1f, insertion steps %.1f per wave; candidates tested %.1f, queued %.1f per query\n", h[2],
            (double)h[0] / (double)(h[2] ? h[2] : 1), (double)h[1] / (double)(h[2] ? h[2] : 1), (double)h[3] / (double)nf, (double)h[4] / (double)nf);
  }
#endif
}

}  // namespace pstk
