// Static-plan instantiations of the tile kernel and the matcher that routes a run-time plan to them (see static_plans.hpp).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "static_plans.hpp"

namespace pstk {

// A run-time plan takes a compile-time plan's kernel only if it equals it FIELD BY FIELD.  Not compared, on purpose: the tile (the static
// plan brings its own: plan_equals<P<.., T>> is the same predicate for every T), and scale / offset / mask / shift / xf_on_source -- they are
// read by the kernels only for entries with a transformation, and an entry with xf_kind != 0 never matches (below).
template <typename SP>
static bool plan_equals(const ConvertPlan& p) {
  const ConvertHeader& h = p.h;
  if (h.n_entries != (uint32_t)SP::n || h.src_stride != SP::src_stride || h.dst_stride != SP::dst_stride || h.quad != SP::quad ||
      h.dst_fully_covered != SP::covered || h.in_place || h.bounds_partials)
    return false;
  for (int m = 0; m < SP::n; ++m) {
    const PlanEntry& e = p.e[m];
    const StaticEntry s = SP::entry(m);
    if (e.src_off != s.src_off || e.dst_off != s.dst_off || e.src_size != s.src_size || e.dst_size != s.dst_size || e.ncomp != s.ncomp || e.src_ct != s.src_ct ||
        e.dst_ct != s.dst_ct || e.convert != s.convert || e.xf_kind != 0 || e.bounds != 0)
      return false;
    const bool shared = (p.masks[0] >> m) & 1u;
    uint32_t owner = 0;
    for (uint32_t w = 0; w < 16; ++w) if ((p.masks[1 + w] >> m) & 1u) owner = 1 + w;
    if ((shared ? 0u : owner) != s.owner) return false;
  }
  return true;
}

// the static plan brings its own tile (tuned per plan): grid and LDS size follow from it, not from the interpreter's choice
template <bool SRC_AOS, bool DST_AOS, typename SP>
static void launch_static(unsigned, size_t, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries) {
  const uint64_t n_tiles = (h.n + SP::tile - 1) / SP::tile;
  const unsigned grid = (unsigned)((std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, 1ull << 22)) + 7) / 8 * 8);
  size_t lds_bytes = 0;
  if (SRC_AOS) lds_bytes += ((size_t)SP::tile * SP::src_stride + 32 + 15) & ~(size_t)15;
  if (DST_AOS) lds_bytes += ((size_t)SP::tile * SP::dst_stride + 32 + 15) & ~(size_t)15;
  auto kfn = convert_tile_static_kernel<256, SRC_AOS, DST_AOS, SP>;
  if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds_bytes, stream, h, entries);
}

// true: the plan matched a compile-time plan and its kernel was launched
bool launch_convert_static(const ConvertPlan& plan, bool src_aos, bool dst_aos, unsigned grid, size_t lds_bytes, const PlanEntry* entries, hipStream_t stream) {
  static const bool enabled = [] { const char* v = std::getenv("PST_STATIC_PLANS"); return !(v && *v == '0'); }();
  if (!enabled) return false;
  if (!src_aos && dst_aos && plan_equals<pststatic::BigColumnsToRecords>(plan)) {
    launch_static<false, true, pststatic::BigColumnsToRecords>(grid, lds_bytes, stream, plan.h, entries);
    return true;
  }
  if (src_aos && !dst_aos && plan_equals<pststatic::BigRecordsToColumns>(plan)) {
    launch_static<true, false, pststatic::BigRecordsToColumns>(grid, lds_bytes, stream, plan.h, entries);
    return true;
  }
  if (src_aos && dst_aos && plan_equals<pststatic::Las1RecordsToXyzIC>(plan)) {
    launch_static<true, true, pststatic::Las1RecordsToXyzIC>(grid, lds_bytes, stream, plan.h, entries);
    return true;
  }
  {
    using namespace pststatic;
    // same-box A/B against the interpreter: records -> columns 0.762 -> 0.766, columns -> records 0.697 -> 0.742, records -> records 0.582 -> 0.716
    if (src_aos && !dst_aos && plan_equals<BenchSourceToTarget<0, 1024>>(plan)) { launch_static<true, false, BenchSourceToTarget<0, 1024>>(grid, lds_bytes, stream, plan.h, entries); return true; }
    if (!src_aos && dst_aos && plan_equals<BenchSourceToTarget<1, 1024>>(plan)) { launch_static<false, true, BenchSourceToTarget<1, 1024>>(grid, lds_bytes, stream, plan.h, entries); return true; }
    if (src_aos && dst_aos && plan_equals<BenchSourceToTarget<2, 1024>>(plan)) {
      launch_static<true, true, BenchSourceToTarget<2, 512>>(grid, lds_bytes, stream, plan.h, entries);  // same-box sweep: 256 0.564, 384 0.667, 512 0.716, 1024 0.572, 2048 0.370
      return true;
    }
  }
  if (std::getenv("PST_STATIC_DEBUG")) {
    const ConvertHeader& h = plan.h;
    fprintf(stderr, "[pst static] no match: aos %d->%d strides %u %u tile %u quad %u covered %u n %u masks %x %x %x %x %x\n", (int)src_aos, (int)dst_aos, h.src_stride,
            h.dst_stride, h.tile, h.quad, h.dst_fully_covered, h.n_entries, plan.masks[0], plan.masks[1], plan.masks[2], plan.masks[3], plan.masks[4]);
    for (uint32_t m = 0; m < h.n_entries; ++m)
      fprintf(stderr, "   e%u: off %u->%u size %u->%u ncomp %u ct %u->%u convert %u xf %u\n", m, plan.e[m].src_off, plan.e[m].dst_off, plan.e[m].src_size,
              plan.e[m].dst_size, plan.e[m].ncomp, plan.e[m].src_ct, plan.e[m].dst_ct, plan.e[m].convert, plan.e[m].xf_kind);
  }
  return false;
}

}  // namespace pstk
