// In-tree instantiations of the plan-specialised conversion kernels (jit_quad.hpp) for the layouts of the reference's own benches and
// tests (pasture-core/src/test_utils.rs:19-31, benches/layout_conversion_bench.rs:15-39, buffer_filter_bench.rs): a warm cache in front of the
// run-time compiler (jit.cpp), and the build-time check that jit_quad.hpp compiles with hipcc.  static_plans.inc is GENERATED from the same
// generator the run-time compiler uses (tools/gen_static_plans.py): a run-time plan takes an in-tree kernel iff the translation unit jit.cpp
// would compile for it equals the recorded text, character for character.  PST_STATIC_PLANS=0 sends every plan to the run-time compiler.
#include <cstdlib>
#include <cstring>
#include <string>

#include "jit_quad.hpp"
#include "kernels.hpp"

#include "static_plans.inc"

namespace {

template <typename P>
__global__ __launch_bounds__(P::blk) void quad_convert_static_kernel(const ConvertHeader h, const PlanEntry* __restrict__ entries) {
  pstq::quad_convert_body<P>(h, entries);
}

template <typename P>
void launch_plan(unsigned grid, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries) {
  const uint32_t lds = pstk::lds_with_resident_cap(pstq::quad_lds_bytes<P>(), pstk::kResidentQuad);
  auto kfn = quad_convert_static_kernel<P>;
  if (lds > 64u * 1024u) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(P::blk), lds, stream, h, entries);
}

}  // namespace

namespace pstk {

// The in-tree kernel for the plan whose generated translation unit is `source`: launches it over h.n points (a multiple of *tile) and
// returns true; false when no in-tree plan has that text.  tile_only: just report the tile.
bool launch_convert_static(const std::string& source, uint32_t* tile, bool tile_only, unsigned grid, const ConvertHeader& h, const PlanEntry* entries,
                           hipStream_t stream) {
  static const bool enabled = [] { const char* v = std::getenv("PST_STATIC_PLANS"); return !(v && *v == '0'); }();
  if (!enabled) return false;
#define PST_TRY_PLAN(NAME)                                             \
  if (source == k##NAME##Source) {                                     \
    *tile = 4u * (uint32_t)NAME::blk;                                  \
    if (!tile_only) launch_plan<NAME>(grid, stream, h, entries);       \
    return true;                                                       \
  }
  PST_STATIC_PLANS(PST_TRY_PLAN)
#undef PST_TRY_PLAN
  return false;
}

}  // namespace pstk
