// convert_tile_kernel<256, true, true> (interleaved -> interleaved) as its own translation unit; see convert.hip / convert_kernels.hpp.
#include "convert_kernels.hpp"

namespace pstk {

void launch_convert_tile_tt(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries) {
  auto kfn = convert_tile_kernel<256, true, true>;
  if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds_bytes, stream, h, entries);
}

}  // namespace pstk
