// rocPRIM calls of the spatial-index builders, isolated in one translation unit (the headers are heavy).
#include <cstdlib>
#include <cstring>
#include <iterator>

#include <rocprim/rocprim.hpp>

#include "device_sort.hpp"

namespace pstk {

namespace {
bool library_sort_only() {
  static const bool lib_only = [] { const char* e = std::getenv("PST_SORT"); return e && std::strcmp(e, "rocprim") == 0; }();
  return lib_only;
}
__global__ void iota_kernel(uint32_t* __restrict__ v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
}  // namespace

RadixFirstPass sort_first_pass(void* tmp, size_t n, unsigned end_bit) {
  static const bool no_fuse = [] { const char* e = std::getenv("PST_SORT_FUSE"); return e && *e == '0'; }();  // A/B: the sort counts its first histogram itself
  if (no_fuse || library_sort_only() || !radix_sort_pairs_supported(n, end_bit) || !tmp || n == 0) return RadixFirstPass{nullptr, 0, 0, 0};
  return radix_sort_first_pass(tmp, n, end_bit);
}

hipError_t sort_pairs_u32(void* tmp, size_t& bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream, bool iota, const RadixFirstPass* first) {
  if (!library_sort_only() && radix_sort_pairs_supported(n, end_bit))
    return radix_sort_pairs_u32(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, iota, first && first->counts);
  if (tmp && iota && n) hipLaunchKernelGGL(iota_kernel, dim3(2048), dim3(256), 0, stream, vals_in, (uint64_t)n);
  return rocprim::radix_sort_pairs(tmp, bytes, (const uint32_t*)keys_in, keys_out, (const uint32_t*)vals_in, vals_out, n, 0u, end_bit, stream);
}
hipError_t sort_pairs_u64(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream) {
  return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
hipError_t exclusive_sum_u32_u64(void* tmp, size_t& bytes, const uint32_t* in, unsigned long long* out, size_t n, hipStream_t stream) {
  return rocprim::exclusive_scan(tmp, bytes, in, out, 0ull, n, rocprim::plus<unsigned long long>(), stream);
}
hipError_t suffix_min_u32(void* tmp, size_t& bytes, uint32_t* data, size_t n, hipStream_t stream) {
  auto rev = std::make_reverse_iterator(data + n);
  return rocprim::inclusive_scan(tmp, bytes, rev, rev, n, rocprim::minimum<uint32_t>(), stream);
}

}  // namespace pstk
