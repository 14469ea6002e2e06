// The sorts of the spatial-index builders behind one interface (device_sort.hpp).  Round 6: everything is the library's own code
// (radix_sort.hip: 32- and 64-bit keys, the exclusive sum, the suffix minimum); rocPRIM, which rounds 1-5 kept for 64-bit keys and the two scans, is gone.
#include <cstdlib>

#include "device_sort.hpp"

namespace pstk {

RadixFirstPass sort_first_pass(void* tmp, size_t n, unsigned end_bit) {
  static const bool no_fuse = [] { const char* e = std::getenv("PST_SORT_FUSE"); return e && *e == '0'; }();  // A/B: the sort counts its first histogram itself
  if (no_fuse || !radix_sort_pairs_supported(n, end_bit) || !tmp || n == 0) return RadixFirstPass{nullptr, 0, 0, 0};
  return radix_sort_first_pass(tmp, n, end_bit);
}

hipError_t sort_pairs_u32(void* tmp, size_t& bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream, bool iota, const RadixFirstPass* first) {
  if (!radix_sort_pairs_supported(n, end_bit)) return hipErrorInvalidValue;
  return radix_sort_pairs_u32(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, iota, first && first->counts);
}
hipError_t sort_pairs_u64(void* tmp, size_t& bytes, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream) {
  return radix_sort_pairs_u64(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream);
}

}  // namespace pstk
