// Sorts and scans shared by the spatial-index builders (normals.hip, voxel.hip): the library's own kernels (radix_sort.hip).  Every function follows
// the two-call convention: tmp == nullptr writes the scratch size to `bytes` and does nothing else.  All of them are stream-ordered; none synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pstk {

// stable LSD radix sort of (key, value) pairs on key bits [0, end_bit); n < 2^32.  BOTH pairs of buffers are scratch (the passes
// alternate between them); the result is in (keys_out, vals_out).  radix_sort.hip: one, three or four passes for 32-bit keys.
// iota: the values are the element numbers 0 .. n-1 -- vals_in is not read (it stays scratch).  first: what sort_first_pass returned for the
// same (tmp, n, end_bit) when the caller's key kernel has left the first pass's histogram there, else nullptr.
struct RadixFirstPass { uint32_t* counts; uint32_t tiles, bits, tile_size; };  // counts == nullptr: not offered (the library sort is in use)
hipError_t sort_pairs_u32(void* tmp, size_t& bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream, bool iota = false, const RadixFirstPass* first = nullptr);
RadixFirstPass sort_first_pass(void* tmp, size_t n, unsigned end_bit);
// radix_sort.hip
bool radix_sort_pairs_supported(size_t n, unsigned end_bit);
RadixFirstPass radix_sort_first_pass(void* tmp, size_t n, unsigned end_bit);
hipError_t radix_sort_pairs_u32(void* tmp, size_t& bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned end_bit,
                                hipStream_t stream, bool iota = false, bool first_hist_ready = false);
// 64-bit keys (fine voxel grids, Morton keys of the hash-grid kNN): ceil(end_bit / 9) passes; both pairs of buffers are scratch here too
hipError_t sort_pairs_u64(void* tmp, size_t& bytes, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
                          size_t n, unsigned end_bit, hipStream_t stream);
hipError_t radix_sort_pairs_u64(void* tmp, size_t& bytes, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned end_bit,
                                hipStream_t stream);
// exclusive prefix sum of u32 counts into u64 offsets (out[0] = 0)
hipError_t exclusive_sum_u32_u64(void* tmp, size_t& bytes, const uint32_t* in, unsigned long long* out, size_t n, hipStream_t stream);

// in place: data[i] = min(data[i], data[i + 1], ..., data[n - 1])  (a directory whose run heads were scattered into a 0xFFFFFFFF-filled array)
hipError_t suffix_min_u32(void* tmp, size_t& bytes, uint32_t* data, size_t n, hipStream_t stream);

// stream-ordered scratch from the library's allocator (pst::dev_alloc / dev_free: HIP's stream-ordered pool, or plain hipMalloc
// when the device has no pool support or PST_NO_POOL is set); freed in stream order by the destructor
// hipMalloc that, out of memory, hands the unused blocks of the stream-ordered pool back to the driver and asks once more (buffer.cpp)
hipError_t device_malloc_retry(void** p, size_t bytes);

struct DevBuf {
  void* p = nullptr;
  hipError_t alloc(size_t bytes, hipStream_t stream);
  void release();
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  template <typename T> T* as() { return (T*)p; }
};

}  // namespace pstk
