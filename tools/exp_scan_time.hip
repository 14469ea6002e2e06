#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device_sort.hpp"
int main() {
  for (size_t n : {size_t(1800000), size_t(30000000)}) {
    uint32_t* d; hipMalloc(&d, n * 4); hipMemset(d, 0x7F, n * 4);
    size_t b = 0; pstk::suffix_min_u32(nullptr, b, d, n, nullptr);
    void* tmp; hipMalloc(&tmp, b);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0); pstk::suffix_min_u32(tmp, b, d, n, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); printf("suffix_min n=%zu: %.3f ms\n", n, ms);
    }
    unsigned long long* o; hipMalloc(&o, n * 8);
    size_t b2 = 0; pstk::exclusive_sum_u32_u64(nullptr, b2, d, o, n, nullptr); void* t2; hipMalloc(&t2, b2);
    for (int r = 0; r < 2; ++r) {
      hipEventRecord(e0); pstk::exclusive_sum_u32_u64(t2, b2, d, o, n, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); printf("exclusive_sum n=%zu: %.3f ms\n", n, ms);
    }
  }
  return 0;
}
