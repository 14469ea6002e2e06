// Standalone check of pasture_amd/csrc/radix_sort.hip against std::stable_sort (built and run on the GPU box):
//   hipcc -O2 --offload-arch=gfx950 -Ipasture_amd/csrc tools/test_radix_sort.hip pasture_amd/csrc/radix_sort.hip -o /tmp/test_radix && /tmp/test_radix
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "device_sort.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  std::mt19937_64 rng(7);
  const size_t sizes[] = {0, 1, 2, 63, 64, 65, 1000, 8191, 8192, 8193, 100003, 1 << 20, 3000001, 100000000};
  const unsigned bitsv[] = {1, 5, 9, 10, 17, 18, 25, 26, 27, 28, 31, 32};
  int fails = 0;
  for (size_t n : sizes) {
    for (unsigned bits : bitsv) {
      if (n == 100000000 && bits != 27 && bits != 31) continue;
      std::vector<uint32_t> k(n), v(n);
      const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);  // (bits = 32: full-range keys)
      // a mix: uniform keys, and (every third case) keys confined to few values so that equal keys abound
      const bool few = (n + bits) % 3 == 0;
      for (size_t i = 0; i < n; ++i) { k[i] = (uint32_t)rng() & mask; if (few) k[i] &= 0x1Fu; v[i] = (uint32_t)i; }
      const bool iota = (n + bits) % 2 == 1;  // the sort numbers the elements itself: the value buffer then holds garbage
      if (iota) for (size_t i = 0; i < n; ++i) v[i] = 0xDEADBEEFu;
      uint32_t *ka, *kb, *va, *vb;
      CK(hipMalloc(&ka, n * 4 + 16)); CK(hipMalloc(&kb, n * 4 + 16)); CK(hipMalloc(&va, n * 4 + 16)); CK(hipMalloc(&vb, n * 4 + 16));
      CK(hipMemcpy(ka, k.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
      size_t bytes = 0;
      CK(pstk::radix_sort_pairs_u32(nullptr, bytes, ka, kb, va, vb, n, bits, nullptr));
      void* tmp;
      CK(hipMalloc(&tmp, bytes));
      CK(pstk::radix_sort_pairs_u32(tmp, bytes, ka, kb, va, vb, n, bits, nullptr, iota));
      CK(hipDeviceSynchronize());
      double ms = 0;
      if (n >= (1 << 20)) {
        // timing: re-upload and sort three times
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemcpy(ka, k.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
          CK(hipEventRecord(e0)); CK(pstk::radix_sort_pairs_u32(tmp, bytes, ka, kb, va, vb, n, bits, nullptr, iota)); CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t);
        }
        ms = best;
      }
      std::vector<uint32_t> gk(n), gv(n);
      CK(hipMemcpy(gk.data(), kb, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gv.data(), vb, n * 4, hipMemcpyDeviceToHost));
      std::vector<uint32_t> order(n);
      std::iota(order.begin(), order.end(), 0u);
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
      bool ok = true;
      for (size_t i = 0; i < n && ok; ++i) ok = gv[i] == order[i] && gk[i] == k[order[i]];
      if (!ok) { ++fails; printf("MISMATCH n=%zu bits=%u few=%d\n", n, bits, (int)few); }
      else if (ms > 0) printf("ok n=%zu bits=%u %s%s: %.3f ms\n", n, bits, few ? "few" : "uniform", iota ? " (values numbered by the sort)" : "", ms);
      CK(hipFree(ka)); CK(hipFree(kb)); CK(hipFree(va)); CK(hipFree(vb)); CK(hipFree(tmp));
    }
  }
  printf("radix sort check (32-bit keys): %d failures\n", fails);
  // ---- round 6: 64-bit keys (ceil(bits / 9) passes over 4096-pair tiles) against std::stable_sort ----
  {
    const size_t sizes64[] = {0, 1, 63, 4095, 4096, 4097, 100003, 3000001, 50000000};
    const unsigned bits64[] = {9, 33, 36, 39, 45, 54, 63, 64};
    for (size_t n : sizes64) {
      for (unsigned bits : bits64) {
        if (n == 50000000 && bits != 39 && bits != 64) continue;
        std::vector<uint64_t> k(n);
        std::vector<uint32_t> v(n);
        const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        const bool few = (n + bits) % 3 == 0;
        for (size_t i = 0; i < n; ++i) { k[i] = rng() & mask; if (few) k[i] &= (0x1Full << (bits > 5 ? bits - 5 : 0)); v[i] = (uint32_t)(i * 2654435761u); }
        uint64_t *ka, *kb; uint32_t *va, *vb;
        CK(hipMalloc(&ka, n * 8 + 16)); CK(hipMalloc(&kb, n * 8 + 16)); CK(hipMalloc(&va, n * 4 + 16)); CK(hipMalloc(&vb, n * 4 + 16));
        CK(hipMemcpy(ka, k.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
        size_t bytes = 0;
        CK(pstk::radix_sort_pairs_u64(nullptr, bytes, ka, kb, va, vb, n, bits, nullptr));
        void* tmp; CK(hipMalloc(&tmp, bytes ? bytes : 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); CK(pstk::radix_sort_pairs_u64(tmp, bytes, ka, kb, va, vb, n, bits, nullptr)); CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> gk(n); std::vector<uint32_t> gv(n);
        CK(hipMemcpy(gk.data(), kb, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gv.data(), vb, n * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
        bool ok = true;
        for (size_t i = 0; i < n && ok; ++i) ok = gv[i] == v[order[i]] && gk[i] == k[order[i]];
        if (!ok) { ++fails; printf("MISMATCH (64-bit keys) n=%zu bits=%u few=%d\n", n, bits, (int)few); }
        else if (n >= 3000001) printf("ok 64-bit keys n=%zu bits=%u %s: %.3f ms\n", n, bits, few ? "few" : "uniform", ms);
        CK(hipFree(ka)); CK(hipFree(kb)); CK(hipFree(va)); CK(hipFree(vb)); CK(hipFree(tmp));
      }
    }
  }
  // ---- the two scans: exclusive sum u32 -> u64, suffix minimum in place ----
  {
    const size_t sizes_s[] = {0, 1, 2, 511, 512, 4095, 4096, 4097, 100003, 4096 * 512, 4096 * 512 + 1, 30000001};
    for (size_t n : sizes_s) {
      std::vector<uint32_t> in(n);
      for (size_t i = 0; i < n; ++i) in[i] = (uint32_t)rng() >> (i % 7 == 0 ? 0 : 12);
      uint32_t* d_in; unsigned long long* d_out;
      CK(hipMalloc(&d_in, n * 4 + 16)); CK(hipMalloc(&d_out, n * 8 + 16));
      CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
      size_t bytes = 0;
      CK(pstk::exclusive_sum_u32_u64(nullptr, bytes, d_in, d_out, n, nullptr));
      void* tmp; CK(hipMalloc(&tmp, bytes ? bytes : 16));
      CK(pstk::exclusive_sum_u32_u64(tmp, bytes, d_in, d_out, n, nullptr));
      std::vector<unsigned long long> got(n);
      CK(hipMemcpy(got.data(), d_out, n * 8, hipMemcpyDeviceToHost));
      unsigned long long run = 0; bool ok = true;
      for (size_t i = 0; i < n && ok; ++i) { ok = got[i] == run; run += in[i]; }
      if (!ok) { ++fails; printf("MISMATCH exclusive sum n=%zu\n", n); }
      size_t b2 = 0;
      CK(pstk::suffix_min_u32(nullptr, b2, d_in, n, nullptr));
      void* tmp2; CK(hipMalloc(&tmp2, b2 ? b2 : 16));
      CK(pstk::suffix_min_u32(tmp2, b2, d_in, n, nullptr));
      std::vector<uint32_t> gm(n);
      CK(hipMemcpy(gm.data(), d_in, n * 4, hipMemcpyDeviceToHost));
      uint32_t m = 0xFFFFFFFFu; ok = true;
      for (size_t i = n; i-- > 0 && ok;) { m = std::min(m, in[i]); ok = gm[i] == m; }
      if (!ok) { ++fails; printf("MISMATCH suffix minimum n=%zu\n", n); }
      CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(tmp)); CK(hipFree(tmp2));
    }
  }
  printf("radix sort + scans check: %d failures\n", fails);
  return fails != 0;
}
