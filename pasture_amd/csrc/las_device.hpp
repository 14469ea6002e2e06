// Compile-time facts about the LAS point formats shared by the encoder (las_encode.hip) and the decoder (las_decode.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace pstlas {

constexpr int kMaxAttrs = 24;

struct Fmt { bool ext, gps, color, nir, wave; };
__host__ __device__ constexpr Fmt fmt_of(int n) {
  return Fmt{n >= 6, n == 1 || n == 3 || n == 4 || n == 5 || n >= 6, n == 2 || n == 3 || n == 5 || n == 7 || n == 8 || n == 10, n == 8 || n == 10,
             n == 4 || n == 5 || n == 9 || n == 10};
}
__host__ __device__ constexpr uint32_t raw_size(Fmt f) {
  return (f.ext ? 30u : 20u) + (f.gps && !f.ext ? 8u : 0u) + (f.color ? 6u : 0u) + (f.nir ? 2u : 0u) + (f.wave ? 29u : 0u);
}

// LasPointFormatN::layout() is packed in field order (las_types.rs), so slot offsets are prefix sums of the field sizes.
__host__ __device__ constexpr uint32_t typed_slot_offset(Fmt f, int slot) {
  uint32_t sizes[kMaxAttrs] = {};
  int n = 0;
  sizes[n++] = 24; sizes[n++] = 2; sizes[n++] = 1; sizes[n++] = 1;
  if (f.ext) { sizes[n++] = 1; sizes[n++] = 1; }
  sizes[n++] = 1; sizes[n++] = 1; sizes[n++] = 1;
  if (f.ext) { sizes[n++] = 1; sizes[n++] = 2; } else { sizes[n++] = 1; sizes[n++] = 1; }
  sizes[n++] = 2;
  if (f.gps) sizes[n++] = 8;
  if (f.color) sizes[n++] = 6;
  if (f.nir) sizes[n++] = 2;
  if (f.wave) { sizes[n++] = 1; sizes[n++] = 8; sizes[n++] = 4; sizes[n++] = 4; sizes[n++] = 12; }
  uint32_t o = 0;
  for (int i = 0; i < slot && i < n; ++i) o += sizes[i];
  return o;
}
__host__ __device__ constexpr uint32_t typed_size(Fmt f) { return typed_slot_offset(f, kMaxAttrs); }


}  // namespace pstlas
