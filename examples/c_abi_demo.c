/* Plain C against the drop-in boundary (include/pasture_amd.h, libpasture_amd.so): the doc-test of pasture's layout conversion
 * (pasture-core/src/layout/conversion/buffer_conversion.rs:98-110 style): an interleaved buffer of {POSITION_3D, INTENSITY}
 * points -> a columnar buffer with POSITION_3D affinely transformed, AABB of the result fused into the same pass.
 *
 *   gcc -std=c11 -I include examples/c_abi_demo.c -L pasture_amd -lpasture_amd -Wl,-rpath,$PWD/pasture_amd -o c_abi_demo
 *
 * Exit code 0 and "OK" on success; needs a gfx950 GPU (there is no CPU fallback: without one the first compute call fails
 * with PST_ERR_NO_DEVICE and the program says so). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pasture_amd.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != PST_OK) {                                                              \
      fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, pst_last_error());        \
      return rc_ == PST_ERR_NO_DEVICE ? 77 : 1;                                       \
    }                                                                                 \
  } while (0)

#pragma pack(push, 1)
typedef struct { double x, y, z; uint16_t intensity; } Point; /* repr(C, packed): 26 bytes */
#pragma pack(pop)

int main(void) {
  enum { N = 100000 };
  pst_datatype vec3f64, u16;
  memset(&vec3f64, 0, sizeof vec3f64);
  memset(&u16, 0, sizeof u16);
  vec3f64.kind = PST_VEC3F64;
  u16.kind = PST_U16;

  pst_layout* layout = NULL;
  CHECK(pst_layout_create(&layout));
  CHECK(pst_layout_add_attribute(layout, "Position3D", &vec3f64, /*packed*/ 1, 1));
  CHECK(pst_layout_add_attribute(layout, "Intensity", &u16, 1, 1));
  uint64_t size = 0;
  CHECK(pst_layout_size_of_point_entry(layout, &size));
  if (size != sizeof(Point)) { fprintf(stderr, "layout size %llu != %zu\n", (unsigned long long)size, sizeof(Point)); return 1; }

  Point* host = malloc(sizeof(Point) * N);
  for (int i = 0; i < N; ++i) { host[i].x = i; host[i].y = 2.0 * i; host[i].z = -0.5 * i; host[i].intensity = (uint16_t)(i & 0xFFFF); }

  pst_buffer *src = NULL, *dst = NULL;
  CHECK(pst_buffer_create(layout, PST_STORAGE_INTERLEAVED, PST_MEM_DEVICE, &src)); /* VectorBuffer */
  CHECK(pst_buffer_create(layout, PST_STORAGE_COLUMNAR, PST_MEM_DEVICE, &dst));    /* HashMapBuffer */
  CHECK(pst_buffer_resize(src, N));
  CHECK(pst_buffer_resize(dst, N));
  CHECK(pst_buffer_write_points(src, 0, N, host));

  pst_converter* conv = NULL;
  CHECK(pst_converter_create(layout, layout, 0, &conv));
  pst_transform xf;
  memset(&xf, 0, sizeof xf);
  xf.kind = PST_XF_AFFINE;
  xf.datatype = vec3f64;
  xf.scale[0] = 0.001; xf.scale[1] = 0.001; xf.scale[2] = 0.001;
  xf.offset[0] = 500000.0; xf.offset[1] = 5400000.0; xf.offset[2] = 100.0;
  CHECK(pst_converter_set_custom_mapping_with_transformation(conv, "Position3D", &vec3f64, "Position3D", &vec3f64, &xf, /*apply_to_source*/ 0));

  double mn[3], mx[3];
  int has = 0;
  CHECK(pst_converter_convert_into_range_with_bounds(conv, src, 0, N, dst, 0, N, mn, mx, &has));

  double* pos = malloc(sizeof(double) * 3 * N);
  uint16_t* inten = malloc(sizeof(uint16_t) * N);
  CHECK(pst_buffer_read_attribute(dst, "Position3D", &vec3f64, 0, N, pos));
  CHECK(pst_buffer_read_attribute(dst, "Intensity", &u16, 0, N, inten));
  for (int i = 0; i < N; ++i) {
    const double ex = (host[i].x * 0.001) + 500000.0, ey = (host[i].y * 0.001) + 5400000.0, ez = (host[i].z * 0.001) + 100.0;
    if (pos[3 * i] != ex || pos[3 * i + 1] != ey || pos[3 * i + 2] != ez || inten[i] != host[i].intensity) {
      fprintf(stderr, "mismatch at point %d\n", i);
      return 1;
    }
  }
  const double last = N - 1;
  if (!has || mn[0] != 500000.0 || mx[0] != (last * 0.001) + 500000.0 || mn[2] != (-0.5 * last * 0.001) + 100.0 || mx[2] != 100.0) {
    fprintf(stderr, "bounds mismatch: x [%.17g, %.17g] z [%.17g, %.17g]\n", mn[0], mx[0], mn[2], mx[2]);
    return 1;
  }
  /* round 4: a slice is a buffer (SliceBuffer::slice, slice.rs:16-43); compute_centroid; a converting attribute view */
  {
    pst_buffer* half = NULL;
    double hmn[3], hmx[3], c[3];
    int hhas = 0;
    CHECK(pst_buffer_slice(dst, N / 2, N - N / 2, &half));
    CHECK(pst_calculate_bounds(half, hmn, hmx, &hhas));
    if (!hhas || hmn[0] != ((double)(N / 2) * 0.001) + 500000.0 || hmx[0] != mx[0]) {
      fprintf(stderr, "slice bounds mismatch: x [%.17g, %.17g]\n", hmn[0], hmx[0]);
      return 1;
    }
    CHECK(pst_compute_centroid(dst, c));
    const double cx = ((0.5 * last) * 0.001) + 500000.0;
    if (c[0] < cx - 1e-6 || c[0] > cx + 1e-6) { fprintf(stderr, "centroid mismatch: %.17g\n", c[0]); return 1; }
    pst_datatype f32 = {0};
    f32.kind = PST_F32;
    float* as_f32 = malloc(sizeof(float) * N);
    CHECK(pst_buffer_read_attribute_converted(dst, "Intensity", &f32, 0, N, as_f32));  /* u16 -> f32, the Rust `as` table */
    for (int i = 0; i < N; ++i)
      if (as_f32[i] != (float)host[i].intensity) { fprintf(stderr, "converted view mismatch at %d\n", i); return 1; }
    free(as_f32);
    pst_buffer_destroy(half);
  }
  printf("OK: %d points converted on the device, bounds x [%.3f, %.3f]\n", N, mn[0], mx[0]);
  pst_converter_destroy(conv);
  pst_buffer_destroy(src);
  pst_buffer_destroy(dst);
  pst_layout_destroy(layout);
  free(host); free(pos); free(inten);
  return 0;
}
