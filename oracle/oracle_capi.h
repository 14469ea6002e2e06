/* ORACLE — TEST INFRASTRUCTURE ONLY (see pasture_oracle.hpp header).
 *
 * C API over the CPU restatement.  It deliberately has the SAME shape as the product's C ABI
 * (include/pasture_amd.h) with the prefix `orc_` instead of `pst_`, so that the parity tests can drive the
 * oracle and the HIP path through one harness.  All memory is host memory.
 */
#ifndef PASTURE_ORACLE_CAPI_H
#define PASTURE_ORACLE_CAPI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_layout orc_layout;
typedef struct orc_buffer orc_buffer;
typedef struct orc_converter orc_converter;

typedef struct orc_datatype {
  uint32_t kind; /* 0..17, declaration order of PointAttributeDataType (point_layout.rs:25-50) */
  uint32_t reserved;
  uint64_t size_param;  /* ByteArray length / Custom size */
  uint64_t align_param; /* Custom min_alignment */
  uint8_t uuid[16];     /* Custom name */
} orc_datatype;

typedef struct orc_member {
  const char* name;
  orc_datatype datatype;
  uint64_t offset;
  uint64_t size;
} orc_member;

typedef struct orc_transform {
  uint32_t kind; /* 0 none, 1 affine, 2 bitfield */
  uint32_t shift;
  orc_datatype datatype; /* the closure's T */
  double scale[3];
  double offset[3];
  uint64_t mask;
} orc_transform;

typedef struct orc_mapping_info {
  const char* source_name;
  const char* target_name;
  orc_datatype source_datatype;
  orc_datatype target_datatype;
  uint64_t source_offset;
  uint64_t target_offset;
  int32_t has_converter;
  uint32_t transform_kind;
  int32_t apply_to_source;
  int32_t reserved;
} orc_mapping_info;

const char* orc_last_error(void);

int orc_layout_create(orc_layout** out);
int orc_layout_destroy(orc_layout* l);
int orc_layout_clone(const orc_layout* l, orc_layout** out);
int orc_layout_add_attribute(orc_layout* l, const char* name, const orc_datatype* dt, uint32_t packed, uint64_t max_alignment);
int orc_layout_from_members(const orc_member* members, size_t n, uint64_t type_alignment, orc_layout** out);
int orc_layout_num_attributes(const orc_layout* l, size_t* out);
int orc_layout_get_member(const orc_layout* l, size_t index, orc_member* out);
int orc_layout_size_of_point_entry(const orc_layout* l, uint64_t* out);
int orc_layout_alignment(const orc_layout* l, uint64_t* out);
int orc_layout_equals(const orc_layout* a, const orc_layout* b, int* out);

/* storage: 0 = interleaved (VectorBuffer), 1 = columnar (HashMapBuffer). memkind ignored (host). */
int orc_buffer_create(const orc_layout* l, uint32_t storage, uint32_t memkind, orc_buffer** out);
int orc_buffer_destroy(orc_buffer* b);
int orc_buffer_len(const orc_buffer* b, size_t* out);
int orc_buffer_resize(orc_buffer* b, size_t count);
/* test infrastructure: the reference's per-point plane fit for neighbour lists given by the caller (points [n_points][3] f64, knn [n_queries][k] int64, < 0 ends a list) */
int orc_fit_neighbourhoods(const double* points_xyz, size_t n_points, const int64_t* knn, size_t n_queries, size_t k, double* out_normals, double* out_curvature);
int orc_buffer_swap(orc_buffer* b, size_t from_index, size_t to_index);  /* BorrowedMutBuffer::swap, point_buffer.rs:229, :770-783, :1276-1292 */
int orc_buffer_is_columnar(const orc_buffer* b, int* out);
int orc_buffer_layout(const orc_buffer* b, orc_layout** out_clone);
int orc_buffer_write_points(orc_buffer* b, size_t first, size_t count, const void* src);
int orc_buffer_read_points(const orc_buffer* b, size_t first, size_t count, void* dst);
int orc_buffer_write_attribute(orc_buffer* b, const char* name, const orc_datatype* dt, size_t first, size_t count, const void* src);
int orc_buffer_read_attribute(const orc_buffer* b, const char* name, const orc_datatype* dt, size_t first, size_t count, void* dst);
int orc_buffer_synth_fill(orc_buffer* b, uint64_t seed, uint64_t first_index);
/* OwningBufferExt::append, point_buffer.rs:419-489 */
int orc_buffer_append(orc_buffer* self, const orc_buffer* other);
/* HashMapBuffer::filter_into / filter, point_buffer.rs:1064-1136; predicate = mask[idx] != 0; mask_memkind ignored (host) */
int orc_buffer_filter_into(const orc_buffer* src, orc_buffer* dst, const uint8_t* mask, uint32_t mask_memkind, int64_t num_matches_hint,
                           size_t* out_matches);
int orc_buffer_filter(const orc_buffer* src, const uint8_t* mask, uint32_t mask_memkind, uint32_t out_storage, orc_buffer** out);

/* RawPointConverter::{from_to, convert}, attribute_conversion.rs:62-109 (point-major; same-datatype attributes are skipped) */
typedef struct orc_point_converter orc_point_converter;
int orc_point_converter_create(const orc_layout* from, const orc_layout* to, orc_point_converter** out);
int orc_point_converter_destroy(orc_point_converter* c);
int orc_point_converter_num_converters(const orc_point_converter* c, size_t* out);
int orc_point_converter_convert(const orc_point_converter* c, const orc_buffer* src, size_t src_first, orc_buffer* dst, size_t dst_first, size_t count);

int orc_converter_create(const orc_layout* from, const orc_layout* to, int with_default, orc_converter** out);
int orc_converter_destroy(orc_converter* c);
int orc_converter_set_custom_mapping(orc_converter* c, const char* from_name, const orc_datatype* from_dt, const char* to_name,
                                     const orc_datatype* to_dt);
int orc_converter_set_custom_mapping_with_transformation(orc_converter* c, const char* from_name, const orc_datatype* from_dt,
                                                         const char* to_name, const orc_datatype* to_dt, const orc_transform* xf,
                                                         int apply_to_source);
int orc_converter_num_mappings(const orc_converter* c, size_t* out);
int orc_converter_get_mapping(const orc_converter* c, size_t index, orc_mapping_info* out);
int orc_converter_convert_into_range(const orc_converter* c, orc_buffer* src, size_t s0, size_t s1, orc_buffer* dst, size_t t0, size_t t1);
int orc_converter_convert(const orc_converter* c, orc_buffer* src, uint32_t out_storage, orc_buffer** out);

int orc_calculate_bounds(const orc_buffer* b, double out_min[3], double out_max[3], int* has_value);
int orc_minmax_attribute(const orc_buffer* b, const char* name, const orc_datatype* dt, void* out_min, void* out_max, int* has_value);
int orc_transform_attribute(orc_buffer* b, const char* name, const orc_datatype* dt, const orc_transform* xf);
int orc_compute_normals(const orc_buffer* b, size_t k, double* out_normals, double* out_curvature, int64_t* out_knn);
/* compute_centroid, normal_estimation.rs:198-237 */
int orc_compute_centroid(const orc_buffer* b, double out_centroid[3]);
/* SliceBuffer::slice, slice.rs:16-43; view_attribute_with_conversion, point_buffer.rs:322-330 */
int orc_buffer_slice(const orc_buffer* parent, size_t first, size_t count, orc_buffer** out);
int orc_buffer_read_attribute_converted(const orc_buffer* b, const char* name, const orc_datatype* target_dt, size_t first, size_t count, void* dst);

/* voxelgrid_filter (pasture-algorithms/src/voxel_grid.rs:109-166): appends one centroid point per occupied voxel to `filtered` */
int orc_voxelgrid_filter(const orc_buffer* buffer, double leafsize_x, double leafsize_y, double leafsize_z, orc_buffer* filtered);

/* RawLASWriter::write_points_default_layout (pasture-io/src/las/raw_writers.rs:203-363); same contract as pst_las_encode_points */
int orc_las_encode_points(const orc_buffer* src, uint32_t point_format, const double scale[3], const double offset[3], orc_buffer* dst,
                          size_t dst_first, double bounds_inout[6], uint64_t points_by_return[15], uint32_t max_return);

/* single value through the `as` table (attribute_conversion.rs:184-271); ERR_INVALID_CONVERSION if unlisted */
int orc_as_convert(uint32_t from_kind, uint32_t to_kind, const void* in, void* out);
/* helpers of normal_estimation.rs exposed for the known-answer tests (:503-550) */
int orc_covariance(const double* points_xyz, size_t n, double out_centroid[3], double out_cov_rowmajor[9], int* ok);
int orc_plane_parameter(const double cov_rowmajor[9], double out_normal[3], double* out_curvature);
uint64_t orc_align_to(uint64_t v, uint64_t alignment);

/* Timed CPU baseline, BASELINE.json configs[0]: n synthetic XYZ f64 points in a VectorBuffer [POSITION_3D] ->
 * HashMapBuffer via BufferLayoutConverter::for_layouts + calculate_bounds, `reps` runs; writes per-run seconds. */
int orc_bench_config1(size_t n, int reps, uint64_t seed, double* out_seconds, double out_bounds[6]);
/* Timed CPU baseline for the bench.py N=1 workload: columnar POSITION_3D -> columnar POSITION_3D with an affine
 * transformation + calculate_bounds on the result (configs[1] on the CPU path). */
int orc_bench_config2(size_t n, int reps, uint64_t seed, const double scale[3], const double offset[3], double* out_seconds,
                      double out_bounds[6]);

#ifdef __cplusplus
}
#endif
#endif
