// Host-callable launchers of the gfx950 kernels (implemented in *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/pasture_amd.h"
#include "plan.h"

namespace pstk {

// K2/K3/K3' generic conversion.  src_aos / dst_aos select the interleaved arms of buffer_conversion.rs:418-662.
// use_lds: stage interleaved records through LDS tiles (plan.tile must be set); otherwise direct strided access.
// Returns false when the launch (or the plan upload) failed; inspect hipGetLastError().
// n_records: number of per-block AABB records written behind plan.h.bounds_partials (plans with fused bounds).
bool launch_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds, hipStream_t stream, unsigned* n_records = nullptr);
// upper bound of those records: a plan with fused bounds needs room for convert_max_records() records of 6 doubles (+ finalize room)
unsigned convert_max_records(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds);
// which kernel families (PST_PLAN_* bits) the calling thread's last conversion call launched
void reset_plan_kinds();
void note_plan_kind(uint32_t kind);
uint32_t plan_kinds();
// One note on stderr per process and `what` when a call of 2^20 points or more ends on a fall-back kernel family (the interpreter, the gather
// kernels): those run at 0.26-0.70 of peak where the plan-specialised families reach 0.75-0.84, and a caller who never asks
// pst_last_plan_kinds would not know.  PST_QUIET=1 silences it.
void note_slow_family(const char* what, uint64_t n_points, const char* why);
// compile (or fetch) the plan-specialised kernel this plan would take (jit.cpp); false + message when it cannot have one
bool prepare_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, std::string* error, bool* in_tree = nullptr);
bool launch_convert_fused_expressions(const ConvertPlan& plan, bool src_aos, bool dst_aos, hipStream_t stream, uint64_t* done, std::string* error);
bool convert_specialised_ready(const ConvertPlan& plan, bool src_aos, bool dst_aos);
size_t bounds_partials_bytes(unsigned n_records);
// fold n_records per-block {min xyz, max xyz} records into out6
void launch_finalize_bounds(double* partials, unsigned n_records, double* out6, hipStream_t stream);
// pst_bounds_record_set_form: records at these addresses leave the last fold kernel as {min, -max}
void set_bounds_record_form(const void* device_rec6, int form);
bool bounds_record_negates_max(const void* device_rec6);
int device_cus();
// Dynamic LDS bytes of a launch that needs `lds_bytes` and wants at most `resident` workgroups per CU (0 = whatever fits): the memory side of
// MI355X saturates with FEW bytes in flight per CU and loses throughput beyond that (profiles/r05_stream_sweeps.txt), so the streaming kernels
// cap their residency by asking for more LDS than they use.  PST_RESIDENT=<n> overrides every family's cap (same-box A/Bs; 0 = no cap).
// Never more than 64 KiB (no per-kernel attribute needed): caps below 2 are not expressible this way.
uint32_t lds_with_resident_cap(size_t lds_bytes, int resident);
constexpr int kResidentQuad = 0;          // plan-specialised conversion kernels (one-wave workgroups, 256-point tiles)
constexpr int kResidentFilterStream = 0;  // plan-specialised streaming compaction
constexpr int kResidentColumn = 4;        // columnar -> columnar conversion of one attribute (columns.hip) when the loads are the 16-byte side:
                                          // f64 -> f32 narrowing at 10^8 points 0.776 -> 0.801 of peak, 8 of 8 ABAB pairs (profiles/r05_abab.txt)

// K1/K2 fast path: columnar Vec3f64 stream. mode bits: 1 = affine, 2 = write dst, 4 = bounds.
// partials must hold stream_partials_bytes(); out6 receives {min xyz, max xyz} when bounds are requested.
int stream_grid();
size_t stream_partials_bytes(uint64_t n_points, unsigned mode);  // bytes `partials` must hold for this launch
void launch_vec3f64_stream(const double* src, double* dst, uint64_t n_points, const double scale[3], const double offset[3],
                           unsigned mode, double* partials, double* out6, hipStream_t stream);

// generic strided min/max over elements of `ncomp` components of component type `ct`.
// acc_f64: accumulate in f64 after a Rust `as` cast (calculate_bounds_from_custom_positions) with +/-f64::MAX seeds;
// otherwise accumulate in the component type with identity seeds.  out holds 2*ncomp accumulators {min.., max..}.
int reduce_grid();
size_t minmax_partials_bytes();
void launch_minmax(const uint8_t* base, uint64_t stride, uint64_t n, uint32_t ct, uint32_t ncomp, bool acc_f64, void* partials,
                   void* out, hipStream_t stream);

// compute_centroid (normal_estimation.rs:198-237): per-block records {sum xyz over all points, sum xyz over the finite points, finite count,
// NaN seen} of Vec3f64 values at base + e * stride; returns the number of records written to `partials` (centroid_partials_bytes())
size_t centroid_partials_bytes();
unsigned launch_centroid(const uint8_t* base, uint64_t stride, uint64_t n, double* partials, hipStream_t stream);

// deterministic synthetic fill of one attribute (see synth.hip)
struct SynthAttr {
  uint64_t base;    // address of the attribute of point 0
  uint64_t stride;  // bytes between points
  uint32_t size;    // attribute bytes
  uint32_t slot;    // index of the attribute in the layout
  uint32_t kind;    // PST_* datatype kind
  uint32_t special; // 0 generic, 1 Position3D, 2 LASLocalPosition, 3 mask 7, 4 mask 1
};
void launch_synth(const SynthAttr& a, uint64_t n, uint64_t seed, uint64_t first_index, hipStream_t stream);

// columnar -> columnar conversion of one attribute (columns.hip): e.src_col / e.dst_col are the range starts.
// With bounds_partials != nullptr (Vec3f64 target) the written values are folded into column_launch_grid() records.
unsigned column_launch_grid(const PlanEntry& e, uint64_t n, bool with_bounds);
bool launch_column(const PlanEntry& e, uint64_t n, double* bounds_partials, hipStream_t stream);

// K4 kNN normal estimation (normals.hip).  Positions: Vec3f64 at pos_base + i*pos_stride.  Outputs (all optional, device
// addresses): normals f64 [n][3], curvature f64 [n], knn int64 [n][k] and / or uint32 [n][k], NORMAL attribute (Vec3f32) and Curvature
// attribute (F64) of a target buffer.  Returns 0, -1 on a HIP failure, -2 beyond 2^32 - 16 points, or the number of neighbourhoods with
// < 3 usable points.
void release_normals_scratch();
struct KnnPlanRecord;  // normals_host.hpp
long long run_normals(const uint8_t* pos_base, uint64_t pos_stride, uint64_t n, uint32_t k, double* out_normals_dev, double* out_curv_dev,
                      long long* out_knn_dev, uint32_t* out_knn_u32_dev, uint64_t normal_attr, uint64_t normal_stride, uint64_t curv_attr,
                      uint64_t curv_stride, hipStream_t stream, KnnPlanRecord* record = nullptr);
// Stream-ordered replay of a recorded call (no host round trip, no allocation: hipGraph-capturable).  KnnPlan owns the scratch of the
// pipeline for its record's capacities; status2 = two device words: [0] = KNN_STATUS_* bits (0: the results are complete and exact),
// [1] = neighbourhoods with fewer than 3 usable points.
struct KnnPlan;
KnnPlan* knn_plan_create(const KnnPlanRecord& rec, bool packed_source, hipStream_t stream);
void knn_plan_free(KnnPlan* p);
const KnnPlanRecord& knn_plan_record(const KnnPlan* p);
// a plan made on a packed, 8-byte aligned Vec3f64 array searches its source in place and has no staging copy: it replays packed sources only;
// a plan made on any other storage owns a staging copy and replays both
bool knn_plan_accepts(const KnnPlan* p, const uint8_t* pos_base, uint64_t pos_stride);
bool run_normals_replay(KnnPlan* p, const uint8_t* pos_base, uint64_t pos_stride, double* out_normals_dev, double* out_curv_dev, uint32_t* out_knn_u32_dev,
                        uint64_t normal_attr, uint64_t normal_stride, uint64_t curv_attr, uint64_t curv_stride, unsigned long long* status2, hipStream_t stream);

// LAS record encoder (las_encode.hip)
uint32_t las_raw_record_size(int format);
size_t las_encode_workspace_bytes();
bool launch_las_encode(int format, const uint64_t* attr_base, const uint32_t* attr_stride, const uint32_t* attr_size, int n_attrs, bool interleaved,
                       uint64_t dst, uint64_t n,
                       const double scale[3], const double offset[3], const double bounds_in[6], uint32_t max_return, uint8_t* workspace,
                       double* out_bounds, unsigned long long* out_counts, hipStream_t stream);


// LAS record decoder (las_decode.hip)
unsigned las_decode_grid(uint64_t n);
bool launch_las_decode(int format, uint64_t src, uint64_t n, const uint64_t* dst_cols, int n_cols, const double scale[3], const double offset[3],
                       double* partials, hipStream_t stream);
unsigned las_decode_aos_grid(int format, uint64_t n);
bool launch_las_decode_aos(int format, uint64_t src, uint64_t dst, uint64_t n, const double scale[3], const double offset[3], double* partials,
                           hipStream_t stream);

// typed LAS points, columns <-> packed records (las_transpose.hip)
unsigned las_transpose_grid(uint64_t n);
bool launch_las_transpose(int format, bool to_records, uint64_t aos, const uint64_t* cols, int n_cols, uint64_t n, double* partials,
                          hipStream_t stream);

// predicate compaction (filter.hip)
size_t filter_workspace_bytes(uint64_t n);
uint32_t filter_tile(bool dst_aos, uint32_t dst_stride);
bool filter_record_tile_fits(uint32_t tile, uint32_t dst_stride);  // interleaved targets: do 16 records fit the LDS record tile?
void launch_filter_count(const uint8_t* mask_dev, uint64_t n, uint32_t tile, uint8_t* workspace, const unsigned long long** out_total_dev,
                         hipStream_t stream, unsigned long long* total_also = nullptr);
// A predicate fused into the streaming compaction kernel (round 6; expr.cpp writes the text): filter's closure (point_buffer.rs:1064-1136) evaluated by
// the count pass on the columns it names and AGAIN by the scatter pass on the values it holds in registers anyway -- no byte mask in between.
struct FilterPredicate {
  struct Attr { int slot; const char* type_name; uint32_t ncomp; };  // an argument of pst_pred: layout slot, component type (C name), components (1 / 3)
  std::string function_text;  // `PstV3`, and `pst_pred(<the named attributes>, i, p0, p1, p2, p3)`
  std::vector<Attr> attrs;    // in pst_pred's parameter order
  const double* p[4] = {nullptr, nullptr, nullptr, nullptr};
};
// does a compaction of attributes of these sizes take the streaming kernel at all (bytes per point, record size)?  The fused predicate lives there only.
bool filter_predicate_streams(const uint32_t* size, int n_attrs, bool dst_aos, uint32_t dst_stride, bool dst_covered);
std::string filter_stream_source(const uint32_t* size, int n_attrs, bool dst_aos, uint32_t dst_stride, bool dst_covered, const FilterPredicate* pred);  // '' = no streaming kernel
uint32_t* filter_counts(uint8_t* workspace, uint64_t n, uint32_t tile);  // where the scan expects the per-tile counts
void launch_filter_scan(uint64_t n, uint32_t tile, uint8_t* workspace, const unsigned long long** out_total_dev, hipStream_t stream,
                        unsigned long long* total_also = nullptr);
// pred != nullptr: the full tiles through the streaming kernel with the predicate inside (compiled in the calling thread; false + *error when it has
// none), `mask_dev` then covers the ragged last tile only -- rebased so that mask_dev[i] is point i's byte -- and may be null when there is none
bool launch_filter_scatter(const uint8_t* mask_dev, uint64_t n, uint32_t tile, uint8_t* workspace, uint64_t limit, const uint64_t* src_addr,
                           const uint32_t* src_stride, const uint64_t* dst_addr, const uint32_t* dst_off, const uint32_t* size, int n_attrs,
                           bool dst_aos, uint64_t dst_aos_base, uint32_t dst_stride, bool dst_covered, hipStream_t stream,
                           const FilterPredicate* pred = nullptr, std::string* error = nullptr);


// voxel-grid down-sampling (voxel.hip)
enum : uint32_t { VX_AVG_VEC = 1, VX_AVG_NUM = 2, VX_MOST_COMMON = 3, VX_MOST_COMMON_BOOL = 4, VX_MAX_POOL = 5 };
enum : uint32_t { VX_STATUS_BOUNDS_INVALID = 1, VX_STATUS_MARKER_CAPACITY = 2, VX_STATUS_VOXEL_CAPACITY = 4, VX_STATUS_LEAF = 8 };
struct VoxelGridState;
// capacities a stream-ordered plan is built for (voxel_plan_create): points, key bits per axis (>= the bits the marker counts need),
// markers in total, occupied voxels, LDS points per 64-voxel group of the reduction, the leaf sizes
struct VoxelPlanShape { uint64_t n; uint32_t bits[3]; uint32_t cap_markers; uint64_t cap_voxels; uint32_t stage_cap; double leaf[3]; };
VoxelGridState* voxel_plan_create(const VoxelPlanShape& shape, hipStream_t stream);
bool voxel_grid_build_async(VoxelGridState* st, const uint8_t* pos_base, uint64_t pos_stride, const double* bounds6, unsigned long long* count_and_status,
                            hipStream_t stream);
long long voxel_grid_build(VoxelGridState*& st, const uint8_t* pos_base, uint64_t pos_stride, uint64_t n, const double* markers_x, uint32_t nx,
                           const double* markers_y, uint32_t ny, const double* markers_z, uint32_t nz, const double origin[3], const double leaf[3],
                           hipStream_t stream);
bool voxel_grid_reduce(VoxelGridState* st, const uint64_t* src_addr, const uint32_t* src_stride, const uint64_t* dst_addr, const uint32_t* dst_stride,
                       const uint32_t* reduce, const uint32_t* kind, int n_attrs, uint64_t dst_first, hipStream_t stream);
void voxel_grid_free(VoxelGridState* st);

}  // namespace pstk
