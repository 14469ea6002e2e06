// pasture-algorithms loops on the device: calculate_bounds, minmax_attribute, transform_attribute.
// Reference: pasture-algorithms/src/bounds.rs:11-85, minmax.rs:13-51, pasture-core/src/containers/point_buffer.rs:391-404.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "runtime.hpp"

namespace pst {

static bool position_convertible_to_vec3f64(const DataType& d) { return d.is_vec3(); }

// partials: scratch for the per-block records (bounds_partials_scratch_bytes); null = the calling thread's workspace for this (device,
// stream) -- which is created on first use, so a caller that must not allocate (a hipGraph capture runs on a stream of its own) brings its own
size_t bounds_partials_scratch_bytes(size_t count) { return std::max(pstk::stream_partials_bytes(count, 4u), pstk::minmax_partials_bytes()); }
void bounds_of_range(const pst_buffer& b, size_t first, size_t count, double* out6, hipStream_t stream, void* partials) {
  const Member* pm = b.layout.find_by_name("Position3D");
  if (!pm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "buffer has no Position3D attribute");
  const size_t slot = (size_t)(pm - b.layout.members.data());
  const DataType& dt = pm->def.datatype;
  // non-default position datatypes go through the `as` table (bounds.rs:56-85); impossible pairs panic in
  // get_generic_converter (attribute_conversion.rs:267-269)
  if (dt.kind != PST_VEC3F64 && !position_convertible_to_vec3f64(dt))
    throw Error(PST_ERR_INVALID_CONVERSION, "Invalid conversion " + dt.display() + " -> Vec3<f64>");
  const uint64_t base = b.columnar ? col_addr(b, slot, first) : aos_addr(b, first) + pm->offset;
  const uint64_t stride = b.columnar ? pm->size : b.layout.size;
  if (b.columnar && dt.kind == PST_VEC3F64 && base % 8 == 0) {
    // K1: coalesced 16-byte stream over the column
    pstk::launch_vec3f64_stream((const double*)(uintptr_t)base, nullptr, count, nullptr, nullptr, 4u,
                                (double*)(partials ? partials : workspace().partials(pstk::stream_partials_bytes(count, 4u))), out6, stream);
  } else {
    pstk::launch_minmax((const uint8_t*)(uintptr_t)base, stride, count, dt.comp_type(), 3, /*acc_f64=*/true,
                        partials ? partials : workspace().partials(pstk::minmax_partials_bytes()), out6, stream);
  }
  PST_HIP_CHECK(hipGetLastError());
}

void check_bounds_record(const double r[6], double out_min[3], double out_max[3]) {
  if (r[0] > r[3] || r[1] > r[4] || r[2] > r[5])
    throw Error(PST_ERR_BOUNDS_INVALID, "AABB::from_min_max: Minimum position must be <= maximum position!");
  for (int c = 0; c < 3; ++c) { out_min[c] = r[c]; out_max[c] = r[3 + c]; }
}

}  // namespace pst

using namespace pst;

extern "C" {

// calculate_bounds, bounds.rs:11-28
int pst_calculate_bounds(const pst_buffer* b, double out_min[3], double out_max[3], int* has_value) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(has_value, "has_value");
  if (b->len == 0 || !b->layout.find_by_name("Position3D")) {  // :12-21 => None
    *has_value = 0;
    return PST_OK;
  }
  Workspace& ws = workspace();
  hipStream_t s = current_stream();
  double* dev_rec = results_to_host() ? (double*)ws.pinned : (double*)(ws.dev + Workspace::kWorkspaceBytes - 64);
  bounds_of_range(*b, 0, b->len, dev_rec, s);
  if (!results_to_host()) PST_HIP_CHECK(hipMemcpyAsync(ws.pinned, dev_rec, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  stream_sync(s);
  check_bounds_record((const double*)ws.pinned, not_null(out_min, "out_min"), not_null(out_max, "out_max"));
  *has_value = 1;
  PST_API_END
}
int pst_calculate_bounds_async(const pst_buffer* b, double* device_out6) {
  PST_API_BEGIN
  not_null(b, "buffer");
  bounds_of_range(*b, 0, b->len, not_null(device_out6, "device_out6"), current_stream());
  PST_API_END
}

// minmax_attribute::<T>, minmax.rs:13-51, with T = the stored datatype (the only reachable configuration: the
// converting branch :39-47 always fails its own assertion, buffer_views.rs:548).
int pst_minmax_attribute(const pst_buffer* b, const char* name, const pst_datatype* dt, void* out_min, void* out_max, int* has_value) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(has_value, "has_value");
  AttributeDef def{not_null(name, "name"), DataType::from_c(dt)};
  if (!b->layout.find_by_name(def.name))  // :17-26
    throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute [" + def.name + ";" + def.datatype.display() + "] not contained in PointLayout buffer");
  if (!(def.datatype.is_scalar() || def.datatype.is_vec3()))  // no MinMax impl for Vec4u8 / ByteArray / Custom (math/minmax.rs)
    throw Error(PST_ERR_INVALID_ARGUMENT, "MinMax is not implemented for datatype " + def.datatype.display());
  const int slot = b->layout.index_of(def);  // view_attribute::<T>: exact (name, datatype) match, buffer_views.rs:301-310
  if (slot < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  if (b->len == 0) { *has_value = 0; return PST_OK; }
  const Member& m = b->layout.members[(size_t)slot];
  Workspace& ws = workspace();
  hipStream_t s = current_stream();
  const uint64_t base = b->columnar ? col_addr(*b, (size_t)slot, 0) : aos_addr(*b, 0) + m.offset;
  const uint64_t stride = b->columnar ? m.size : b->layout.size;
  const uint32_t ncomp = def.datatype.num_components();
  const size_t csize = m.size / ncomp;
  uint8_t* dev_rec = ws.dev + Workspace::kWorkspaceBytes - 64;  // 2 * ncomp * csize <= 48 bytes
  pstk::launch_minmax((const uint8_t*)(uintptr_t)base, stride, b->len, def.datatype.comp_type(), ncomp, /*acc_f64=*/false,
                      ws.partials(pstk::minmax_partials_bytes()), dev_rec, s);
  PST_HIP_CHECK(hipGetLastError());
  // record {min.., max..} + the FIRST value: a NaN first value seeds (NaN, NaN) and sticks (minmax.rs:30, math/minmax.rs:78-94)
  PST_HIP_CHECK(hipMemcpyAsync(ws.pinned, dev_rec, 2 * m.size, hipMemcpyDeviceToHost, s));
  PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 64, (const void*)(uintptr_t)base, m.size, hipMemcpyDeviceToHost, s));
  stream_sync(s);
  std::memcpy(not_null(out_min, "out_min"), ws.pinned, m.size);
  std::memcpy(not_null(out_max, "out_max"), ws.pinned + m.size, m.size);
  const CompType ct = def.datatype.comp_type();
  if (ct == CT_F32 || ct == CT_F64) {
    for (uint32_t c = 0; c < ncomp; ++c) {
      bool first_nan;
      if (ct == CT_F32) { float f; std::memcpy(&f, ws.pinned + 64 + c * 4, 4); first_nan = std::isnan(f); }
      else { double d; std::memcpy(&d, ws.pinned + 64 + c * 8, 8); first_nan = std::isnan(d); }
      if (first_nan) {
        std::memcpy((uint8_t*)out_min + c * csize, ws.pinned + 64 + c * csize, csize);
        std::memcpy((uint8_t*)out_max + c * csize, ws.pinned + 64 + c * csize, csize);
      }
    }
  }
  *has_value = 1;
  PST_API_END
}

// compute_centroid, normal_estimation.rs:198-237.  Panics: empty cloud ("The point cloud is empty!"); no Position3D of datatype Vec3f64
// (view_attribute::<Vector3<f64>>: exact match, buffer_views.rs:301-310).
int pst_compute_centroid(const pst_buffer* b, double out_centroid[3]) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(out_centroid, "out_centroid");
  if (b->len == 0) throw Error(PST_ERR_TOO_FEW_POINTS, "The point cloud is empty!");
  DataType v3; v3.kind = PST_VEC3F64;
  const int slot = b->layout.index_of(AttributeDef{"Position3D", v3});
  if (slot < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  ensure_device();
  const Member& m = b->layout.members[(size_t)slot];
  Workspace& ws = workspace();
  hipStream_t s = current_stream();
  const uint64_t base = b->columnar ? col_addr(*b, (size_t)slot, 0) : aos_addr(*b, 0) + m.offset;
  const uint64_t stride = b->columnar ? m.size : b->layout.size;
  double* partials = (double*)ws.partials(pstk::centroid_partials_bytes());
  const unsigned n_rec = pstk::launch_centroid((const uint8_t*)(uintptr_t)base, stride, b->len, partials, s);
  PST_HIP_CHECK(hipGetLastError());
  std::vector<double> h((size_t)n_rec * 8);
  PST_HIP_CHECK(hipMemcpyAsync(h.data(), partials, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  stream_sync(s);
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (unsigned r = 0; r < n_rec; ++r)
    for (int c = 0; c < 8; ++c) a[c] += h[(size_t)r * 8 + c];
  const bool dense = a[7] == 0.0;  // is_dense :133-140
  const double div = dense ? (double)b->len : a[6];  // (0 finite points: 0.0 / 0.0 = NaN, as in the reference)
  for (int c = 0; c < 3; ++c) out_centroid[c] = (dense ? a[c] : a[3 + c]) / div;
  PST_API_END
}

// transform_attribute(attribute, |_, v| f(v)), point_buffer.rs:391-404, with a closed-set transformation (in place).
int pst_transform_attribute(pst_buffer* b, const char* name, const pst_datatype* dt, const pst_transform* xf) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(xf, "transform");
  AttributeDef def{not_null(name, "name"), DataType::from_c(dt)};
  const DataType xt = DataType::from_c(&xf->datatype);
  if (xt != def.datatype)  // AttributeViewMut::new: assert_eq!(T::data_type(), attribute.datatype())
    throw Error(PST_ERR_TRANSFORM_TYPE_MISMATCH, "assertion `left == right` failed: T::data_type() is " + xt.display() + " but the attribute is " + def.datatype.display());
  const int slot = b->layout.index_of(def);
  if (slot < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  const uint32_t k = xt.kind;
  const bool ok = (xf->kind == PST_XF_AFFINE && (k == PST_F64 || k == PST_F32 || k == PST_VEC3F64 || k == PST_VEC3F32)) ||
                  (xf->kind == PST_XF_BITFIELD && (k == PST_U8 || k == PST_U16 || k == PST_U32 || k == PST_U64) && xf->shift < 64);
  if (!ok) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "Unsupported transformation descriptor for datatype " + xt.display());
  if (b->len == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  const Member& m = b->layout.members[(size_t)slot];
  PlanEntry e = identity_entry(m, m);
  e.xf_kind = (uint8_t)xf->kind;
  e.xf_on_source = 0;
  for (int c = 0; c < 3; ++c) { e.scale[c] = xf->scale[c]; e.offset[c] = xf->offset[c]; }
  e.shift = xf->shift;
  e.mask = xf->mask;
  if (b->columnar) {
    e.src_col = e.dst_col = col_addr(*b, (size_t)slot, 0);
    if (k == PST_VEC3F64 && xf->kind == PST_XF_AFFINE && e.src_col % 8 == 0) {
      Workspace& ws = workspace();
      pstk::launch_vec3f64_stream((const double*)(uintptr_t)e.src_col, (double*)(uintptr_t)e.dst_col, b->len, e.scale, e.offset, 3u,
                                  (double*)ws.partials(pstk::stream_partials_bytes(b->len, 3u)), nullptr, s);
    } else {
      execute_entries(false, 0, 0, false, 0, 0, b->len, {e}, false, s);
    }
  } else {
    // in place on interleaved records: the record tile is staged once in LDS, transformed there (each component is read and
    // written by the same lane) and written back with 16-byte stores — whole cache lines move either way, so this beats
    // strided 8-byte accesses (1.72 -> ~1.3 ms at 10^8 LAS-0 points)
    // Round 4: on a layout without padding the same transformation is a records -> records conversion whose other attributes are identity
    // copies -- every byte of a record is written, so nothing is read-modify-written -- and the plan-specialised kernels (one wave per tile,
    // four records per lane in registers) run it faster than the interpreter's in-LDS form (same box, 10^8 LAS-0 records: see DESIGN.md);
    // taken when such a kernel is at hand (compiled, or PST_JIT=sync), the in-LDS form otherwise.
    bool done = false;
    uint64_t attr_bytes = 0;
    for (const Member& mm : b->layout.members) attr_bytes += mm.size;
    static const bool whole_env = [] { const char* v = std::getenv("PST_TRANSFORM_WHOLE_RECORDS"); return !(v && *v == '0'); }();  // the A/B switch
    if (whole_env && attr_bytes == b->layout.size && b->layout.members.size() <= PST_PLAN_MAX_ENTRIES) {
      std::vector<PlanEntry> all;
      for (size_t a = 0; a < b->layout.members.size(); ++a) all.push_back((int)a == slot ? e : identity_entry(b->layout.members[a], b->layout.members[a]));
      const uint64_t base = aos_addr(*b, 0);
      if (specialised_kernel_ready(true, base, (uint32_t)b->layout.size, true, base, (uint32_t)b->layout.size, b->len, all, false)) {
        execute_entries(true, base, (uint32_t)b->layout.size, true, base, (uint32_t)b->layout.size, b->len, all, true, s, nullptr, /*whole_records_in_place=*/true);
        done = true;
      }
    }
    if (!done) execute_entries(true, aos_addr(*b, 0), (uint32_t)b->layout.size, true, aos_addr(*b, 0), (uint32_t)b->layout.size, b->len, {e}, true, s);
  }
  PST_HIP_CHECK(hipGetLastError());
  stream_sync(s);
  PST_API_END
}

}  // extern "C"
