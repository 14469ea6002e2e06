// compute_normals (pasture-algorithms/src/normal_estimation.rs:79-130) — argument checks + host/device plumbing; the
// neighbour search and the plane fit run in normals.hip.
#include "normals_host.hpp"
#include "normals_plan.hpp"
#include "runtime.hpp"

using namespace pst;

namespace {

struct TempDev {
  uint8_t* p = nullptr;
  explicit TempDev(size_t bytes) { p = dev_alloc(bytes, PST_MEM_DEVICE); }
  ~TempDev() { dev_free(p, PST_MEM_DEVICE); }
};

const Member& checked_position(const pst_buffer& b, size_t k) {
  if (b.len < 3)  // :86-88
    throw Error(PST_ERR_TOO_FEW_POINTS, "The point cloud is too small. Please use a point cloud that has 3 or more points!");
  if (k < 3) throw Error(PST_ERR_K_TOO_SMALL, "The k nearest neigbors attribute is too small!");  // :89-91
  AttributeDef pos{"Position3D", DataType{}};
  pos.datatype.kind = PST_VEC3F64;
  const Member* m = b.layout.find(pos);  // view_attribute::<Vector3<f64>>(&POSITION_3D): exact (name, datatype) :97
  if (!m) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  if (k > 64) throw Error(PST_ERR_UNSUPPORTED, "pst_compute_normals: k > 64 is not supported by the register-resident k-best list");
  if (b.len >= 0xFFFFFFF0ull)  // sorted indices and directory entries of the spatial index are uint32_t
    throw Error(PST_ERR_UNSUPPORTED, "pst_compute_normals: more than 2^32 - 17 points per call");
  return *m;
}

void raise_degenerate(long long rc) {
  if (rc == -2) throw Error(PST_ERR_UNSUPPORTED, "pst_compute_normals: more than 2^32 - 17 points per call");
  if (rc < 0) throw hip_failure("normal estimation failed: ");
  if (rc > 0)  // compute_covariance_matrix Err(..) :293-295, unwrapped at :471
    throw Error(PST_ERR_NOT_ENOUGH_NEIGHBOURS,
                "called `Result::unwrap()` on an `Err` value: \"The number of valid (finite and non-NaN values) points in a k nearest "
                "neighborhood is not enough to span a plane!\" (" + std::to_string(rc) + " neighbourhoods)");
}

}  // namespace

extern "C" {

int pst_compute_normals(const pst_buffer* b, size_t k, double* out_normals, double* out_curvature, int64_t* out_knn) {
  PST_API_BEGIN
  not_null(b, "buffer");
  const Member& pm = checked_position(*b, k);
  ensure_device();
  hipStream_t s = current_stream();
  const size_t n = b->len;
  const size_t slot = (size_t)(&pm - b->layout.members.data());
  const uint64_t base = b->columnar ? col_addr(*b, slot, 0) : aos_addr(*b, 0) + pm.offset;
  const uint64_t stride = b->columnar ? pm.size : b->layout.size;
  TempDev d_normals(n * 24), d_curv(n * 8), d_knn(out_knn ? n * k * 8 : 0);
  raise_degenerate(pstk::run_normals((const uint8_t*)(uintptr_t)base, stride, n, (uint32_t)k, (double*)d_normals.p, (double*)d_curv.p,
                                     (long long*)d_knn.p, nullptr, 0, 0, 0, 0, s));
  PST_HIP_CHECK(hipMemcpyAsync(not_null(out_normals, "out_normals"), d_normals.p, n * 24, hipMemcpyDeviceToHost, s));
  PST_HIP_CHECK(hipMemcpyAsync(not_null(out_curvature, "out_curvature"), d_curv.p, n * 8, hipMemcpyDeviceToHost, s));
  if (out_knn) PST_HIP_CHECK(hipMemcpyAsync(out_knn, d_knn.p, n * k * 8, hipMemcpyDeviceToHost, s));
  stream_sync(s);
  PST_API_END
}

// Device-resident: NORMAL (Vec3f32, point_layout.rs:594-597) receives the f64 normal narrowed with `as`; an F64 attribute
// named "Curvature" receives the curvature.  Either attribute may be absent from `dst`; at least one must be present.
int pst_compute_normals_into(const pst_buffer* b, size_t k, pst_buffer* dst) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(dst, "dst");
  const Member& pm = checked_position(*b, k);
  if (dst->len != b->len) throw Error(PST_ERR_RANGE, "target buffer length must equal the point cloud length");
  AttributeDef nd{"Normal", DataType{}}, cd{"Curvature", DataType{}};
  nd.datatype.kind = PST_VEC3F32;
  cd.datatype.kind = PST_F64;
  const int ns = dst->layout.index_of(nd), cs = dst->layout.index_of(cd);
  if (ns < 0 && cs < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "target PointLayout has neither Normal (Vec3f32) nor Curvature (F64)");
  ensure_device();
  hipStream_t s = current_stream();
  const size_t slot = (size_t)(&pm - b->layout.members.data());
  const uint64_t base = b->columnar ? col_addr(*b, slot, 0) : aos_addr(*b, 0) + pm.offset;
  const uint64_t stride = b->columnar ? pm.size : b->layout.size;
  auto attr_addr = [&](int sl, uint64_t& addr, uint64_t& st) {
    if (sl < 0) { addr = 0; st = 0; return; }
    const Member& m = dst->layout.members[(size_t)sl];
    addr = dst->columnar ? col_addr(*dst, (size_t)sl, 0) : aos_addr(*dst, 0) + m.offset;
    st = dst->columnar ? m.size : dst->layout.size;
  };
  uint64_t na, nst, ca, cst;
  attr_addr(ns, na, nst);
  attr_addr(cs, ca, cst);
  raise_degenerate(pstk::run_normals((const uint8_t*)(uintptr_t)base, stride, b->len, (uint32_t)k, nullptr, nullptr, nullptr, nullptr, na, nst, ca, cst, s));
  PST_API_END
}

// Device-resident raw outputs (any of them may be null): normals f64 [n][3], curvature f64 [n], neighbour lists uint32 [n][k] in
// ascending distance (0xFFFFFFFF where the cloud has fewer than k points).  Same checks and panics as pst_compute_normals.
int pst_compute_normals_device(const pst_buffer* b, size_t k, double* d_normals, double* d_curvature, uint32_t* d_knn) {
  PST_API_BEGIN
  not_null(b, "buffer");
  const Member& pm = checked_position(*b, k);
  if (!d_normals && !d_curvature && !d_knn) throw Error(PST_ERR_INVALID_ARGUMENT, "pst_compute_normals_device: no output requested");
  ensure_device();
  hipStream_t s = current_stream();
  const size_t slot = (size_t)(&pm - b->layout.members.data());
  const uint64_t base = b->columnar ? col_addr(*b, slot, 0) : aos_addr(*b, 0) + pm.offset;
  const uint64_t stride = b->columnar ? pm.size : b->layout.size;
  raise_degenerate(pstk::run_normals((const uint8_t*)(uintptr_t)base, stride, b->len, (uint32_t)k, d_normals, d_curvature, nullptr, d_knn, 0, 0, 0, 0, s));
  PST_API_END
}

// ---- stream-ordered form (round 4) -------------------------------------------------------------------------------------------------------------
struct pst_normals_plan {
  pstk::KnnPlan* plan = nullptr;
  size_t k = 0;
  int device = 0;  // the plan's scratch lives there
  ~pst_normals_plan() { if (plan) pstk::knn_plan_free(plan); }
};

namespace {
struct NormalTargets { uint64_t base, stride, na, nst, ca, cst; };
NormalTargets resolve_normal_targets(const pst_buffer& b, size_t k, pst_buffer& dst) {
  const Member& pm = checked_position(b, k);
  if (dst.len != b.len) throw Error(PST_ERR_RANGE, "target buffer length must equal the point cloud length");
  AttributeDef nd{"Normal", DataType{}}, cd{"Curvature", DataType{}};
  nd.datatype.kind = PST_VEC3F32;
  cd.datatype.kind = PST_F64;
  const int ns = dst.layout.index_of(nd), cs = dst.layout.index_of(cd);
  if (ns < 0 && cs < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "target PointLayout has neither Normal (Vec3f32) nor Curvature (F64)");
  const size_t slot = (size_t)(&pm - b.layout.members.data());
  NormalTargets t{};
  t.base = b.columnar ? col_addr(b, slot, 0) : aos_addr(b, 0) + pm.offset;
  t.stride = b.columnar ? pm.size : b.layout.size;
  auto attr_addr = [&](int sl, uint64_t& addr, uint64_t& st) {
    if (sl < 0) { addr = 0; st = 0; return; }
    const Member& m = dst.layout.members[(size_t)sl];
    addr = dst.columnar ? col_addr(dst, (size_t)sl, 0) : aos_addr(dst, 0) + m.offset;
    st = dst.columnar ? m.size : dst.layout.size;
  };
  attr_addr(ns, t.na, t.nst);
  attr_addr(cs, t.ca, t.cst);
  return t;
}
}  // namespace

// One synchronous pst_compute_normals_into (dst holds its results) whose decisions -- frame, grid, box shape, capacities -- are kept.
int pst_compute_normals_plan_create(const pst_buffer* b, size_t k, pst_buffer* dst, pst_normals_plan** out) {
  PST_API_BEGIN
  not_null(b, "buffer");
  not_null(dst, "dst");
  not_null(out, "out");
  const NormalTargets t = resolve_normal_targets(*b, k, *dst);
  ensure_device();
  hipStream_t s = current_stream();
  pstk::KnnPlanRecord rec;
  raise_degenerate(pstk::run_normals((const uint8_t*)(uintptr_t)t.base, t.stride, b->len, (uint32_t)k, nullptr, nullptr, nullptr, nullptr, t.na, t.nst, t.ca, t.cst, s, &rec));
  if (!rec.valid)
    throw Error(PST_ERR_UNSUPPORTED, std::string("pst_compute_normals_plan_create: this cloud has no stream-ordered plan (") + rec.why_not +
                                         "); dst holds the result of the synchronous call");
  auto plan = std::make_unique<pst_normals_plan>();
  plan->k = k;
  const bool packed = t.stride == 24 && (t.base & 7u) == 0;
  plan->plan = pstk::knn_plan_create(rec, packed, s);
  if (!plan->plan) throw hip_failure("normals plan: allocation failed: ");
  PST_HIP_CHECK(hipGetDevice(&plan->device));
  stream_sync(s);
  *out = plan.release();
  PST_API_END
}
int pst_normals_plan_destroy(pst_normals_plan* plan) { delete plan; return PST_OK; }

int pst_compute_normals_into_async(pst_normals_plan* plan, const pst_buffer* b, pst_buffer* dst, uint64_t* device_status2) {
  PST_API_BEGIN
  not_null(plan, "plan");
  not_null(b, "buffer");
  not_null(dst, "dst");
  not_null(device_status2, "device_status2");
  const NormalTargets t = resolve_normal_targets(*b, plan->k, *dst);
  const pstk::KnnPlanRecord& r = pstk::knn_plan_record(plan->plan);
  if (b->len != r.n) throw Error(PST_ERR_INVALID_ARGUMENT, "compute_normals_into_async: the plan was made for " + std::to_string(r.n) + " points, the buffer holds " + std::to_string(b->len));
  if (!pstk::knn_plan_accepts(plan->plan, (const uint8_t*)(uintptr_t)t.base, t.stride))
    throw Error(PST_ERR_INVALID_ARGUMENT, "compute_normals_into_async: the plan was made on a packed Vec3f64 position array (searched in place, no staging copy); this buffer stores "
                                          "Position3D with stride " + std::to_string(t.stride) + " -- make the plan on a buffer with this storage");
  ensure_device();
  int dev = 0;
  PST_HIP_CHECK(hipGetDevice(&dev));
  if (dev != plan->device)
    throw Error(PST_ERR_INVALID_ARGUMENT, "compute_normals_into_async: the plan was made on device " + std::to_string(plan->device) + ", the current device is " + std::to_string(dev));
  if (!pstk::run_normals_replay(plan->plan, (const uint8_t*)(uintptr_t)t.base, t.stride, nullptr, nullptr, nullptr, t.na, t.nst, t.ca, t.cst,
                                (unsigned long long*)device_status2, current_stream()))
    throw hip_failure("normal estimation (stream-ordered) failed: ");
  PST_API_END
}

// The kNN search keeps its device scratch (sorted copy of the positions, keys, directory, directory: ~55 bytes per point, at most PST_SCRATCH_MAX_BYTES = 16 GiB by default) in a
// per-thread cache between calls -- allocating it per call stalled every few calls for seconds at 10^8 points.  This hands it back.
int pst_release_scratch(void) {
  PST_API_BEGIN
  pstk::release_normals_scratch();
  pst::trim_device_pool();  // ... and whatever freed buffers left in the stream-ordered pool (buffer.cpp: release threshold = never)
  PST_API_END
}

int pst_reload_tuning(void) {
  PST_API_BEGIN
  pstk::knn_reload_tuning();
  PST_API_END
}

}  // extern "C"
