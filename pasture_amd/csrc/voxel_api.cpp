// pst_voxelgrid_filter: argument checks, axis markers and plumbing for voxel.hip.
// Reference: pasture-algorithms/src/voxel_grid.rs:55-83 (create_markers_for_axis), :109-166 (voxelgrid_filter), :459-689.
#include <climits>
#include <cstring>

#include "runtime.hpp"

using namespace pst;

namespace {

struct Rule { const char* name; uint32_t kind; uint32_t reduce; };
// set_all_attributes :478-689: name AND datatype must equal the builtin definition
const Rule kRules[] = {
    {"Position3D", PST_VEC3F64, pstk::VX_AVG_VEC},       {"Intensity", PST_U16, pstk::VX_AVG_NUM},
    {"ReturnNumber", PST_U8, pstk::VX_MOST_COMMON},      {"NumberOfReturns", PST_U8, pstk::VX_MOST_COMMON},
    {"ClassificationFlags", PST_U8, pstk::VX_MAX_POOL},  {"ScannerChannel", PST_U8, pstk::VX_MOST_COMMON},
    {"ScanDirectionFlag", PST_U8, pstk::VX_MOST_COMMON_BOOL}, {"EdgeOfFlightLine", PST_U8, pstk::VX_MOST_COMMON_BOOL},
    {"Classification", PST_U8, pstk::VX_MOST_COMMON},    {"ScanAngleRank", PST_I8, pstk::VX_MOST_COMMON},
    {"ScanAngle", PST_I16, pstk::VX_MOST_COMMON},        {"UserData", PST_U8, pstk::VX_MOST_COMMON},
    {"PointSourceID", PST_U16, pstk::VX_MOST_COMMON},    {"ColorRGB", PST_VEC3U16, pstk::VX_AVG_VEC},
    {"GpsTime", PST_F64, pstk::VX_MAX_POOL},             {"NIR", PST_U16, pstk::VX_AVG_NUM},
    {"PointID", PST_U64, pstk::VX_MAX_POOL},             {"Normal", PST_VEC3F32, pstk::VX_AVG_VEC},
};
const char* kWaveform[] = {"WaveformDataOffset", "WaveformPacketSize", "WaveformParameters", "WavePacketDescriptorIndex", "ReturnPointWaveformLocation"};

// create_markers_for_axis :55-83: curr = min; while curr < max { curr += leafsize; push(curr) } (accumulated, not min + k*leaf)
std::vector<double> create_markers(double mn, double mx, double leafsize) {
  std::vector<double> m;
  double curr = mn;
  while (curr < mx) {
    const double next = curr + leafsize;
    if (!(next > curr))
      throw Error(PST_ERR_INVALID_ARGUMENT, "voxelgrid_filter: leaf size does not advance the marker (the reference loops forever)");
    curr = next;
    m.push_back(curr);
    if (m.size() > 2097151u) throw Error(PST_ERR_UNSUPPORTED, "voxelgrid_filter: more than 2^21 - 1 markers on one axis");
  }
  return m;
}

const Member* find_member(const Layout& l, const char* name, uint32_t kind) {
  for (auto& m : l.members)
    if (m.def.name == name && m.def.datatype.kind == kind) return &m;
  return nullptr;
}
const Member* find_by_name(const Layout& l, const char* name) {
  for (auto& m : l.members)
    if (m.def.name == name) return &m;
  return nullptr;
}

// plan of set_all_attributes over the TARGET layout (:459-689); checked before any point is produced... the reference would panic at the
// first voxel, i.e. also before filtered_buffer changes.
struct AttrPlan {
  std::vector<const Member*> src_m;
  std::vector<uint32_t> reduce, kind;
};
AttrPlan attribute_plan(const pst_buffer& buffer, const Layout& tl) {
  for (const char* w : kWaveform)
    if (find_by_name(tl, w)) throw Error(PST_ERR_UNSUPPORTED_ATTRIBUTE, "Waveform data currently not supported!");
  const size_t na = tl.members.size();
  AttrPlan p;
  p.src_m.resize(na); p.reduce.resize(na); p.kind.resize(na);
  for (size_t a = 0; a < na; ++a) {
    const Member& t = tl.members[a];
    const Rule* rule = nullptr;
    for (auto& r : kRules)
      if (t.def.name == r.name && t.def.datatype.kind == r.kind) rule = &r;
    if (!rule) throw Error(PST_ERR_UNSUPPORTED_ATTRIBUTE, "attribute is non-standard which is not supported currently: " + t.def.name);
    p.src_m[a] = find_member(buffer.layout, rule->name, rule->kind);  // view_attribute::<T>(&attributes::X) on the source
    if (!p.src_m[a]) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute " + t.def.name + " not found in PointLayout of buffer");
    p.reduce[a] = rule->reduce;
    p.kind[a] = rule->kind;
  }
  return p;
}
// the reductions of every attribute of `filtered`'s layout into its points [first, first + voxels)
void run_reductions(pstk::VoxelGridState* st, const pst_buffer& buffer, pst_buffer& filtered, const AttrPlan& p, size_t first, hipStream_t s) {
  const Layout& tl = filtered.layout;
  const size_t na = tl.members.size();
  std::vector<uint64_t> src_addr(na), dst_addr(na);
  std::vector<uint32_t> src_stride(na), dst_stride(na);
  for (size_t a = 0; a < na; ++a) {
    const size_t sslot = (size_t)(p.src_m[a] - buffer.layout.members.data());
    src_addr[a] = buffer.columnar ? col_addr(buffer, sslot, 0) : aos_addr(buffer, 0) + p.src_m[a]->offset;
    src_stride[a] = (uint32_t)(buffer.columnar ? p.src_m[a]->size : buffer.layout.size);
    dst_addr[a] = filtered.columnar ? col_addr(filtered, a, 0) : aos_addr(filtered, 0) + tl.members[a].offset;
    dst_stride[a] = (uint32_t)(filtered.columnar ? tl.members[a].size : tl.size);
  }
  if (na && !pstk::voxel_grid_reduce(st, src_addr.data(), src_stride.data(), dst_addr.data(), dst_stride.data(), p.reduce.data(), p.kind.data(), (int)na, first, s))
    throw hip_failure("voxel grid reduction failed: ");
}

struct StateGuard {
  pstk::VoxelGridState* st = nullptr;
  ~StateGuard() { if (st) pstk::voxel_grid_free(st); }
};

}  // namespace

extern "C" int pst_voxelgrid_filter(const pst_buffer* buffer, double leafsize_x, double leafsize_y, double leafsize_z, pst_buffer* filtered) {
  PST_API_BEGIN
  not_null(buffer, "buffer");
  not_null(filtered, "filtered");
  const Member* pos = find_member(buffer->layout, "Position3D", PST_VEC3F64);
  if (!pos)  // :116-122
    throw Error(PST_ERR_MISSING_ATTRIBUTE,
                "The PointBuffer does not have the attribute attributes::POSITION_3D which is needed for the creation of the voxel grid.");
  const size_t n = buffer->len;
  if (n == 0) throw Error(PST_ERR_BOUNDS_INVALID, "called `Option::unwrap()` on a `None` value");  // :125 calculate_bounds(buffer).unwrap()
  if (n >= 0xFFFFFFF0ull) throw Error(PST_ERR_UNSUPPORTED, "voxelgrid_filter: more than 2^32 - 17 points per call (sorted point indices are uint32_t)");
  ensure_device();
  hipStream_t s = current_stream();
  Workspace& ws = workspace();
  double* dev6 = (double*)(ws.dev + 2048);
  bounds_of_range(*buffer, 0, n, dev6, s);
  PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 1024, dev6, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  stream_sync(s);
  double rec[6], mn[3], mx[3];
  std::memcpy(rec, ws.pinned + 1024, sizeof(rec));
  check_bounds_record(rec, mn, mx);  // AABB::from_min_max inside calculate_bounds
  const std::vector<double> mkx = create_markers(mn[0], mx[0], leafsize_x), mky = create_markers(mn[1], mx[1], leafsize_y),
                            mkz = create_markers(mn[2], mx[2], leafsize_z);

  const AttrPlan ap = attribute_plan(*buffer, filtered->layout);

  StateGuard g;
  const size_t pslot = (size_t)(pos - buffer->layout.members.data());
  const uint8_t* pos_base = buffer->columnar ? buffer->columns[pslot] : buffer->data + pos->offset;
  const uint64_t pos_stride = buffer->columnar ? pos->size : buffer->layout.size;
  const double leafs[3] = {leafsize_x, leafsize_y, leafsize_z};
  const long long nv = pstk::voxel_grid_build(g.st, pos_base, pos_stride, n, mkx.data(), (uint32_t)mkx.size(), mky.data(), (uint32_t)mky.size(), mkz.data(),
                                              (uint32_t)mkz.size(), mn, leafs, s);
  if (nv < 0) throw hip_failure("voxel grid build failed: ");

  // filtered_buffer.push_points(centroid) per voxel :161-164 == append nv zero-initialised points and fill the attributes
  const size_t old_len = filtered->len;
  if (filtered->owns && old_len + (size_t)nv > filtered->capacity && old_len) {
    resize_buffer(*filtered, std::max(old_len + (size_t)nv, filtered->capacity * 2), false);
    filtered->len = old_len;
  }
  resize_buffer(*filtered, old_len + (size_t)nv, true);  // UntypedPointBuffer::new zero-fills; padding stays zero
  run_reductions(g.st, *buffer, *filtered, ap, old_len, s);
  // no final synchronisation: nothing is returned to host memory (include/pasture_amd.h conventions); the reductions and the
  // stream-ordered release of the grid state stay in flight on the current stream -- a third host round trip per call would triple
  // the cost of a loaded host (8 ms -> 40+ ms measured with three)
  PST_API_END
}

// ---- stream-ordered form (round 4): plan once per cloud shape, then calls without a host round trip -----------------------------------------
struct pst_voxel_plan {
  pstk::VoxelGridState* st = nullptr;
  pstk::VoxelPlanShape shape{};
  double* bounds6 = nullptr;  // device record of the call's calculate_bounds
  uint8_t* partials = nullptr;  // its per-block partial records (the thread's workspace is per stream and would be created inside a capture)
  int device = 0;
  ~pst_voxel_plan() {
    if (st) pstk::voxel_grid_free(st);
    if (bounds6) dev_free((uint8_t*)bounds6, PST_MEM_DEVICE);
    if (partials) dev_free(partials, PST_MEM_DEVICE);
  }
};

extern "C" int pst_voxelgrid_plan_create(const pst_buffer* buffer, double leafsize_x, double leafsize_y, double leafsize_z, pst_voxel_plan** out, size_t* max_voxels) {
  PST_API_BEGIN
  not_null(buffer, "buffer");
  not_null(out, "out");
  const Member* pos = find_member(buffer->layout, "Position3D", PST_VEC3F64);
  if (!pos)
    throw Error(PST_ERR_MISSING_ATTRIBUTE,
                "The PointBuffer does not have the attribute attributes::POSITION_3D which is needed for the creation of the voxel grid.");
  const size_t n = buffer->len;
  if (n == 0) throw Error(PST_ERR_BOUNDS_INVALID, "called `Option::unwrap()` on a `None` value");
  if (n >= 0xFFFFFFF0ull) throw Error(PST_ERR_UNSUPPORTED, "voxelgrid_filter: more than 2^32 - 17 points per call (sorted point indices are uint32_t)");
  ensure_device();
  hipStream_t s = current_stream();
  Workspace& ws = workspace();
  // ONE synchronous pass over this cloud: its bounds, the marker counts of its axes and its number of occupied voxels size the plan
  double* dev6 = (double*)(ws.dev + 2048);
  bounds_of_range(*buffer, 0, n, dev6, s);
  PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 1024, dev6, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  stream_sync(s);
  double rec[6], mn[3], mx[3];
  std::memcpy(rec, ws.pinned + 1024, sizeof(rec));
  check_bounds_record(rec, mn, mx);
  const std::vector<double> mkx = create_markers(mn[0], mx[0], leafsize_x), mky = create_markers(mn[1], mx[1], leafsize_y),
                            mkz = create_markers(mn[2], mx[2], leafsize_z);
  StateGuard g;
  const size_t pslot = (size_t)(pos - buffer->layout.members.data());
  const uint8_t* pos_base = buffer->columnar ? buffer->columns[pslot] : buffer->data + pos->offset;
  const uint64_t pos_stride = buffer->columnar ? pos->size : buffer->layout.size;
  const double leafs[3] = {leafsize_x, leafsize_y, leafsize_z};
  const long long nv = pstk::voxel_grid_build(g.st, pos_base, pos_stride, n, mkx.data(), (uint32_t)mkx.size(), mky.data(), (uint32_t)mky.size(), mkz.data(),
                                              (uint32_t)mkz.size(), mn, leafs, s);
  if (nv < 0) throw hip_failure("voxel grid build failed: ");
  stream_sync(s);
  // capacities: an eighth more markers per axis (a later cloud of the same shape may be a little larger), a quarter more voxels
  auto plan = std::make_unique<pst_voxel_plan>();
  pstk::VoxelPlanShape& sh = plan->shape;
  sh.n = n;
  const size_t counts[3] = {mkx.size(), mky.size(), mkz.size()};
  size_t total = 0;
  for (int a = 0; a < 3; ++a) {
    const size_t cap = counts[a] + std::max<size_t>(8, counts[a] / 8);
    uint32_t b = 1;
    while (b < 21 && (1ull << b) < cap) ++b;
    sh.bits[a] = b;
    total += std::min<size_t>(cap, 1ull << b);
    sh.leaf[a] = leafs[a];
  }
  sh.cap_markers = (uint32_t)std::min<size_t>(total, 3u * 2097152u);
  sh.cap_voxels = std::min<uint64_t>((uint64_t)n, (uint64_t)nv + (uint64_t)nv / 4 + 1024);
  {
    const uint64_t groups = ((uint64_t)nv + 63) / 64;
    sh.stage_cap = (uint32_t)std::min<uint64_t>(6144, std::max<uint64_t>(1024, ((uint64_t)n * 8 / 5) / std::max<uint64_t>(1, groups) + 63)) & ~63u;
  }
  plan->st = pstk::voxel_plan_create(sh, s);
  if (!plan->st) throw hip_failure("voxel plan: allocation failed: ");
  plan->bounds6 = (double*)dev_alloc(64, PST_MEM_DEVICE);
  PST_HIP_CHECK(hipGetDevice(&plan->device));
  plan->partials = dev_alloc(bounds_partials_scratch_bytes(n), PST_MEM_DEVICE);
  stream_sync(s);
  if (max_voxels) *max_voxels = (size_t)sh.cap_voxels;
  *out = plan.release();
  PST_API_END
}

extern "C" int pst_voxelgrid_plan_destroy(pst_voxel_plan* plan) { delete plan; return PST_OK; }

extern "C" int pst_voxelgrid_filter_async(pst_voxel_plan* plan, const pst_buffer* buffer, pst_buffer* filtered, size_t dst_first, uint64_t* device_count_and_status) {
  PST_API_BEGIN
  not_null(plan, "plan");
  not_null(buffer, "buffer");
  not_null(filtered, "filtered");
  not_null(device_count_and_status, "device_count_and_status");
  const Member* pos = find_member(buffer->layout, "Position3D", PST_VEC3F64);
  if (!pos)
    throw Error(PST_ERR_MISSING_ATTRIBUTE,
                "The PointBuffer does not have the attribute attributes::POSITION_3D which is needed for the creation of the voxel grid.");
  if (buffer->len != plan->shape.n)
    throw Error(PST_ERR_INVALID_ARGUMENT, "voxelgrid_filter_async: the plan was made for " + std::to_string(plan->shape.n) + " points, the buffer holds " +
                                              std::to_string(buffer->len));
  if (dst_first + plan->shape.cap_voxels < dst_first || dst_first + plan->shape.cap_voxels > filtered->len)
    throw Error(PST_ERR_RANGE, "voxelgrid_filter_async: filtered must hold dst_first + max_voxels = " + std::to_string(dst_first + plan->shape.cap_voxels) +
                                   " points (it holds " + std::to_string(filtered->len) + "): the voxel count is only known on the device");
  const AttrPlan ap = attribute_plan(*buffer, filtered->layout);
  ensure_device();
  int dev = 0;
  PST_HIP_CHECK(hipGetDevice(&dev));
  if (dev != plan->device)  // the plan's scratch lives on the device it was made on
    throw Error(PST_ERR_INVALID_ARGUMENT, "voxelgrid_filter_async: the plan was made on device " + std::to_string(plan->device) + ", the current device is " + std::to_string(dev));
  hipStream_t s = current_stream();
  bounds_of_range(*buffer, 0, buffer->len, plan->bounds6, s, plan->partials);
  const size_t pslot = (size_t)(pos - buffer->layout.members.data());
  const uint8_t* pos_base = buffer->columnar ? buffer->columns[pslot] : buffer->data + pos->offset;
  const uint64_t pos_stride = buffer->columnar ? pos->size : buffer->layout.size;
  if (!pstk::voxel_grid_build_async(plan->st, pos_base, pos_stride, plan->bounds6, (unsigned long long*)device_count_and_status, s))
    throw hip_failure("voxel grid build failed: ");
  run_reductions(plan->st, *buffer, *filtered, ap, dst_first, s);
  PST_API_END
}
