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

  // plan of set_all_attributes over the TARGET layout (:459-689); checked before any point is produced... the reference
  // would panic at the first voxel, i.e. also before filtered_buffer changes.
  const Layout& tl = filtered->layout;
  for (const char* w : kWaveform)
    if (find_by_name(tl, w)) throw Error(PST_ERR_UNSUPPORTED_ATTRIBUTE, "Waveform data currently not supported!");
  const size_t na = tl.members.size();
  std::vector<const Member*> src_m(na);
  std::vector<uint32_t> reduce(na), kind(na);
  for (size_t a = 0; a < na; ++a) {
    const Member& t = tl.members[a];
    const Rule* rule = nullptr;
    for (auto& r : kRules)
      if (t.def.name == r.name && t.def.datatype.kind == r.kind) rule = &r;
    if (!rule) throw Error(PST_ERR_UNSUPPORTED_ATTRIBUTE, "attribute is non-standard which is not supported currently: " + t.def.name);
    src_m[a] = find_member(buffer->layout, rule->name, rule->kind);  // view_attribute::<T>(&attributes::X) on the source
    if (!src_m[a]) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute " + t.def.name + " not found in PointLayout of buffer");
    reduce[a] = rule->reduce;
    kind[a] = rule->kind;
  }

  StateGuard g;
  const size_t pslot = (size_t)(pos - buffer->layout.members.data());
  const uint8_t* pos_base = buffer->columnar ? buffer->columns[pslot] : buffer->data + pos->offset;
  const uint64_t pos_stride = buffer->columnar ? pos->size : buffer->layout.size;
  const double leafs[3] = {leafsize_x, leafsize_y, leafsize_z};
  const long long nv = pstk::voxel_grid_build(g.st, pos_base, pos_stride, n, mkx.data(), (uint32_t)mkx.size(), mky.data(), (uint32_t)mky.size(), mkz.data(),
                                              (uint32_t)mkz.size(), mn, leafs, s);
  if (nv < 0) throw Error(PST_ERR_HIP, std::string("voxel grid build failed: ") + hipGetErrorString(hipGetLastError()));

  // filtered_buffer.push_points(centroid) per voxel :161-164 == append nv zero-initialised points and fill the attributes
  const size_t old_len = filtered->len;
  if (filtered->owns && old_len + (size_t)nv > filtered->capacity && old_len) {
    resize_buffer(*filtered, std::max(old_len + (size_t)nv, filtered->capacity * 2), false);
    filtered->len = old_len;
  }
  resize_buffer(*filtered, old_len + (size_t)nv, true);  // UntypedPointBuffer::new zero-fills; padding stays zero
  std::vector<uint64_t> src_addr(na), dst_addr(na);
  std::vector<uint32_t> src_stride(na), dst_stride(na);
  for (size_t a = 0; a < na; ++a) {
    const size_t sslot = (size_t)(src_m[a] - buffer->layout.members.data());
    src_addr[a] = buffer->columnar ? col_addr(*buffer, sslot, 0) : aos_addr(*buffer, 0) + src_m[a]->offset;
    src_stride[a] = (uint32_t)(buffer->columnar ? src_m[a]->size : buffer->layout.size);
    dst_addr[a] = filtered->columnar ? col_addr(*filtered, a, 0) : aos_addr(*filtered, 0) + tl.members[a].offset;
    dst_stride[a] = (uint32_t)(filtered->columnar ? tl.members[a].size : tl.size);
  }
  if (na && !pstk::voxel_grid_reduce(g.st, src_addr.data(), src_stride.data(), dst_addr.data(), dst_stride.data(), reduce.data(), kind.data(), (int)na,
                                     old_len, s))
    throw Error(PST_ERR_HIP, std::string("voxel grid reduction failed: ") + hipGetErrorString(hipGetLastError()));
  // no final synchronisation: nothing is returned to host memory (include/pasture_amd.h conventions); the reductions and the
  // stream-ordered release of the grid state stay in flight on the current stream -- a third host round trip per call would triple
  // the cost of a loaded host (8 ms -> 40+ ms measured with three)
  PST_API_END
}
