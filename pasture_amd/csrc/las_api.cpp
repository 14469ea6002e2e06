// pst_las_encode_points: argument checks + plumbing for the LAS record encoder (las_encode.hip).
// Reference: pasture-io/src/las/raw_writers.rs:203-363, 606-613; las_layout.rs:64-125; las_types.rs.
#include "las_layouts.hpp"
#include "runtime.hpp"

using namespace pst;

using namespace pst::laslayout;

namespace {

// checks + launch of the encoder for source points [src_first, src_first + count); results land in device memory
void encode_range(const pst_buffer* src, size_t src_first, size_t count, uint32_t point_format, const double scale[3], const double offset[3],
                  pst_buffer* dst, size_t dst_first, const double seeds[6], uint32_t max_return, double* dev_bounds, unsigned long long* dev_counts,
                  hipStream_t s) {
  not_null(src, "src"); not_null(dst, "dst"); not_null(scale, "scale"); not_null(offset, "offset");
  if (point_format > 10) throw Error(PST_ERR_INVALID_ARGUMENT, "Unsupported LAS point format " + std::to_string(point_format));
  if (max_return == 0 || max_return > 15) throw Error(PST_ERR_INVALID_ARGUMENT, "max_return must be 5 (legacy header) or 15 (large_file)");
  const Layout typed = typed_layout(point_format), raw = raw_layout(point_format);
  if (src->layout != typed)  // raw_writers.rs:607-613: only the default layout takes this path
    throw Error(PST_ERR_LAYOUT_MISMATCH, "source PointLayout is not the default layout of LAS point format " + std::to_string(point_format));
  if (dst->layout != raw || dst->columnar)
    throw Error(PST_ERR_LAYOUT_MISMATCH, "target must be an interleaved buffer in the exact-binary LAS record layout of the format");
  if (src_first + count < src_first || src_first + count > src->len) throw Error(PST_ERR_RANGE, "source range out of bounds");
  if (dst_first + count < dst_first || dst_first + count > dst->len) throw Error(PST_ERR_RANGE, "target range out of bounds");
  if (count == 0) return;  // :207-209
  uint64_t base[24];
  uint32_t stride[24], esize[24];
  const size_t na = typed.members.size();
  for (size_t a = 0; a < na; ++a) {
    base[a] = src->columnar ? col_addr(*src, a, src_first) : aos_addr(*src, src_first) + typed.members[a].offset;
    stride[a] = (uint32_t)(src->columnar ? typed.members[a].size : typed.size);
    esize[a] = (uint32_t)typed.members[a].size;
  }
  uint8_t* scratch = workspace().partials(pstk::las_encode_workspace_bytes());
  if (!pstk::launch_las_encode((int)point_format, base, stride, esize, (int)na, !src->columnar, aos_addr(*dst, dst_first), count, scale, offset, seeds,
                               max_return, scratch, dev_bounds, dev_counts, s))
    throw hip_failure("LAS encode launch failed: ");
}

}  // namespace

// Asynchronous, ranged variant for pipelined writers: encodes source points [src_first, src_first + count) into
// dst[dst_first ..) on the current stream and leaves THIS call's header contribution in device memory: device_bounds6 =
// {min xyz, max xyz} of the positions (seeded with the identities), device_counts16[0] = positions outside the i32 range (the
// caller must treat > 0 as the panic of write_helpers.rs:15-17), device_counts16[r] = points with return number r.
extern "C" int pst_las_encode_range_async(const pst_buffer* src, size_t src_first, size_t count, uint32_t point_format, const double scale[3],
                                          const double offset[3], pst_buffer* dst, size_t dst_first, double* device_bounds6,
                                          uint64_t* device_counts16, uint32_t max_return) {
  PST_API_BEGIN
  not_null(device_bounds6, "device_bounds6"); not_null(device_counts16, "device_counts16");
  if (count == 0) return PST_OK;
  ensure_device();
  const double seeds[6] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
                           -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  encode_range(src, src_first, count, point_format, scale, offset, dst, dst_first, seeds, max_return, device_bounds6,
               (unsigned long long*)device_counts16, current_stream());
  PST_API_END
}

extern "C" int pst_las_encode_points(const pst_buffer* src, uint32_t point_format, const double scale[3], const double offset[3], pst_buffer* dst,
                                     size_t dst_first, double bounds_inout[6], uint64_t points_by_return[15], uint32_t max_return) {
  PST_API_BEGIN
  not_null(src, "src"); not_null(bounds_inout, "bounds_inout"); not_null(points_by_return, "points_by_return");
  const size_t n = src->len;
  if (n) ensure_device();
  if (n == 0) {  // argument checks only
    encode_range(src, 0, 0, point_format, scale, offset, dst, dst_first, bounds_inout, max_return, nullptr, nullptr, nullptr);
    return PST_OK;
  }
  hipStream_t s = current_stream();
  Workspace& ws = workspace();
  const bool direct = results_to_host();  // the fold kernel's last block writes the 48 + 128 bytes into the pinned mirror itself
  double* dev_bounds = direct ? (double*)(ws.pinned + 256) : (double*)(ws.dev + 1024);
  unsigned long long* dev_counts = direct ? (unsigned long long*)(ws.pinned + 512) : (unsigned long long*)(ws.dev + 1024 + 64);
  encode_range(src, 0, n, point_format, scale, offset, dst, dst_first, bounds_inout, max_return, dev_bounds, dev_counts, s);
  if (!direct) {
    PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 256, dev_bounds, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 512, dev_counts, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  }
  stream_sync(s);
  const unsigned long long* counts = (const unsigned long long*)(ws.pinned + 512);
  if (counts[0] != 0)  // write_helpers.rs:15-17 .expect(...)
    throw Error(PST_ERR_RANGE, "write_position_as_las_position: Position is out of bounds given the current LAS offset and scale! (" +
                                   std::to_string(counts[0]) + " positions)");
  std::memcpy(bounds_inout, ws.pinned + 256, 6 * sizeof(double));
  for (uint32_t r = 1; r <= max_return; ++r) points_by_return[r - 1] += counts[r];
  PST_API_END
}
