// pst_buffer_append / pst_buffer_filter_into / pst_buffer_filter: host plumbing for append and predicate compaction.
// Reference: pasture-core/src/containers/point_buffer.rs:419-489 (OwningBufferExt::append), :1064-1136 (HashMapBuffer::filter,
// filter_into); the predicate arrives as a byte mask (benches/buffer_filter_bench.rs:62-64).
#include <cstdint>
#include <memory>

#include "runtime.hpp"

using namespace pst;

namespace {

struct TempDev {
  uint8_t* p = nullptr;
  explicit TempDev(size_t bytes) { p = bytes ? dev_alloc(bytes, PST_MEM_DEVICE) : nullptr; }
  ~TempDev() { dev_free(p, PST_MEM_DEVICE); }
};

// Compaction of src's points with mask != 0 into dst[0, matches); returns the number of matches.
// counted: the caller has just run launch_filter_count for the same mask / length (the tile counts and offsets are still in the
// workspace and `num_matches_hint` is the count it read): the count and scan launches are not repeated.
// count_out (stream-ordered form): no host synchronisation at all -- the count is copied to `count_out` (device-accessible memory, optional)
// in stream order and the "more matches than the hint" panic is the caller's to check; needs a device mask and a hint.
size_t filter_into(const pst_buffer& src, pst_buffer& dst, const uint8_t* mask, bool mask_on_device, int64_t num_matches_hint, bool counted = false,
                   bool stream_ordered = false, unsigned long long* count_out = nullptr) {
  if (!src.columnar) throw Error(PST_ERR_INVALID_ARGUMENT, "filter is defined on HashMapBuffer (point_buffer.rs:1064)");
  if (dst.layout != src.layout) throw Error(PST_ERR_LAYOUT_MISMATCH, "PointLayouts must match");  // :1088-1090
  const size_t n = src.len;
  if (n == 0) return 0;
  not_null(mask, "mask");
  ensure_device();
  hipStream_t s = current_stream();
  TempDev staged(mask_on_device ? 0 : n);
  const uint8_t* mask_dev = mask;
  if (!mask_on_device) {
    PST_HIP_CHECK(hipMemcpyAsync(staged.p, mask, n, hipMemcpyHostToDevice, s));
    mask_dev = staged.p;
  }
  const bool dst_aos = !dst.columnar;
  const uint32_t dst_stride = (uint32_t)dst.layout.size;
  const uint32_t tile = pstk::filter_tile(dst_aos, dst_stride);
  uint8_t* scratch = workspace().partials(pstk::filter_workspace_bytes(n));
  const unsigned long long* total_dev = nullptr;
  Workspace& ws = workspace();
  if (!counted) {
    // (stream-ordered form: the scan kernel writes the total to the caller's word itself -- no copy operation between the kernels)
    // (synchronous form: the same, into the pinned mirror the host reads after the wait -- results_to_host, runtime.hpp)
    unsigned long long* const host_word = (unsigned long long*)(ws.pinned + 768);
    pstk::launch_filter_count(mask_dev, n, tile, scratch, &total_dev, s, stream_ordered ? count_out : results_to_host() ? host_word : nullptr);
    if (!stream_ordered && !results_to_host()) PST_HIP_CHECK(hipMemcpyAsync(host_word, total_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  }
  // With Some(num_matches) (the reference's bench passes it) nothing on the host depends on the count before the copies are
  // launched: count, scan and scatter run back to back and the count is read once, at the end.  Without it the target check
  // (:1093-1095) needs the count first.
  const bool hinted = num_matches_hint >= 0;
  size_t matches = 0;
  if (!hinted) {
    stream_sync(s);
    matches = (size_t)*(const unsigned long long*)(ws.pinned + 768);
  }
  const size_t num_matches = hinted ? (size_t)num_matches_hint : matches;  // :1091-1092
  if (dst.len < num_matches)  // :1093-1095
    throw Error(PST_ERR_RANGE, "buffer.len() must be at least as large as the number of predicate matches");
  const size_t na = src.layout.members.size();
  // An interleaved target whose records do not fit the kernels' LDS record tile (points beyond ~10 KB: ByteArray attributes; the reference copies any
  // size, point_buffer.rs:1096-1133): the target's first num_matches points are transposed into temporary columns, compacted INTO those (points the
  // mask does not reach keep what the target held), and transposed back -- three passes instead of one, on a path no bulk workload takes.
  std::unique_ptr<pst_buffer> staged_cols;
  pst_buffer* out = &dst;
  std::vector<PlanEntry> to_cols, to_recs;
  if (dst_aos && na && num_matches > 0 && !pstk::filter_record_tile_fits(tile, dst_stride)) {
    staged_cols = std::make_unique<pst_buffer>();
    staged_cols->layout = dst.layout;
    staged_cols->columnar = true;
    staged_cols->columns.assign(na, nullptr);
    resize_buffer(*staged_cols, num_matches, false);
    for (size_t a = 0; a < na; ++a) {
      const Member& m = dst.layout.members[a];
      PlanEntry e = identity_entry(m, m);
      e.dst_col = col_addr(*staged_cols, a, 0);
      to_cols.push_back(e);
      e.dst_col = 0;
      e.src_col = col_addr(*staged_cols, a, 0);
      to_recs.push_back(e);
    }
    execute_entries(true, aos_addr(dst, 0), dst_stride, false, 0, 0, num_matches, to_cols, true, s);
    out = staged_cols.get();
  }
  const bool out_aos = !out->columnar;
  std::vector<uint64_t> src_addr(na), dst_addr(na);
  std::vector<uint32_t> src_stride(na), dst_off(na), size(na);
  size_t covered = 0;
  for (size_t a = 0; a < na; ++a) {
    const Member& m = src.layout.members[a];
    src_addr[a] = src.columnar ? col_addr(src, a, 0) : aos_addr(src, 0) + m.offset;
    src_stride[a] = (uint32_t)(src.columnar ? m.size : src.layout.size);
    dst_addr[a] = out->columnar ? col_addr(*out, a, 0) : 0;
    dst_off[a] = (uint32_t)m.offset;
    size[a] = (uint32_t)m.size;
    covered += m.size;
  }
  if (na && (hinted ? num_matches : std::min(matches, num_matches)) > 0 &&
      !pstk::launch_filter_scatter(mask_dev, n, tile, scratch, num_matches, src_addr.data(), src_stride.data(), dst_addr.data(), dst_off.data(),
                                   size.data(), (int)na, out_aos, out_aos ? aos_addr(*out, 0) : 0, dst_stride, covered == dst.layout.size, s))
    throw hip_failure("filter launch failed: ");
  if (staged_cols) execute_entries(false, 0, 0, true, aos_addr(dst, 0), dst_stride, num_matches, to_recs, true, s);
  if (stream_ordered) return num_matches;
  stream_sync(s);  // the staged mask is released on return
  if (hinted) matches = (size_t)*(const unsigned long long*)(ws.pinned + 768);
  if (matches > num_matches)  // the reference indexes dst_attribute_data[..num_matches] out of range (:1103-1108)
    throw Error(PST_ERR_RANGE, "range end index out of range for slice (more predicate matches than num_matches_hint)");
  return matches;
}

}  // namespace

extern "C" {

int pst_buffer_filter_into(const pst_buffer* src, pst_buffer* dst, const uint8_t* mask, uint32_t mask_memkind, int64_t num_matches_hint,
                           size_t* out_matches) {
  PST_API_BEGIN
  if (mask_memkind > PST_MEM_PINNED_HOST) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid mask memory kind");
  const size_t m = filter_into(*not_null(src, "src"), *not_null(dst, "dst"), mask, mask_memkind == PST_MEM_DEVICE, num_matches_hint);
  if (out_matches) *out_matches = m;
  PST_API_END
}

int pst_buffer_filter_into_async(const pst_buffer* src, pst_buffer* dst, const uint8_t* device_mask, size_t num_matches, uint64_t* device_count_out) {
  PST_API_BEGIN
  if (num_matches > (size_t)INT64_MAX) throw Error(PST_ERR_INVALID_ARGUMENT, "num_matches out of range");
  if (not_null(src, "src")->len == 0 && device_count_out) {
    ensure_device();
    PST_HIP_CHECK(hipMemsetAsync(device_count_out, 0, sizeof(unsigned long long), current_stream()));
  }
  filter_into(*src, *not_null(dst, "dst"), device_mask, true, (int64_t)num_matches, false, true, reinterpret_cast<unsigned long long*>(device_count_out));
  PST_API_END
}

int pst_buffer_filter(const pst_buffer* src, const uint8_t* mask, uint32_t mask_memkind, uint32_t out_storage, pst_buffer** out) {
  PST_API_BEGIN
  not_null(src, "src");
  not_null(out, "out");
  if (!src->columnar) throw Error(PST_ERR_INVALID_ARGUMENT, "filter is defined on HashMapBuffer (point_buffer.rs:1064)");
  if (out_storage > PST_STORAGE_COLUMNAR) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid storage kind");
  if (mask_memkind > PST_MEM_PINNED_HOST) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid mask memory kind");
  auto b = std::make_unique<pst_buffer>();
  b->layout = src->layout;
  b->columnar = out_storage == PST_STORAGE_COLUMNAR;
  if (b->columnar) b->columns.assign(b->layout.members.size(), nullptr);
  // filter(): count, allocate exactly, filter_into (:1071-1075).  The count comes from the same device pass: allocate for the
  // worst case is wasteful, so count first through a zero-length probe.
  size_t matches = 0;
  if (src->len) {
    ensure_device();
    hipStream_t s = current_stream();
    not_null(mask, "mask");
    TempDev staged(mask_memkind == PST_MEM_DEVICE ? 0 : src->len);
    const uint8_t* mask_dev = mask;
    if (mask_memkind != PST_MEM_DEVICE) {
      PST_HIP_CHECK(hipMemcpyAsync(staged.p, mask, src->len, hipMemcpyHostToDevice, s));
      mask_dev = staged.p;
    }
    const uint32_t tile = pstk::filter_tile(!b->columnar, (uint32_t)b->layout.size);
    uint8_t* scratch = workspace().partials(pstk::filter_workspace_bytes(src->len));
    const unsigned long long* total_dev = nullptr;
    pstk::launch_filter_count(mask_dev, src->len, tile, scratch, &total_dev, s);
    Workspace& ws = workspace();
    PST_HIP_CHECK(hipMemcpyAsync(ws.pinned + 768, total_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    stream_sync(s);
    matches = (size_t)*(const unsigned long long*)(ws.pinned + 768);
    resize_buffer(*b, matches, false);  // every byte of the attributes is overwritten; padding stays as allocated
    if (!b->columnar && matches) {      // ... except record padding, which VectorBuffer::resize zero-fills (:831-835)
      size_t covered = 0;
      for (auto& m : b->layout.members) covered += m.size;
      if (covered != b->layout.size) PST_HIP_CHECK(hipMemsetAsync(b->data, 0, matches * b->layout.size, s));
    }
    filter_into(*src, *b, mask_dev, true, (int64_t)matches, true);
  }
  *out = b.release();
  PST_API_END
}

int pst_buffer_append(pst_buffer* self, const pst_buffer* other) {
  PST_API_BEGIN
  not_null(self, "self");
  not_null(other, "other");
  if (self->layout != other->layout)  // :420 assert_eq!
    throw Error(PST_ERR_LAYOUT_MISMATCH, "assertion failed: self.point_layout() == other.point_layout()");
  const size_t old_len = self->len, add = other->len, new_len = old_len + add;
  if (add == 0) return PST_OK;
  ensure_device();
  // amortised growth like Vec (push_points / resize): at least double the capacity
  if (self->owns && new_len > self->capacity) {
    const size_t want = std::max(new_len, self->capacity * 2);
    resize_buffer(*self, want, false);
    self->len = old_len;
  }
  size_t covered = 0;
  for (auto& m : self->layout.members) covered += m.size;
  if (!self->columnar && !other->columnar) {  // :430-439: Vec::extend_from_slice of whole records (padding bytes included)
    resize_buffer(*self, new_len, false);
    hipStream_t s = current_stream();
    PST_HIP_CHECK(hipMemcpyAsync(self->data + old_len * self->layout.size, other->data, add * self->layout.size, hipMemcpyDefault, s));
    stream_sync(s);
    return PST_OK;
  }
  // :441-443 resize() zero-fills; only record padding can stay visible, every attribute byte is overwritten below
  resize_buffer(*self, new_len, !self->columnar && covered != self->layout.size);
  pst_layout l{self->layout};
  pst_converter* c = nullptr;
  int rc = pst_converter_create(&l, &l, 0, &c);
  if (rc != PST_OK) return rc;
  rc = pst_converter_convert_into_range(c, const_cast<pst_buffer*>(other), 0, add, self, old_len, new_len);
  pst_converter_destroy(c);
  if (rc != PST_OK) return rc;
  PST_API_END
}

}  // extern "C"
