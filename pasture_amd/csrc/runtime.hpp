// Internal host runtime declarations: device buffers, per-thread workspace, plan execution.
#pragma once
#include <cstdlib>
#include <atomic>
#include <memory>
#include <vector>

#include "core.hpp"
#include "kernels.hpp"
#include "plan.h"

// Device-backed point buffer.  One struct serves the three reference buffer kinds:
//   VectorBuffer          (point_buffer.rs:659-945)   columnar = false, owns = true
//   HashMapBuffer         (point_buffer.rs:1031-1474) columnar = true,  owns = true   (one 256-B aligned column per attribute;
//                          the per-point HashMap lookup of :1222-1235 becomes a pointer table indexed by layout slot)
//   ExternalMemoryBuffer  (point_buffer.rs:1479-1708) owns = false (caller's device memory)
struct pst_buffer {
  pst::Layout layout;
  bool columnar = false;
  bool owns = true;
  uint32_t memkind = PST_MEM_DEVICE;
  size_t len = 0;       // points
  size_t capacity = 0;  // points
  uint8_t* data = nullptr;         // interleaved storage
  std::vector<uint8_t*> columns;   // columnar storage, layout order
  // Slices borrow the parent's storage (slice.rs:16-43; Rust's borrow checker keeps the parent untouched while one lives).  The C ABI has no
  // borrow checker, so an owning buffer carries a storage epoch -- bumped whenever its storage moves, shrinks, grows or dies -- and a slice
  // remembers the epoch it was cut at: any use of a slice whose parent has been resized or destroyed since is PST_ERR_INVALID_ARGUMENT
  // instead of a read of freed device memory.
  // owning buffers: their own, made WITH the buffer (cutting a slice only reads the parent: two threads may slice one buffer at once); slices: the owning
  // ancestor's; external memory: none
  std::shared_ptr<std::atomic<uint64_t>> epoch = std::make_shared<std::atomic<uint64_t>>(0);
  uint64_t epoch_cut = 0;                        // slices: the ancestor's epoch when the slice was cut
  bool is_slice = false;
  ~pst_buffer();
};

namespace pst {

// per-thread scratch: device partials / result records + a pinned host mirror for small read-backs
struct Workspace {
  uint8_t* dev = nullptr;     // kWorkspaceBytes: small result records
  uint8_t* pinned = nullptr;  // kPinnedBytes
  uint8_t* partials_buf = nullptr;  // grow-only scratch for per-block reduction records
  size_t partials_cap = 0;
  static constexpr size_t kWorkspaceBytes = 1u << 20;
  static constexpr size_t kPinnedBytes = 1u << 12;
  // device scratch of at least `bytes` (grows by reallocation; growth synchronises the device once)
  uint8_t* partials(size_t bytes);
};
Workspace& workspace();
// Small result records a SYNCHRONOUS entry point waits for (an AABB, the encoder's header contribution) are written by the call's last kernel straight
// into the pinned mirror -- host memory the device addresses -- instead of into device memory and copied out by one more launch (a blit kernel per
// hipMemcpyAsync).  PST_RESULTS_TO_HOST=0: the copy (the A/B switch).
inline bool results_to_host() {
  static const bool on = [] { const char* v = std::getenv("PST_RESULTS_TO_HOST"); return !(v && *v == '0'); }();
  return on;
}

uint8_t* dev_alloc(size_t bytes, uint32_t memkind);
void dev_free(uint8_t* p, uint32_t memkind);
// no-throw forms on an explicit stream (stream-ordered pool, or hipMalloc / hipFree without pool support or with PST_NO_POOL)
hipError_t dev_alloc_stream(void** p, size_t bytes, hipStream_t s);  // (out of memory: the pool's unused blocks are given back and the request repeated once)
hipError_t dev_malloc_retry(void** p, size_t bytes);  // hipMalloc with the same second chance
void trim_device_pool();  // pst_release_scratch: unused pool blocks of the current device back to the driver
void dev_free_stream(void* p, hipStream_t s);

// every API entry takes its buffers through not_null(b, "..."): this overload is where a stale slice is caught
void check_live(const pst_buffer& b);
inline const pst_buffer* not_null(const pst_buffer* p, const char* what) {
  if (!p) throw Error(PST_ERR_INVALID_ARGUMENT, std::string(what) + " must not be NULL");
  check_live(*p);
  return p;
}
inline pst_buffer* not_null(pst_buffer* p, const char* what) {
  if (!p) throw Error(PST_ERR_INVALID_ARGUMENT, std::string(what) + " must not be NULL");
  check_live(*p);
  return p;
}
void bump_epoch(pst_buffer& b);  // the storage of an owning buffer is about to move or die

// address helpers
inline uint64_t aos_addr(const pst_buffer& b, size_t point) { return (uint64_t)(uintptr_t)b.data + (uint64_t)point * b.layout.size; }
inline uint64_t col_addr(const pst_buffer& b, size_t slot, size_t point) {
  return (uint64_t)(uintptr_t)b.columns[slot] + (uint64_t)point * b.layout.members[slot].size;
}

// Plan execution: `entries` hold per-mapping descriptors with src_col/dst_col already resolved; the function splits them
// into launches of <= PST_PLAN_MAX_ENTRIES, picks the LDS tile and the kernel body, and enqueues on `stream`.
// If an entry has .bounds set, its AABB record {min xyz, max xyz} is written to bounds_out6 (device-accessible).
void execute_entries(bool src_aos, uint64_t src_base, uint32_t src_stride, bool dst_aos, uint64_t dst_base, uint32_t dst_stride,
                     uint64_t n, const std::vector<PlanEntry>& entries, bool allow_lds, hipStream_t stream, double* bounds_out6 = nullptr,
                     bool whole_records_in_place = false);
bool specialised_kernel_ready(bool src_aos, uint64_t src_base, uint32_t src_stride, bool dst_aos, uint64_t dst_base, uint32_t dst_stride, uint64_t n,
                              const std::vector<PlanEntry>& entries, bool with_bounds);

// identity (same datatype, no transformation) entry between two members
PlanEntry identity_entry(const Member& src, const Member& dst);

void stream_sync(hipStream_t s);
// OwningBuffer::resize; zero_fill = false leaves new points uninitialised (callers that overwrite every byte)
void resize_buffer(pst_buffer& b, size_t count, bool zero_fill);

// {min xyz, max xyz} of POSITION_3D over points [first, first+count) written to out6 (device-accessible); seeds
// +/-f64::MAX (bounds.rs:31-32).  The buffer must have a Position3D attribute.
size_t bounds_partials_scratch_bytes(size_t count);
void bounds_of_range(const pst_buffer& b, size_t first, size_t count, double* out6, hipStream_t stream, void* partials = nullptr);
// AABB::from_min_max (math/bounds.rs:21-26): throws PST_ERR_BOUNDS_INVALID if min > max on any axis
void check_bounds_record(const double r[6], double out_min[3], double out_max[3]);

}  // namespace pst
