// Device-backed buffers + their C ABI.
// Reference: pasture-core/src/containers/point_buffer.rs (VectorBuffer :659-945, HashMapBuffer :1031-1474,
// ExternalMemoryBuffer :1479-1708) — storage shape and accessor semantics (resize zero-fills, ranges are checked).
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "device_sort.hpp"
#include "runtime.hpp"

namespace pst {

// ---- device / stream state -----------------------------------------------------------------------------
static thread_local hipStream_t t_stream = nullptr;
hipStream_t current_stream() { return t_stream; }

void ensure_device() {
  static std::once_flag once;
  static int n_dev = 0;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] { err = hipGetDeviceCount(&n_dev); });
  if (err != hipSuccess || n_dev <= 0)
    throw Error(PST_ERR_NO_DEVICE,
                "no HIP device available: pasture_amd has no CPU fallback for compute paths (hipGetDeviceCount: " +
                    std::string(err == hipSuccess ? "0 devices" : hipGetErrorString(err)) + ")");
}

void stream_sync(hipStream_t s) { PST_HIP_CHECK(hipStreamSynchronize(s)); }

// One workspace per (thread, device, stream): the asynchronous entry points of one thread may run on several streams at once (a compute
// and a copy stream, the pipelined LAS reader / writer), and their per-block partial records and result records must not alias; and the
// null stream is the same key on every GPU, so after pst_set_device(d) a thread must not be handed scratch that lives on another device.
Workspace& workspace() {
  struct Key {
    int dev;
    hipStream_t stream;
    bool operator==(const Key& o) const { return dev == o.dev && stream == o.stream; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const { return std::hash<const void*>()((const void*)k.stream) ^ ((size_t)(unsigned)k.dev * 0x9E3779B97F4A7C15ull); }
  };
  static thread_local std::unordered_map<Key, Workspace, KeyHash> by_key;
  ensure_device();
  Key cur{0, current_stream()};
  PST_HIP_CHECK(hipGetDevice(&cur.dev));
  auto it = by_key.find(cur);
  if (it == by_key.end()) {
    // a thread that keeps creating streams (one per request, say) must not keep a workspace for each of them for ever: beyond eight
    // the idle ones are released (after synchronising every device that owns one: their last launches may still be reading them)
    if (by_key.size() >= 8) {
      for (auto& kv : by_key) {
        (void)hipSetDevice(kv.first.dev);
        (void)hipDeviceSynchronize();
        if (kv.second.dev) (void)hipFree(kv.second.dev);
        if (kv.second.pinned) (void)hipHostFree(kv.second.pinned);
        if (kv.second.partials_buf) (void)hipFree(kv.second.partials_buf);
      }
      (void)hipSetDevice(cur.dev);
      by_key.clear();
    }
    it = by_key.emplace(cur, Workspace{}).first;
  }
  Workspace& ws = it->second;
  if (!ws.dev) {
    PST_HIP_CHECK(dev_malloc_retry((void**)&ws.dev, Workspace::kWorkspaceBytes));
    PST_HIP_CHECK(hipHostMalloc((void**)&ws.pinned, Workspace::kPinnedBytes, hipHostMallocDefault));
  }
  return ws;
}

uint8_t* Workspace::partials(size_t bytes) {
  if (bytes > partials_cap) {
    if (partials_buf) (void)hipFree(partials_buf);  // implicit device synchronisation: nothing in flight uses it afterwards
    partials_buf = nullptr;
    partials_cap = 0;
    const size_t want = std::max<size_t>(bytes, 16u << 20);  // pre-sized for the largest launch geometry in use: growth mid-pipeline would stall it
    PST_HIP_CHECK(dev_malloc_retry((void**)&partials_buf, want));
    partials_cap = want;
  }
  return partials_buf;
}

// Device memory comes from HIP's stream-ordered pool (hipMallocAsync): hipMalloc / hipFree of multi-GB buffers cost
// ~100 ms each on this platform, which would dominate `convert()` (allocate + convert + return).  The pool keeps freed
// blocks (release threshold = never), so steady-state allocations are sub-microsecond and stream-ordered.
static bool pool_ready() {
  // per device: every GPU has its own default pool, and each needs its release threshold raised once
  static std::mutex mu;
  static int state[64] = {};  // 0 = not asked yet, 1 = ready, 2 = no pool support
  if (std::getenv("PST_NO_POOL")) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0) {
    state[dev] = 2;
    int supported = 0;
    hipMemPool_t pool;
    uint64_t keep = ~0ull;
    if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, dev) == hipSuccess && supported &&
        hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess)
      state[dev] = 1;
  }
  return state[dev] == 1;
}

// Pool blocks remember the stream they were allocated on.  hipFreeAsync orders the release behind THAT stream's work only, so a
// buffer destroyed while another stream is current (a different thread whose stream is the default one, or after pst_set_stream)
// first waits for the whole device: work enqueued by the asynchronous entry points may still be using the block.
// The owner is remembered with its DEVICE: after pst_set_device(d') a buffer that lives on device d is synchronised and released there.
struct AllocOwner { int device; hipStream_t stream; };
static std::mutex g_alloc_mu;
static std::unordered_map<void*, AllocOwner> g_alloc_stream;

// Hands the blocks the pool holds but nobody uses back to the driver (the rest of the process allocates with hipMalloc and cannot reach them).
void trim_device_pool() {
  int dev = 0;
  hipMemPool_t pool;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess) { (void)hipGetLastError(); return; }
  (void)hipDeviceSynchronize();  // releases enqueued with hipFreeAsync become "unused" when their stream gets there
  if (hipMemPoolTrimTo(pool, 0) != hipSuccess) (void)hipGetLastError();
  // (belt and braces: should the pool count a completed release as unused only when it is next asked for memory, one small request makes it look,
  //  and what it then holds for nobody goes back as well)
  void* nudge = nullptr;
  if (hipMallocAsync(&nudge, 256, nullptr) == hipSuccess) { (void)hipFreeAsync(nudge, nullptr); (void)hipDeviceSynchronize(); }
  else (void)hipGetLastError();
  if (hipMemPoolTrimTo(pool, 0) != hipSuccess) (void)hipGetLastError();
}

// hipMalloc with the same second chance (workspaces, the kNN scratch cache: blocks that live outside the pool)
hipError_t dev_malloc_retry(void** p, size_t bytes) {
  *p = nullptr;
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    *p = nullptr;
    trim_device_pool();
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) *p = nullptr;  // (the caller reads the last error)
  }
  return e;
}

hipError_t dev_alloc_stream(void** p, size_t bytes, hipStream_t s) {
  *p = nullptr;
  if (!pool_ready()) return dev_malloc_retry(p, bytes);
  hipError_t e = hipMallocAsync(p, bytes, s);
  if (e == hipErrorOutOfMemory) {
    // the pool may be holding freed blocks of other sizes (release threshold = never): give them back and ask once more
    (void)hipGetLastError();
    *p = nullptr;
    trim_device_pool();
    e = hipMallocAsync(p, bytes, s);
    if (e != hipSuccess) *p = nullptr;  // (callers that only see a failed helper read the last error: hip_failure, core.hpp)
  }
  if (e == hipSuccess) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    g_alloc_stream[*p] = AllocOwner{dev, s};
  }
  return e;
}
void dev_free_stream(void* p, hipStream_t s) {
  if (!p) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  AllocOwner owner{cur, s};
  bool pooled = false;
  {
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    auto it = g_alloc_stream.find(p);
    if (it != g_alloc_stream.end()) { owner = it->second; g_alloc_stream.erase(it); pooled = true; }
  }
  if (!pooled) { (void)hipFree(p); return; }  // (allocated without the pool: PST_NO_POOL, or a device without pool support)
  if (owner.device != cur) {
    // the block lives on another device: wait for that device's work and hand the block back there; the caller's device stays current
    (void)hipSetDevice(owner.device);
    (void)hipDeviceSynchronize();
    if (hipFree(p) != hipSuccess) (void)hipGetLastError();
    (void)hipSetDevice(cur);
    return;
  }
  // Freed on the stream it was allocated on when that is the current one.  Otherwise the whole device is synchronised first (work the
  // asynchronous entry points enqueued on the owner may still use the block) and after that ANY live stream is a safe place for the
  // release: the current one is used, never the remembered handle -- the caller may have destroyed that stream since.
  if (owner.stream != s) (void)hipDeviceSynchronize();
  if (hipFreeAsync(p, s) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
  }
}

uint8_t* dev_alloc(size_t bytes, uint32_t memkind) {
  if (bytes == 0) return nullptr;
  ensure_device();
  void* p = nullptr;
  if (memkind == PST_MEM_PINNED_HOST) PST_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  else PST_HIP_CHECK(dev_alloc_stream(&p, bytes, current_stream()));
  return (uint8_t*)p;
}
void dev_free(uint8_t* p, uint32_t memkind) {
  if (!p) return;
  if (memkind == PST_MEM_PINNED_HOST) (void)hipHostFree(p);
  else dev_free_stream(p, current_stream());
}

}  // namespace pst

// scratch of the spatial-index builders (device_sort.hpp): same allocator, same PST_NO_POOL / no-pool fallback
namespace pstk {
hipError_t device_malloc_retry(void** p, size_t bytes) { return pst::dev_malloc_retry(p, bytes); }
hipError_t DevBuf::alloc(size_t bytes, hipStream_t stream) {
  release();
  return pst::dev_alloc_stream(&p, bytes ? bytes : 16, stream);
}
void DevBuf::release() {
  if (p) pst::dev_free_stream(p, pst::current_stream());
  p = nullptr;
}
}  // namespace pstk

namespace pst {

PlanEntry identity_entry(const Member& src, const Member& dst) {
  PlanEntry e{};
  e.src_off = (uint32_t)src.offset;
  e.dst_off = (uint32_t)dst.offset;
  e.src_size = (uint32_t)src.size;
  e.dst_size = (uint32_t)dst.size;
  e.ncomp = src.def.datatype.num_components();
  e.src_ct = e.dst_ct = (uint8_t)src.def.datatype.comp_type();
  for (int c = 0; c < 3; ++c) { e.scale[c] = 1.0; e.offset[c] = 0.0; }
  e.mask = ~0ull;
  return e;
}

void check_live(const pst_buffer& b) {
  if (b.is_slice && b.epoch && b.epoch->load(std::memory_order_acquire) != b.epoch_cut)
    throw Error(PST_ERR_INVALID_ARGUMENT, "this slice outlived its parent's storage: the parent buffer was resized or destroyed after the slice was cut "
                                          "(a slice borrows the parent's memory, slice.rs:16-43)");
}
void bump_epoch(pst_buffer& b) {
  if (b.owns && b.epoch) b.epoch->fetch_add(1, std::memory_order_acq_rel);
}

static void check_range(const pst_buffer& b, size_t first, size_t count) {
  if (first + count < first || first + count > b.len)
    throw Error(PST_ERR_RANGE, "range end index " + std::to_string(first + count) + " out of range for buffer of length " + std::to_string(b.len));
}
static size_t slot_of(const pst_buffer& b, const char* name, const pst_datatype* dt) {
  AttributeDef d{not_null(name, "name"), DataType::from_c(dt)};
  int i = b.layout.index_of(d);
  if (i < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of this buffer");
  return (size_t)i;
}

// grow storage to `count` points (contents preserved, new points zero-filled: Vec::resize(_, 0))
void resize_buffer(pst_buffer& b, size_t count, bool zero_fill) {
  if (!b.owns) {
    if (count != b.len) throw Error(PST_ERR_UNSUPPORTED, "ExternalMemoryBuffer is not an OwningBuffer: it cannot be resized");
    return;
  }
  if (count != b.len) bump_epoch(b);  // Vec::resize needs &mut: no slice of this buffer can be alive in the reference
  hipStream_t s = current_stream();
  if (count > b.capacity) {
    // Vec::resize panics with "capacity overflow" when len * size exceeds isize::MAX; here the product must not wrap into a small allocation
    // that the next kernel overruns
    const size_t widest = b.columnar ? [&] { size_t w = 0; for (const auto& m : b.layout.members) w = std::max<size_t>(w, m.size); return w; }() : (size_t)b.layout.size;
    if (widest && count > (size_t)INT64_MAX / widest) throw Error(PST_ERR_OUT_OF_MEMORY, "capacity overflow: " + std::to_string(count) + " points of " + std::to_string(widest) + " bytes");
    // pool allocations are stream-ordered (hipMallocAsync / hipFreeAsync on the current stream): copy and free need no host round trip
    // -- one per column made every growing call (append, filter, voxel grid output) pay several on a loaded host
    const bool ordered = pool_ready() && b.memkind != PST_MEM_PINNED_HOST;
    if (b.columnar) {
      for (size_t a = 0; a < b.columns.size(); ++a) {
        const size_t sz = b.layout.members[a].size;
        uint8_t* fresh = dev_alloc(count * sz, b.memkind);
        if (b.len && fresh) PST_HIP_CHECK(hipMemcpyAsync(fresh, b.columns[a], b.len * sz, hipMemcpyDefault, s));
        if (!ordered) stream_sync(s);
        dev_free(b.columns[a], b.memkind);
        b.columns[a] = fresh;
      }
    } else {
      const size_t sz = b.layout.size;
      uint8_t* fresh = dev_alloc(count * sz, b.memkind);
      if (b.len && fresh) PST_HIP_CHECK(hipMemcpyAsync(fresh, b.data, b.len * sz, hipMemcpyDefault, s));
      if (!ordered) stream_sync(s);
      dev_free(b.data, b.memkind);
      b.data = fresh;
    }
    b.capacity = count;
  }
  if (count > b.len && zero_fill) {
    const size_t extra = count - b.len;
    if (b.columnar) {
      for (size_t a = 0; a < b.columns.size(); ++a) {
        const size_t sz = b.layout.members[a].size;
        if (sz) PST_HIP_CHECK(hipMemsetAsync(b.columns[a] + b.len * sz, 0, extra * sz, s));
      }
    } else if (b.layout.size) {
      PST_HIP_CHECK(hipMemsetAsync(b.data + b.len * b.layout.size, 0, extra * b.layout.size, s));
    }
  }
  b.len = count;
}

}  // namespace pst

pst_buffer::~pst_buffer() {
  if (owns) {
    pst::bump_epoch(*this);
    pst::dev_free(data, memkind);
    for (auto* c : columns) pst::dev_free(c, memkind);
  }
}

using namespace pst;

// RAII temporary device allocation
struct TempDev {
  uint8_t* p = nullptr;
  explicit TempDev(size_t bytes) { p = dev_alloc(bytes, PST_MEM_DEVICE); }
  ~TempDev() { dev_free(p, PST_MEM_DEVICE); }
};

extern "C" {

int pst_device_count(int* out) {
  PST_API_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *not_null(out, "out") = (e == hipSuccess) ? n : 0;
  PST_API_END
}
int pst_set_device(int device) { PST_API_BEGIN ensure_device(); PST_HIP_CHECK(hipSetDevice(device)); PST_API_END }
int pst_set_stream(void* hip_stream) { pst::t_stream = (hipStream_t)hip_stream; return PST_OK; }
int pst_get_stream(void** out_hip_stream) { PST_API_BEGIN *not_null(out_hip_stream, "out_hip_stream") = (void*)pst::t_stream; PST_API_END }
int pst_stream_synchronize(void) { PST_API_BEGIN ensure_device(); stream_sync(current_stream()); PST_API_END }

int pst_buffer_create(const pst_layout* l, uint32_t storage, uint32_t memkind, pst_buffer** out) {
  PST_API_BEGIN
  if (storage > PST_STORAGE_COLUMNAR) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid storage kind");
  if (memkind > PST_MEM_PINNED_HOST) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid memory kind");
  auto b = std::make_unique<pst_buffer>();
  b->layout = not_null(l, "layout")->l;
  check_layout_fits_kernels(b->layout, "pst_buffer_create");
  b->columnar = storage == PST_STORAGE_COLUMNAR;
  b->memkind = memkind;
  if (b->columnar) b->columns.assign(b->layout.members.size(), nullptr);
  *not_null(out, "out") = b.release();
  PST_API_END
}
// ExternalMemoryBuffer wraps any `T: AsRef<[u8]>` in the reference (point_buffer.rs:1479-1497) -- there, ordinary host memory.  Here the kernels must be
// able to reach the bytes: device memory, or host memory the device maps (hipHostMalloc / hipHostRegister / managed).  A pointer the HIP runtime
// does not know -- malloc'ed or mmap'ed host memory passed as if it were the reference's buffer -- would fault the GPU at the first kernel and
// take the process down; it is refused here, with the way out named.  Both ends of the range are asked (an allocation that ends inside it).
// PST_EXTERNAL_UNCHECKED=1 skips the question (memory of another runtime that HIP cannot describe but the device can reach).
// Returns PST_MEM_PINNED_HOST for host memory the device maps, PST_MEM_DEVICE otherwise.
static uint32_t check_device_reaches(const void* p, size_t nbytes, const char* what) {
  static const bool unchecked = [] { const char* v = std::getenv("PST_EXTERNAL_UNCHECKED"); return v && *v == '1'; }();
  if (!nbytes || unchecked) return PST_MEM_DEVICE;
  ensure_device();
  uint32_t kind = PST_MEM_DEVICE;
  for (const uint8_t* q : {(const uint8_t*)p, (const uint8_t*)p + (nbytes - 1)}) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, q) != hipSuccess || at.type == hipMemoryTypeUnregistered) {
      (void)hipGetLastError();
      throw Error(PST_ERR_INVALID_ARGUMENT, std::string(what) + ": the memory is not known to the HIP runtime (ordinary host memory?) -- the kernels could not reach it.  "
                                            "Use device memory, hipHostMalloc / hipHostRegister'ed host memory, or a buffer created with PST_MEM_PINNED_HOST and copy into it");
    }
    if (at.type == hipMemoryTypeHost) kind = PST_MEM_PINNED_HOST;
  }
  return kind;
}

int pst_buffer_wrap_external(const pst_layout* l, void* device_ptr, size_t nbytes, pst_buffer** out) {
  PST_API_BEGIN
  auto b = std::make_unique<pst_buffer>();
  b->layout = not_null(l, "layout")->l;
  check_layout_fits_kernels(b->layout, "pst_buffer_wrap_external");
  const size_t stride = b->layout.size;
  if (stride == 0) { if (nbytes != 0) throw Error(PST_ERR_INVALID_ARGUMENT, "zero-sized PointLayout with non-empty memory"); }
  else if (nbytes % stride != 0)  // ExternalMemoryBuffer::new, point_buffer.rs:1488-1497
    throw Error(PST_ERR_INVALID_ARGUMENT, "external memory size is not a multiple of the point size");
  if (nbytes && !device_ptr) throw Error(PST_ERR_INVALID_ARGUMENT, "device_ptr must not be NULL");
  b->memkind = check_device_reaches(device_ptr, nbytes, "pst_buffer_wrap_external");
  b->owns = false;
  b->epoch.reset();  // (the caller's memory: nothing of ours can move under a slice of it)
  b->data = (uint8_t*)device_ptr;
  b->len = b->capacity = stride ? nbytes / stride : 0;
  *not_null(out, "out") = b.release();
  PST_API_END
}
int pst_buffer_wrap_external_columns(const pst_layout* l, void* const* column_ptrs, size_t len, pst_buffer** out) {
  PST_API_BEGIN
  auto b = std::make_unique<pst_buffer>();
  b->layout = not_null(l, "layout")->l;
  check_layout_fits_kernels(b->layout, "pst_buffer_wrap_external_columns");
  b->columnar = true;
  b->owns = false;
  b->epoch.reset();
  for (size_t a = 0; a < b->layout.members.size(); ++a) {
    void* p = not_null(column_ptrs, "column_ptrs")[a];
    if (len && !p) throw Error(PST_ERR_INVALID_ARGUMENT, "column pointer must not be NULL");
    if (check_device_reaches(p, len * b->layout.members[a].size, "pst_buffer_wrap_external_columns") == PST_MEM_PINNED_HOST) b->memkind = PST_MEM_PINNED_HOST;
    b->columns.push_back((uint8_t*)p);
  }
  b->len = b->capacity = len;
  *not_null(out, "out") = b.release();
  PST_API_END
}
// SliceBuffer::slice / SliceBufferMut::slice_mut, slice.rs:16-43: a non-owning view (owns = false, like ExternalMemoryBuffer) whose
// base addresses are the parent's shifted by `first` points -- every kernel then works on it unchanged
int pst_buffer_slice(const pst_buffer* parent, size_t first, size_t count, pst_buffer** out) {
  PST_API_BEGIN
  check_range(*not_null(parent, "parent"), first, count);
  auto b = std::make_unique<pst_buffer>();
  b->layout = parent->layout;
  b->columnar = parent->columnar;
  b->owns = false;
  b->memkind = parent->memkind;
  b->len = b->capacity = count;
  b->is_slice = true;
  b->epoch = parent->epoch;  // a slice of a slice watches the same owning ancestor; a slice of external memory watches nothing
  b->epoch_cut = parent->is_slice ? parent->epoch_cut : (parent->epoch ? parent->epoch->load(std::memory_order_acquire) : 0);
  if (parent->columnar) {
    for (size_t a = 0; a < parent->layout.members.size(); ++a)
      b->columns.push_back(parent->columns[a] ? parent->columns[a] + first * parent->layout.members[a].size : nullptr);
  } else {
    b->data = parent->data ? parent->data + first * parent->layout.size : nullptr;
  }
  *not_null(out, "out") = b.release();
  PST_API_END
}
int pst_buffer_destroy(pst_buffer* b) { delete b; return PST_OK; }
int pst_buffer_len(const pst_buffer* b, size_t* out) { PST_API_BEGIN *not_null(out, "out") = not_null(b, "buffer")->len; PST_API_END }
int pst_buffer_resize(pst_buffer* b, size_t count) { PST_API_BEGIN resize_buffer(*not_null(b, "buffer"), count, true); PST_API_END }
int pst_buffer_swap(pst_buffer* b, size_t from_index, size_t to_index) {
  PST_API_BEGIN
  not_null(b, "buffer");
  if (!(from_index < b->len)) throw Error(PST_ERR_RANGE, "assertion failed: from_index < self.len()");
  if (!(to_index < b->len)) throw Error(PST_ERR_RANGE, "assertion failed: to_index < self.len()");
  if (from_index == to_index) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  uint8_t* tmp = workspace().dev;  // Workspace::kWorkspaceBytes (1 MiB) of per-thread device scratch; stream order makes the three copies a swap
  auto swap_bytes = [&](uint8_t* base, size_t size) {
    if (!size) return;
    uint8_t *a = base + from_index * size, *c = base + to_index * size;
    const hipMemcpyKind kind = b->memkind == PST_MEM_PINNED_HOST ? hipMemcpyDefault : hipMemcpyDeviceToDevice;
    for (size_t done = 0; done < size; done += Workspace::kWorkspaceBytes) {  // values wider than the scratch (ByteArray attributes) piece by piece
      const size_t piece = std::min(size - done, Workspace::kWorkspaceBytes);
      PST_HIP_CHECK(hipMemcpyAsync(tmp, a + done, piece, kind, s));
      PST_HIP_CHECK(hipMemcpyAsync(a + done, c + done, piece, kind, s));
      PST_HIP_CHECK(hipMemcpyAsync(c + done, tmp, piece, kind, s));
    }
  };
  if (b->columnar) {
    for (size_t a = 0; a < b->columns.size(); ++a) swap_bytes(b->columns[a], b->layout.members[a].size);
  } else {
    swap_bytes(b->data, b->layout.size);
  }
  stream_sync(s);
  PST_API_END
}
int pst_buffer_is_columnar(const pst_buffer* b, int* out) { PST_API_BEGIN *not_null(out, "out") = not_null(b, "buffer")->columnar; PST_API_END }
int pst_buffer_layout(const pst_buffer* b, pst_layout** out_clone) { PST_API_BEGIN *not_null(out_clone, "out") = new pst_layout{not_null(b, "buffer")->layout}; PST_API_END }
int pst_buffer_points_ptr(const pst_buffer* b, void** out) {
  PST_API_BEGIN
  if (not_null(b, "buffer")->columnar) throw Error(PST_ERR_UNSUPPORTED, "buffer is columnar: as_interleaved() is None");
  *not_null(out, "out") = b->data;
  PST_API_END
}
int pst_buffer_column_ptr(const pst_buffer* b, const char* name, const pst_datatype* dt, void** out) {
  PST_API_BEGIN
  if (!not_null(b, "buffer")->columnar) throw Error(PST_ERR_UNSUPPORTED, "buffer is interleaved: as_columnar() is None");
  *not_null(out, "out") = b->columns[slot_of(*b, name, dt)];
  PST_API_END
}

// set_point_range: interleaved = one memcpy (:792-795); columnar = per attribute x per point scatter (:1294-1315), done here
// by staging the records in HBM and running the interleaved->columnar kernel.
int pst_buffer_write_points(pst_buffer* b, size_t first, size_t count, const void* host_src) {
  PST_API_BEGIN
  check_range(*not_null(b, "buffer"), first, count);
  const size_t stride = b->layout.size;
  if (count == 0 || stride == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  if (!b->columnar) {
    PST_HIP_CHECK(hipMemcpyAsync(b->data + first * stride, not_null(host_src, "host_src"), count * stride, hipMemcpyHostToDevice, s));
    stream_sync(s);
  } else {
    TempDev tmp(count * stride);
    PST_HIP_CHECK(hipMemcpyAsync(tmp.p, not_null(host_src, "host_src"), count * stride, hipMemcpyHostToDevice, s));
    std::vector<PlanEntry> es;
    for (size_t a = 0; a < b->layout.members.size(); ++a) {
      PlanEntry e = identity_entry(b->layout.members[a], b->layout.members[a]);
      e.dst_col = col_addr(*b, a, first);
      es.push_back(e);
    }
    execute_entries(true, (uint64_t)(uintptr_t)tmp.p, (uint32_t)stride, false, 0, 0, count, es, true, s);
    stream_sync(s);
  }
  PST_API_END
}
int pst_buffer_read_points(const pst_buffer* b, size_t first, size_t count, void* host_dst) {
  PST_API_BEGIN
  check_range(*not_null(b, "buffer"), first, count);
  const size_t stride = b->layout.size;
  if (count == 0 || stride == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  if (!b->columnar) {
    PST_HIP_CHECK(hipMemcpyAsync(not_null(host_dst, "host_dst"), b->data + first * stride, count * stride, hipMemcpyDeviceToHost, s));
    stream_sync(s);
  } else {
    TempDev tmp(count * stride);
    PST_HIP_CHECK(hipMemsetAsync(tmp.p, 0, count * stride, s));  // padding bytes of the record are unspecified in the reference
    std::vector<PlanEntry> es;
    for (size_t a = 0; a < b->layout.members.size(); ++a) {
      PlanEntry e = identity_entry(b->layout.members[a], b->layout.members[a]);
      e.src_col = col_addr(*b, a, first);
      es.push_back(e);
    }
    execute_entries(false, 0, 0, true, (uint64_t)(uintptr_t)tmp.p, (uint32_t)stride, count, es, true, s);
    PST_HIP_CHECK(hipMemcpyAsync(not_null(host_dst, "host_dst"), tmp.p, count * stride, hipMemcpyDeviceToHost, s));
    stream_sync(s);
  }
  PST_API_END
}
// set_attribute_range / get_attribute_range: columnar = one memcpy (:1340-1347); interleaved = strided per-point copies (:797-820)
int pst_buffer_write_attribute(pst_buffer* b, const char* name, const pst_datatype* dt, size_t first, size_t count, const void* host_src) {
  PST_API_BEGIN
  const size_t slot = slot_of(*not_null(b, "buffer"), name, dt);
  check_range(*b, first, count);
  const Member& m = b->layout.members[slot];
  if (count == 0 || m.size == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  if (b->columnar) {
    PST_HIP_CHECK(hipMemcpyAsync(b->columns[slot] + first * m.size, not_null(host_src, "host_src"), count * m.size, hipMemcpyHostToDevice, s));
    stream_sync(s);
  } else {
    TempDev tmp(count * m.size);
    PST_HIP_CHECK(hipMemcpyAsync(tmp.p, not_null(host_src, "host_src"), count * m.size, hipMemcpyHostToDevice, s));
    PlanEntry e = identity_entry(m, m);
    e.src_col = (uint64_t)(uintptr_t)tmp.p;
    execute_entries(false, 0, 0, true, aos_addr(*b, first), (uint32_t)b->layout.size, count, {e}, true, s);
    stream_sync(s);
  }
  PST_API_END
}
int pst_buffer_read_attribute(const pst_buffer* b, const char* name, const pst_datatype* dt, size_t first, size_t count, void* host_dst) {
  PST_API_BEGIN
  const size_t slot = slot_of(*not_null(b, "buffer"), name, dt);
  check_range(*b, first, count);
  const Member& m = b->layout.members[slot];
  if (count == 0 || m.size == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  if (b->columnar) {
    PST_HIP_CHECK(hipMemcpyAsync(not_null(host_dst, "host_dst"), b->columns[slot] + first * m.size, count * m.size, hipMemcpyDeviceToHost, s));
    stream_sync(s);
  } else {
    TempDev tmp(count * m.size);
    PlanEntry e = identity_entry(m, m);
    e.dst_col = (uint64_t)(uintptr_t)tmp.p;
    execute_entries(true, aos_addr(*b, first), (uint32_t)b->layout.size, false, 0, 0, count, {e}, true, s);
    PST_HIP_CHECK(hipMemcpyAsync(not_null(host_dst, "host_dst"), tmp.p, count * m.size, hipMemcpyDeviceToHost, s));
    stream_sync(s);
  }
  PST_API_END
}

int pst_buffer_synth_fill(pst_buffer* b, uint64_t seed, uint64_t first_index) {
  PST_API_BEGIN
  not_null(b, "buffer");
  if (b->len == 0) return PST_OK;
  ensure_device();
  hipStream_t s = current_stream();
  for (size_t a = 0; a < b->layout.members.size(); ++a) {
    const Member& m = b->layout.members[a];
    pstk::SynthAttr sa{};
    sa.base = b->columnar ? (uint64_t)(uintptr_t)b->columns[a] : (uint64_t)(uintptr_t)b->data + m.offset;
    sa.stride = b->columnar ? m.size : b->layout.size;
    sa.size = (uint32_t)m.size;
    sa.slot = (uint32_t)a;
    sa.kind = m.def.datatype.kind;
    const uint32_t k = sa.kind;
    const std::string& nm = m.def.name;
    if (nm == "Position3D" && (k == PST_VEC3F64 || k == PST_VEC3F32)) sa.special = 1;
    else if (nm == "LASLocalPosition" && k == PST_VEC3I32) sa.special = 2;
    else if (k == PST_U8 && (nm == "ReturnNumber" || nm == "NumberOfReturns")) sa.special = 3;
    else if (k == PST_U8 && (nm == "ScanDirectionFlag" || nm == "EdgeOfFlightLine")) sa.special = 4;
    pstk::launch_synth(sa, b->len, seed, first_index, s);
  }
  PST_HIP_CHECK(hipGetLastError());
  stream_sync(s);
  PST_API_END
}

}  // extern "C"
