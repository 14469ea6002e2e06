// Multi-GPU boundary: the ONLY exchange of the sharded path is the global AABB -- ONE all-reduce of the
// 6-double record over RCCL (xGMI between the GPUs of a node).  The reference has no distributed code; its index-range processing
// (convert_into_range buffer_conversion.rs:292, 1 MiB chunks raw_readers.rs:309-349) is what shards without a data-path collective,
// and calculate_bounds' seeds (bounds.rs:31-32) are the identities an empty shard contributes.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already holds an RCCL instance -- torch.distributed's -- keeps using
// that one (same SONAME), and the library stays loadable where RCCL is absent; the entry points then fail with PST_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only (ncclComm_t, ncclUniqueId, ncclFloat64, ncclMin): the functions are bound with dlsym below

#include <mutex>
#include <vector>

#include "device_common.hpp"
#include "kernels.hpp"
#include "runtime.hpp"

using namespace pst;

namespace {

struct Rccl {
  void* lib = nullptr;
  // pointer types taken from the header's own declarations: a signature change in RCCL is a compile error here, not a silent ABI mismatch
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    auto sym = [&](const char* n) { return dlsym(r.lib, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd)
    throw Error(PST_ERR_UNSUPPORTED, "RCCL (librccl.so.1) is not available in this process: the multi-GPU entry points need it");
  return r;
}

void check(ncclResult_t rc, const char* what) {
  if (rc != ncclSuccess) {
    const Rccl& r = rccl();
    throw Error(PST_ERR_HIP, std::string(what) + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error ") + " (" + std::to_string((int)rc) + ")");
  }
}

// restores the caller's device and closes an open RCCL group when a multi-device call unwinds
struct MultiGuard {
  int prev = 0;
  bool group_open = false;
  MultiGuard() { PST_HIP_CHECK(hipGetDevice(&prev)); }
  ~MultiGuard() {
    if (group_open) (void)rccl().GroupEnd();
    (void)hipSetDevice(prev);
  }
};

// {min xyz, max xyz} <-> {min xyz, -max xyz}: ONE ncclMin all-reduce of 6 doubles then folds minima and maxima together
__global__ void negate_max_kernel(double* rec6) {
  if (threadIdx.x < 3) rec6[3 + threadIdx.x] = -rec6[3 + threadIdx.x];
}

}  // namespace

struct pst_comm {
  std::vector<ncclComm_t> comms;  // one per device this handle drives (init_rank: exactly one)
  std::vector<int> devices;
  int n_ranks = 0;
  int rank = -1;                  // -1: single-process handle over devices[]
};

extern "C" {

int pst_comm_unique_id(pst_comm_id* out_id) {
  PST_API_BEGIN
  static_assert(sizeof(pst_comm_id) == sizeof(ncclUniqueId), "pst_comm_id must be RCCL's 128 opaque bytes");
  ensure_device();
  ncclUniqueId id;
  check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(not_null(out_id, "out_id"), &id, sizeof(id));
  PST_API_END
}

int pst_comm_init_rank(int n_ranks, int rank, const pst_comm_id* id, pst_comm** out) {
  PST_API_BEGIN
  not_null(out, "out");
  *out = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw Error(PST_ERR_INVALID_ARGUMENT, "pst_comm_init_rank: need 0 <= rank < n_ranks");
  ensure_device();
  ncclUniqueId uid;
  memcpy(&uid, not_null(id, "id"), sizeof(uid));
  auto c = std::unique_ptr<pst_comm>(new pst_comm());
  int dev = 0;
  PST_HIP_CHECK(hipGetDevice(&dev));
  c->comms.resize(1);
  c->devices = {dev};
  c->n_ranks = n_ranks;
  c->rank = rank;
  check(rccl().CommInitRank(&c->comms[0], n_ranks, uid, rank), "ncclCommInitRank");
  *out = c.release();
  PST_API_END
}

int pst_comm_init(int n_gpus, pst_comm** out) {
  PST_API_BEGIN
  not_null(out, "out");
  *out = nullptr;
  ensure_device();
  int have = 0;
  PST_HIP_CHECK(hipGetDeviceCount(&have));
  if (n_gpus < 1 || n_gpus > have)
    throw Error(PST_ERR_NO_DEVICE, "pst_comm_init: " + std::to_string(n_gpus) + " GPUs requested, " + std::to_string(have) + " present");
  auto c = std::unique_ptr<pst_comm>(new pst_comm());
  c->comms.resize((size_t)n_gpus);
  for (int d = 0; d < n_gpus; ++d) c->devices.push_back(d);
  c->n_ranks = n_gpus;
  check(rccl().CommInitAll(c->comms.data(), n_gpus, c->devices.data()), "ncclCommInitAll");
  *out = c.release();
  PST_API_END
}

int pst_comm_size(const pst_comm* comm, int* out_n_ranks) {
  PST_API_BEGIN
  *not_null(out_n_ranks, "out_n_ranks") = not_null(comm, "comm")->n_ranks;
  PST_API_END
}

int pst_comm_destroy(pst_comm* comm) {
  PST_API_BEGIN
  if (comm) {
    for (ncclComm_t c : comm->comms) if (c) (void)rccl().CommDestroy(c);
    delete comm;
  }
  PST_API_END
}

// in place on the current stream: encode, ONE ncclAllReduce(6 x f64, ncclMin), decode.  Stream-ordered, no host synchronisation.
int pst_bounds_allreduce(pst_comm* comm, double* device_rec6) {
  PST_API_BEGIN
  not_null(comm, "comm");
  not_null(device_rec6, "device_rec6");
  if (comm->comms.size() != 1)
    throw Error(PST_ERR_INVALID_ARGUMENT, "pst_bounds_allreduce: this handle drives several devices; use pst_bounds_allreduce_multi");
  int dev = -1;
  PST_HIP_CHECK(hipGetDevice(&dev));
  if (dev != comm->devices[0])
    throw Error(PST_ERR_INVALID_ARGUMENT, "pst_bounds_allreduce: the communicator was created on device " + std::to_string(comm->devices[0]) +
                                              ", the calling thread's current device is " + std::to_string(dev));
  hipStream_t s = current_stream();
  // a record the library itself leaves as {min, -max} (pst_bounds_record_set_form: written in that form by the producing kernel's last fold) needs
  // nothing around the collective and stays in that form: the exposed exchange is ONE launch
  const bool encoded = pstk::bounds_record_negates_max(device_rec6);
  if (!encoded) hipLaunchKernelGGL(negate_max_kernel, dim3(1), dim3(64), 0, s, device_rec6);
  check(rccl().AllReduce(device_rec6, device_rec6, 6, ncclFloat64, ncclMin, comm->comms[0], s), "ncclAllReduce");
  if (!encoded) hipLaunchKernelGGL(negate_max_kernel, dim3(1), dim3(64), 0, s, device_rec6);
  PST_HIP_CHECK(hipGetLastError());
  PST_API_END
}

// From now on (form = 1) every AABB record the library is asked to leave at this device address -- pst_calculate_bounds_async,
// pst_converter_convert_into_range_with_bounds_async -- is written as {min xyz, -max xyz} by the producing kernel's own last fold, and
// pst_bounds_allreduce on it is the collective alone; form = 0 forgets the address.  An empty shard's record is then {+f64::MAX x 6} (bounds.rs:31-32).
int pst_bounds_record_set_form(double* device_rec6, int form) {
  PST_API_BEGIN
  pstk::set_bounds_record_form(not_null(device_rec6, "device_rec6"), form);
  PST_API_END
}

// single-process handle (pst_comm_init): device_recs[d] is the record of device d, streams[d] (nullable array: default streams) its stream
int pst_bounds_allreduce_multi(pst_comm* comm, double* const* device_recs, void* const* streams) {
  PST_API_BEGIN
  not_null(comm, "comm");
  not_null(device_recs, "device_recs");
  const size_t n = comm->comms.size();
  const Rccl& r = rccl();
  // every pointer is validated (null, and which device owns it) BEFORE the first hipSetDevice / ncclGroupStart
  for (size_t d = 0; d < n; ++d) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, not_null(device_recs[d], "device_recs[d]")) != hipSuccess || at.device != comm->devices[d]) {
      (void)hipGetLastError();
      throw Error(PST_ERR_INVALID_ARGUMENT, "pst_bounds_allreduce_multi: device_recs[" + std::to_string(d) + "] is not memory of device " + std::to_string(comm->devices[d]));
    }
  }
  MultiGuard guard;
  for (size_t d = 0; d < n; ++d) {
    PST_HIP_CHECK(hipSetDevice(comm->devices[d]));
    if (!pstk::bounds_record_negates_max(device_recs[d])) hipLaunchKernelGGL(negate_max_kernel, dim3(1), dim3(64), 0, streams ? (hipStream_t)streams[d] : nullptr, device_recs[d]);
  }
  check(r.GroupStart(), "ncclGroupStart");
  guard.group_open = true;
  for (size_t d = 0; d < n; ++d)
    check(r.AllReduce(device_recs[d], device_recs[d], 6, ncclFloat64, ncclMin, comm->comms[d], streams ? (hipStream_t)streams[d] : nullptr), "ncclAllReduce");
  guard.group_open = false;
  check(r.GroupEnd(), "ncclGroupEnd");
  for (size_t d = 0; d < n; ++d) {
    PST_HIP_CHECK(hipSetDevice(comm->devices[d]));
    if (!pstk::bounds_record_negates_max(device_recs[d])) hipLaunchKernelGGL(negate_max_kernel, dim3(1), dim3(64), 0, streams ? (hipStream_t)streams[d] : nullptr, device_recs[d]);
  }
  PST_HIP_CHECK(hipGetLastError());
  PST_API_END
}

}  // extern "C"
