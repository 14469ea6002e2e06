// K2 / K3 / K3' / K5 — generic attribute gather -> Rust-`as` convert -> transform -> scatter kernels (gfx950).
//
// Replaces the four attribute-major CPU loops of BufferLayoutConverter
// (pasture-core/src/layout/conversion/buffer_conversion.rs:418-487 columnar->columnar, :489-544 columnar->interleaved,
//  :546-604 interleaved->columnar, :606-662 interleaved->interleaved) and the per-value converter table
// (attribute_conversion.rs:184-343).  One launch handles every mapping of the plan; the interleaved side is read /
// written ONCE per call (the reference re-streams it once per mapping).
//
// Two bodies:
//  * convert_tile_kernel  — interleaved records are staged through LDS tiles with 16-byte coalesced global accesses;
//    attributes are then picked out of / assembled in LDS at byte granularity (packed(1) layouts have no alignment).
//    The columnar side is accessed in chunks of >= 4 bytes per lane (four u8 / two u16 values are packed into one
//    dword), so a wave always moves >= 256 contiguous bytes per column instruction.
//  * convert_direct_kernel — no LDS; flat per-component global accesses.  Used for columnar<->columnar (already
//    coalesced: consecutive lanes touch consecutive components) and as the universal fall-back (huge records,
//    in-place transform_attribute on interleaved buffers).
// Entries flagged `.bounds` fold the Vec3f64 values they write into a per-block AABB record (fused calculate_bounds).
//
// HBM-bound integer/byte work: no MFMA.  Compiled with -ffp-contract=off (affine = two roundings).
#include "convert_kernels.hpp"
#include "jit.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace pstk {

// one translation unit per tile kernel (convert_tile_{tt,tf,ft}.hip); 256-thread blocks
void launch_convert_tile_tt(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
void launch_convert_tile_tf(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
void launch_convert_tile_ft(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
bool launch_convert_static(const std::string& source, uint32_t* tile, bool tile_only, unsigned grid, const ConvertHeader& h, const PlanEntry* entries, hipStream_t stream);

// (per device: pst_set_device may move a thread to a GPU with another CU count)
int device_cus() {
  static std::mutex mu;
  static std::map<int, int> by_device;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  auto it = by_device.find(dev);
  if (it != by_device.end()) return it->second;
  int n = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
  by_device[dev] = n;
  return n;
}

static long env_long(const char* name, long dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::strtol(v, nullptr, 10) : dflt;
}

// Plan entries are uploaded into a small ring of device slots; hipMemcpyAsync from pageable host memory stages the
// bytes before returning, so the caller's ConvertPlan may die immediately.  Slot reuse is stream-ordered for the
// common single-stream case and 256 launches deep otherwise.
static const PlanEntry* upload_entries(const ConvertPlan& plan, hipStream_t stream) {
  constexpr int kSlots = 256;
  struct Ring { uint8_t* base = nullptr; unsigned next = 0; };
  static std::map<int, Ring> rings;  // one ring per device: a kernel on device 1 cannot read plan entries that live in device 0's memory
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  Ring& r = rings[dev];
  uint8_t*& ring = r.base;
  unsigned& next = r.next;
  constexpr size_t kSlotBytes = sizeof(plan.e) + sizeof(plan.masks);  // entries immediately followed by the wave masks
  static_assert(offsetof(ConvertPlan, masks) == offsetof(ConvertPlan, e) + sizeof(plan.e), "masks must follow the entries");
  if (!ring) {
    if (hipMalloc((void**)&ring, kSlotBytes * kSlots) != hipSuccess) return nullptr;
  }
  uint8_t* slot = ring + (size_t)(next++ % kSlots) * kSlotBytes;
  if (hipMemcpyAsync(slot, plan.e, kSlotBytes, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
  return (const PlanEntry*)slot;
}

static size_t tile_lds_bytes(const ConvertHeader& h, bool src_aos, bool dst_aos) {
  size_t lds_bytes = 0;
  if (src_aos && !(dst_aos && h.in_place)) lds_bytes += ((size_t)h.tile * h.src_stride + 32 + 15) & ~(size_t)15;
  if (dst_aos) lds_bytes += ((size_t)h.tile * h.dst_stride + 32 + 15) & ~(size_t)15;
  return lds_bytes;
}

// grid of an interpreted conversion launch
unsigned convert_grid(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds) {
  const ConvertHeader& h = plan.h;
  const int cus = device_cus();
  if (use_lds && (src_aos || dst_aos)) {
    const uint64_t n_tiles = (h.n + h.tile - 1) / h.tile;
    // one tile per block: staggered blocks keep HBM reads and writes interleaved (see stream.hip); cap for huge inputs
    static const long cap = env_long("PST_TILE_GRID_CAP", 1 << 22);
    return (unsigned)((std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)cap)) + 7) / 8 * 8);  // multiple of 8: xcd_block_id()
  }
  uint64_t max_comp = 1;
  for (uint32_t m = 0; m < h.n_entries; ++m) max_comp = std::max<uint64_t>(max_comp, plan.e[m].ncomp);
  const uint64_t work = h.n * max_comp;
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((work + kBlock - 1) / kBlock, (uint64_t)cus * 8));
}

// Upper bound of the AABB records a conversion launch (or the pair specialised kernel + interpreted tail) writes: the caller sizes the
// fused-bounds partials with it; launch_convert reports the number actually written.
unsigned convert_max_records(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds) {
  const uint64_t jit_tiles = plan.h.n / 256 + 8;  // the smallest tile of the specialised kernels
  return convert_grid(plan, src_aos, dst_aos, use_lds) + (unsigned)std::min<uint64_t>(jit_tiles, 1u << 24) + 8u;
}

// which kernel families the calling thread's last conversion call used (pst_last_plan_kinds)
static thread_local uint32_t t_plan_kinds = 0;
void reset_plan_kinds() { t_plan_kinds = 0; }
void note_plan_kind(uint32_t kind) { t_plan_kinds |= 1u << kind; }
uint32_t plan_kinds() { return t_plan_kinds; }
void note_slow_family(const char* what, uint64_t n_points, const char* why) {
  static const bool quiet = [] { const char* v = std::getenv("PST_QUIET"); return v && *v && *v != '0'; }();
  if (quiet || n_points < ((uint64_t)1 << 20)) return;
  static std::mutex mu;
  static std::set<std::string> seen;
  std::lock_guard<std::mutex> lock(mu);
  if (!seen.insert(std::string(what) + '|' + why).second) return;
  fprintf(stderr, "pasture_amd: note: %s of %llu points ran on a fall-back kernel family (%s); pst_last_plan_kinds reports the family of every call, "
                  "pst_converter_prepare compiles the plan-specialised kernel ahead of the first call, PST_QUIET=1 silences this note (printed once)\n",
          what, (unsigned long long)n_points, why);
}

// the same plan for the points [first, n) of its range
static ConvertPlan plan_tail(const ConvertPlan& plan, bool src_aos, bool dst_aos, uint64_t first) {
  ConvertPlan t = plan;
  t.h.n = plan.h.n - first;
  if (src_aos) t.h.src_aos += first * plan.h.src_stride;
  if (dst_aos) t.h.dst_aos += first * plan.h.dst_stride;
  for (uint32_t m = 0; m < plan.h.n_entries; ++m) {
    if (!src_aos) t.e[m].src_col += first * plan.e[m].src_size;
    if (!dst_aos) t.e[m].dst_col += first * plan.e[m].dst_size;
  }
  return t;
}

static bool launch_interpreted(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds, hipStream_t stream, unsigned* n_records) {
  const ConvertHeader& h = plan.h;
  const PlanEntry* entries = upload_entries(plan, stream);
  if (!entries) return false;
  const unsigned grid = convert_grid(plan, src_aos, dst_aos, use_lds);
  if (n_records) *n_records = grid;
  if (use_lds && (src_aos || dst_aos)) {
    const size_t lds_bytes = tile_lds_bytes(h, src_aos, dst_aos);
    // 256-thread blocks: 512 / 1024 measured 15-60 % slower (per-wave interpretation cost is amortised over fewer points)
    note_plan_kind(PST_PLAN_INTERPRETED);
    {
      pstjit::QuadSpec spec;
      const char* why = pstjit::mode() == pstjit::Mode::Off ? "the run-time compiler is switched off: PST_JIT=0"
                        : !pstjit::spec_from_plan(plan, src_aos, dst_aos, &spec) ? "this plan has no specialised form: more than 24 mappings, records beyond 224 bytes per point pair, or unaligned record bases"
                        : pstjit::mode() == pstjit::Mode::Async ? "its specialised kernel is still compiling in the background: later calls take it"
                                                                : "its specialised kernel failed to compile";
      note_slow_family("a conversion", h.n, why);
    }
    if (src_aos && dst_aos) launch_convert_tile_tt(grid, lds_bytes, stream, h, entries);
    else if (src_aos) launch_convert_tile_tf(grid, lds_bytes, stream, h, entries);
    else launch_convert_tile_ft(grid, lds_bytes, stream, h, entries);
    return hipGetLastError() == hipSuccess;
  }
  note_plan_kind(PST_PLAN_DIRECT);
  if (src_aos || dst_aos) note_slow_family("a conversion", h.n, "records too large for an LDS tile: strided accesses without staging");
  if (src_aos && dst_aos) hipLaunchKernelGGL((convert_direct_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (src_aos) hipLaunchKernelGGL((convert_direct_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (dst_aos) hipLaunchKernelGGL((convert_direct_kernel<false, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else hipLaunchKernelGGL((convert_direct_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  return hipGetLastError() == hipSuccess;
}

// Plan-specialised kernel (jit.cpp) over the full tiles of the range; `done` = points it covered (0: not taken).
static bool launch_specialised(const ConvertPlan& plan, bool src_aos, bool dst_aos, hipStream_t stream, unsigned* n_records, uint64_t* done) {
  *done = 0;
  const pstjit::Mode mode = pstjit::mode();
  if (mode == pstjit::Mode::Off) return true;
  pstjit::QuadSpec spec;
  if (!pstjit::spec_from_plan(plan, src_aos, dst_aos, &spec)) return true;
  const std::string source = pstjit::spec_source(spec);
  // 1. the in-tree instantiations for the reference's bench layouts (a warm cache); 2. the run-time compiler's cache
  uint32_t tile = 0;
  pstjit::Kernel k;
  const bool in_tree = launch_convert_static(source, &tile, true, 0, plan.h, nullptr, stream);
  if (!in_tree) {
    // small calls never start a compilation; a kernel some larger call (or pst_converter_prepare) had compiled is used whatever the size
    const pstjit::Acquire how = mode == pstjit::Mode::Sync ? pstjit::Acquire::Wait
                                : plan.h.n >= pstjit::min_points() ? pstjit::Acquire::Enqueue : pstjit::Acquire::IfReady;
    if (!pstjit::acquire(spec, source, how, &k)) return true;  // not ready (or failed): interpret
    tile = k.tile;
  }
  const uint64_t n_tiles = plan.h.n / tile;
  if (n_tiles == 0 || n_tiles > (1ull << 30)) return true;
  const PlanEntry* entries = upload_entries(plan, stream);
  if (!entries) return false;
  ConvertHeader h = plan.h;
  h.n = n_tiles * tile;
  const unsigned grid = (unsigned)((n_tiles + 7) / 8 * 8);
  if (in_tree) {
    launch_convert_static(source, &tile, false, grid, h, entries, stream);
    if (hipGetLastError() != hipSuccess) return false;
    note_plan_kind(PST_PLAN_STATIC);
  } else {
    void* args[] = {(void*)&h, (void*)&entries};
    if (hipModuleLaunchKernel(k.fn, grid, 1, 1, k.blk, 1, 1, lds_with_resident_cap(k.lds_bytes, kResidentQuad), stream, args, nullptr) != hipSuccess) return false;
    note_plan_kind(PST_PLAN_JIT);
  }
  if (n_records) *n_records = grid;
  *done = h.n;
  return true;
}

// A plan with fused expression entries (PST_XF_EXPR; plan.expr_texts): the plan-specialised kernel or nothing -- the interpreter cannot evaluate an
// expression.  Compiled in the calling thread (an expression has always been compiled at its first use: its syntax errors are that call's
// error).  *done = the points covered (full tiles; 0 = this plan has no specialised form, or, with *error set, its kernel does not compile); the
// caller runs the rest -- the ragged tail, or everything -- through expr.cpp's strided kernel and the interpreter.
bool launch_convert_fused_expressions(const ConvertPlan& plan, bool src_aos, bool dst_aos, hipStream_t stream, uint64_t* done, std::string* error) {
  *done = 0;
  if (pstjit::mode() == pstjit::Mode::Off) return true;
  pstjit::QuadSpec spec;
  if (!pstjit::spec_from_plan(plan, src_aos, dst_aos, &spec)) return true;
  pstjit::Kernel k;
  if (!pstjit::acquire(spec, pstjit::spec_source(spec), pstjit::Acquire::Wait, &k, error)) return true;
  const uint64_t n_tiles = plan.h.n / k.tile;
  if (n_tiles == 0 || n_tiles > (1ull << 30)) return true;
  const PlanEntry* entries = upload_entries(plan, stream);
  if (!entries) return false;
  ConvertHeader h = plan.h;
  h.n = n_tiles * k.tile;
  const unsigned grid = (unsigned)((n_tiles + 7) / 8 * 8);
  void* args[] = {(void*)&h, (void*)&entries};
  if (hipModuleLaunchKernel(k.fn, grid, 1, 1, k.blk, 1, 1, lds_with_resident_cap(k.lds_bytes, kResidentQuad), stream, args, nullptr) != hipSuccess) return false;
  note_plan_kind(PST_PLAN_JIT);
  *done = h.n;
  return true;
}

// Is a plan-specialised kernel (in-tree instantiation or run-time compiled) at hand for this plan?  A missing one is queued for the compiler
// thread (or compiled here in PST_JIT=sync), exactly as a launch would.  converter.cpp asks before it prefers the generic path over a
// format-specialised LAS kernel that the plan-specialised kernels have overtaken (round 4).
bool convert_specialised_ready(const ConvertPlan& plan, bool src_aos, bool dst_aos) {
  const pstjit::Mode mode = pstjit::mode();
  if (mode == pstjit::Mode::Off) return false;
  pstjit::QuadSpec spec;
  if (!pstjit::spec_from_plan(plan, src_aos, dst_aos, &spec)) return false;
  const std::string source = pstjit::spec_source(spec);
  uint32_t tile = 0;
  if (launch_convert_static(source, &tile, true, 0, plan.h, nullptr, nullptr)) return true;
  pstjit::Kernel k;
  const pstjit::Acquire how = mode == pstjit::Mode::Sync ? pstjit::Acquire::Wait
                              : plan.h.n >= pstjit::min_points() ? pstjit::Acquire::Enqueue : pstjit::Acquire::IfReady;
  return pstjit::acquire(spec, source, how, &k);
}

bool launch_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds, hipStream_t stream, unsigned* n_records) {
  const ConvertHeader& h = plan.h;
  if (n_records) *n_records = 0;
  if (h.n == 0 || h.n_entries == 0) return true;
  if (use_lds && (src_aos || dst_aos)) {
    unsigned rec = 0;
    uint64_t done = 0;
    if (!launch_specialised(plan, src_aos, dst_aos, stream, &rec, &done)) return false;
    if (done == h.n) { if (n_records) *n_records = rec; return true; }
    if (done > 0) {  // the ragged tail (less than one tile) is interpreted
      ConvertPlan t = plan_tail(plan, src_aos, dst_aos, done);
      if (t.h.bounds_partials) t.h.bounds_partials += (uint64_t)rec * 6 * sizeof(double);
      unsigned rec_tail = 0;
      if (!launch_interpreted(t, src_aos, dst_aos, use_lds, stream, &rec_tail)) return false;
      if (n_records) *n_records = rec + rec_tail;
      return true;
    }
  }
  return launch_interpreted(plan, src_aos, dst_aos, use_lds, stream, n_records);
}

// Compile (or fetch) the specialised kernel this plan would take; true when a later launch_convert of the same plan shape will use it.
bool prepare_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, std::string* error, bool* in_tree) {
  if (in_tree) *in_tree = false;
  if (pstjit::mode() == pstjit::Mode::Off) { if (error) *error = "PST_JIT=0"; return false; }
  pstjit::QuadSpec spec;
  if (!pstjit::spec_from_plan(plan, src_aos, dst_aos, &spec)) { if (error) *error = "plan not eligible for a specialised kernel"; return false; }
  const std::string source = pstjit::spec_source(spec);
  uint32_t tile = 0;
  if (launch_convert_static(source, &tile, true, 0, plan.h, nullptr, nullptr)) { if (in_tree) *in_tree = true; return true; }
  pstjit::Kernel k;
  return pstjit::acquire(spec, source, pstjit::Acquire::Wait, &k, error);
}

}  // namespace pstk
