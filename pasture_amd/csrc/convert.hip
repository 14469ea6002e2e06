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


namespace pstk {

// one translation unit per tile kernel (convert_tile_{tt,tf,ft}.hip); 256-thread blocks
void launch_convert_tile_tt(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
void launch_convert_tile_tf(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
void launch_convert_tile_ft(unsigned grid, size_t lds_bytes, hipStream_t stream, const ConvertHeader& h, const PlanEntry* entries);
bool launch_convert_static(const ConvertPlan& plan, bool src_aos, bool dst_aos, unsigned grid, size_t lds_bytes, const PlanEntry* entries, hipStream_t stream);

int device_cus() {
  static int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    }
    return n;
  }();
  return cus;
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
  static uint8_t* ring = nullptr;
  static unsigned next = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
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

// grid of a conversion launch (the caller sizes the fused-bounds partials with it)
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

bool launch_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds, hipStream_t stream) {
  const ConvertHeader& h = plan.h;
  if (h.n == 0 || h.n_entries == 0) return true;
  const PlanEntry* entries = upload_entries(plan, stream);
  if (!entries) return false;
  const unsigned grid = convert_grid(plan, src_aos, dst_aos, use_lds);
  if (use_lds && (src_aos || dst_aos)) {
    const size_t lds_bytes = tile_lds_bytes(h, src_aos, dst_aos);
    // 256-thread blocks: 512 / 1024 measured 15-60 % slower (per-wave interpretation cost is amortised over fewer points)
    if (launch_convert_static(plan, src_aos, dst_aos, grid, lds_bytes, entries, stream)) return hipGetLastError() == hipSuccess;
    if (src_aos && dst_aos) launch_convert_tile_tt(grid, lds_bytes, stream, h, entries);
    else if (src_aos) launch_convert_tile_tf(grid, lds_bytes, stream, h, entries);
    else launch_convert_tile_ft(grid, lds_bytes, stream, h, entries);
    return hipGetLastError() == hipSuccess;
  }
  if (src_aos && dst_aos) hipLaunchKernelGGL((convert_direct_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (src_aos) hipLaunchKernelGGL((convert_direct_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (dst_aos) hipLaunchKernelGGL((convert_direct_kernel<false, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else hipLaunchKernelGGL((convert_direct_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
