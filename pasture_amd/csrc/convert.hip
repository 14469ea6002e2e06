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
//  * convert_direct_kernel — no LDS; flat per-component global accesses.  Used for columnar<->columnar (already
//    coalesced: consecutive lanes touch consecutive components) and as the universal fall-back (huge records,
//    in-place transform_attribute on interleaved buffers).
//
// HBM-bound integer/byte work: no MFMA.  Compiled with -ffp-contract=off (affine = two roundings).
#include "device_common.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <mutex>

using namespace pstd;

namespace {

__device__ __forceinline__ uint32_t round_up16(uint32_t v) { return (v + 15u) & ~15u; }

// One component: load S, optional pre-transform, `as` D, optional post-transform, store D.
struct XfRegs {  // transformation parameters held in (scalar) registers for the duration of one mapping
  uint32_t kind, pre, shift;
  uint64_t mask;
  double s0, s1, s2, o0, o1, o2;
};
__device__ __forceinline__ XfRegs load_xf(const PlanEntry& e) {
  XfRegs x;
  x.kind = e.xf_kind; x.pre = e.xf_on_source; x.shift = e.shift; x.mask = e.mask;
  x.s0 = e.scale[0]; x.s1 = e.scale[1]; x.s2 = e.scale[2];
  x.o0 = e.offset[0]; x.o1 = e.offset[1]; x.o2 = e.offset[2];
  return x;
}
template <typename S, typename D, typename SP, typename DP>
__device__ __forceinline__ void convert_component(SP sp, DP dp, uint32_t kind, uint32_t pre, double sc, double of, uint32_t shift,
                                                  uint64_t mask) {
  S v = load_un<S>(sp);
  if (kind != 0 && pre != 0) v = apply_xf<S>(v, kind, sc, of, shift, mask);
  D w = rust_as<D, S>(v);
  if (kind != 0 && pre == 0) w = apply_xf<D>(w, kind, sc, of, shift, mask);
  store_un<D>(dp, w);
}

// Split a flat component index into (point, component).
__device__ __forceinline__ void split_comp(uint64_t k, uint32_t ncomp, uint64_t& p, uint32_t& c) {
  if (ncomp == 1) { p = k; c = 0; }
  else if (ncomp == 3) { p = k / 3; c = (uint32_t)(k - 3 * p); }
  else { p = k / ncomp; c = (uint32_t)(k - p * ncomp); }
}
__device__ __forceinline__ void split_comp32(uint32_t k, uint32_t ncomp, uint32_t& p, uint32_t& c) {
  if (ncomp == 1) { p = k; c = 0; }
  else if (ncomp == 3) { p = k / 3; c = k - 3 * p; }
  else { p = k / ncomp; c = k - p * ncomp; }
}

// wave-uniform fetch of one plan entry through the constant address space (scalar loads)
__device__ __forceinline__ PlanEntry fetch_entry(const PlanEntry* entries, uint32_t m) {
  static_assert(sizeof(PlanEntry) % 4 == 0, "PlanEntry must be a whole number of dwords");
  const PST_AS_CONST uint32_t* w = (const PST_AS_CONST uint32_t*)(entries + m);
  PlanEntry e;
  uint32_t* d = reinterpret_cast<uint32_t*>(&e);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(PlanEntry) / 4; ++i) d[i] = w[i];
  return e;
}

// ------------------------------------------------------------------------------------------------------
// direct kernel
// ------------------------------------------------------------------------------------------------------
template <bool SRC_AOS, bool DST_AOS, typename S, typename D>
__device__ __forceinline__ void run_direct(const ConvertHeader& h, const PlanEntry& e) {
  const uint64_t total = h.n * e.ncomp;
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  const XfRegs x = load_xf(e);
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < total; k += step) {
    uint64_t p;
    uint32_t c;
    split_comp(k, e.ncomp, p, c);
    cgptr_t sp;
    gptr_t dp;
    if constexpr (SRC_AOS) sp = as_global(h.src_aos) + p * h.src_stride + e.src_off + c * sizeof(S);
    else sp = as_global(e.src_col) + k * sizeof(S);
    if constexpr (DST_AOS) dp = as_global(h.dst_aos) + p * h.dst_stride + e.dst_off + c * sizeof(D);
    else dp = as_global(e.dst_col) + k * sizeof(D);
    convert_component<S, D>(sp, dp, x.kind, x.pre, pick3(c, x.s0, x.s1, x.s2), pick3(c, x.o0, x.o1, x.o2), x.shift, x.mask);
  }
}

template <bool SRC_AOS, bool DST_AOS>
__global__ __launch_bounds__(kBlock) void convert_direct_kernel(const ConvertHeader h, const PlanEntry* __restrict__ entries) {
  for (uint32_t m = 0; m < h.n_entries; ++m) {
    const PlanEntry e = fetch_entry(entries, m);
    dispatch_ct(e.src_ct, [&](auto s) __attribute__((always_inline)) {
      using S = decltype(s);
      if (!e.convert) {
        run_direct<SRC_AOS, DST_AOS, S, S>(h, e);
      } else {
        dispatch_ct(e.dst_ct, [&](auto d) __attribute__((always_inline)) { run_direct<SRC_AOS, DST_AOS, S, decltype(d)>(h, e); });
      }
    });
  }
}

// ------------------------------------------------------------------------------------------------------
// LDS-tiled kernel
// ------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16-byte vector (dwordx4 / b128 accesses)
typedef PST_AS_GLOBAL u32x4* g4ptr_t;
typedef const PST_AS_GLOBAL u32x4* cg4ptr_t;
typedef PST_AS_LDS u32x4* l4ptr_t;
typedef const PST_AS_LDS u32x4* cl4ptr_t;

// Coalesced global -> LDS copy of the 16-byte aligned span [gbase, gbase + nbytes16).
__device__ __forceinline__ void tile_load(lptr_t lds, cgptr_t gbase, uint32_t nbytes16) {
  cg4ptr_t g = reinterpret_cast<cg4ptr_t>(gbase);
  l4ptr_t l = reinterpret_cast<l4ptr_t>(lds);
  const uint32_t nvec = nbytes16 >> 4;
  for (uint32_t i = threadIdx.x; i < nvec; i += kBlock) l[i] = g[i];
}
// LDS -> global copy of the bytes [mis, mis + nbytes) of the staged span; 16-byte stores for whole chunks, byte stores
// on the two ragged edges so that no byte outside the target range is ever written.
__device__ __forceinline__ void tile_store(clptr_t lds, gptr_t gbase, uint32_t mis, uint32_t nbytes) {
  const uint32_t end = mis + nbytes;
  const uint32_t nvec = (end + 15u) >> 4;
  for (uint32_t i = threadIdx.x; i < nvec; i += kBlock) {
    const uint32_t b0 = i << 4, b1 = b0 + 16;
    if (b0 >= mis && b1 <= end) {
      reinterpret_cast<g4ptr_t>(gbase)[i] = reinterpret_cast<cl4ptr_t>(lds)[i];
    } else {
      const uint32_t lo = b0 > mis ? b0 : mis, hi = b1 < end ? b1 : end;
      for (uint32_t b = lo; b < hi; ++b) gbase[b] = lds[b];
    }
  }
}

template <bool SRC_AOS, bool DST_AOS, typename S, typename D>
__device__ __forceinline__ void run_tile(const ConvertHeader& h, const PlanEntry& e, clptr_t lds_src, lptr_t lds_dst, uint64_t first,
                                         uint32_t cnt) {
  const uint32_t total = cnt * e.ncomp;
  const uint64_t kbase = first * e.ncomp;
  const XfRegs x = load_xf(e);
  for (uint32_t k = threadIdx.x; k < total; k += kBlock) {
    uint32_t p, c;
    split_comp32(k, e.ncomp, p, c);
    const double sc = pick3(c, x.s0, x.s1, x.s2), of = pick3(c, x.o0, x.o1, x.o2);
    if constexpr (SRC_AOS && DST_AOS) {
      convert_component<S, D>(lds_src + (p * h.src_stride + e.src_off + c * (uint32_t)sizeof(S)),
                              lds_dst + (p * h.dst_stride + e.dst_off + c * (uint32_t)sizeof(D)), x.kind, x.pre, sc, of, x.shift, x.mask);
    } else if constexpr (SRC_AOS) {
      convert_component<S, D>(lds_src + (p * h.src_stride + e.src_off + c * (uint32_t)sizeof(S)),
                              as_global(e.dst_col) + (kbase + k) * sizeof(D), x.kind, x.pre, sc, of, x.shift, x.mask);
    } else {
      convert_component<S, D>((cgptr_t)(as_global(e.src_col) + (kbase + k) * sizeof(S)),
                              lds_dst + (p * h.dst_stride + e.dst_off + c * (uint32_t)sizeof(D)), x.kind, x.pre, sc, of, x.shift, x.mask);
    }
  }
}

template <bool SRC_AOS, bool DST_AOS>
__global__ __launch_bounds__(kBlock) void convert_tile_kernel(const ConvertHeader h, const PlanEntry* __restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  lptr_t lds = (lptr_t)lds_raw;
  const uint32_t T = h.tile;
  const uint32_t src_cap = SRC_AOS ? round_up16(T * h.src_stride + 32u) : 0u;
  lptr_t lds_s = lds;
  lptr_t lds_d = lds + src_cap;
  const uint64_t n_tiles = (h.n + T - 1) / T;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t first = tile * T;
    const uint32_t cnt = (uint32_t)((h.n - first) < T ? (h.n - first) : T);
    uint32_t s_mis = 0, d_mis = 0;
    gptr_t g_dst = nullptr;
    if constexpr (SRC_AOS) {
      const uint64_t ga = h.src_aos + first * h.src_stride;
      s_mis = (uint32_t)(ga & 15u);
      tile_load(lds_s, as_global(ga - s_mis), round_up16(s_mis + cnt * h.src_stride));
    }
    if constexpr (DST_AOS) {
      const uint64_t ga = h.dst_aos + first * h.dst_stride;
      d_mis = (uint32_t)(ga & 15u);
      g_dst = as_global(ga - d_mis);
      // record bytes no mapping writes (unmapped attributes, padding) must survive: read-modify-write the tile
      if (!h.dst_fully_covered) tile_load(lds_d, g_dst, round_up16(d_mis + cnt * h.dst_stride));
    }
    __syncthreads();
    for (uint32_t m = 0; m < h.n_entries; ++m) {
      const PlanEntry e = fetch_entry(entries, m);
      dispatch_ct(e.src_ct, [&](auto s) __attribute__((always_inline)) {
        using S = decltype(s);
        if (!e.convert) {
          run_tile<SRC_AOS, DST_AOS, S, S>(h, e, lds_s + s_mis, lds_d + d_mis, first, cnt);
        } else {
          dispatch_ct(e.dst_ct, [&](auto d) __attribute__((always_inline)) {
            run_tile<SRC_AOS, DST_AOS, S, decltype(d)>(h, e, lds_s + s_mis, lds_d + d_mis, first, cnt);
          });
        }
      });
    }
    __syncthreads();
    if constexpr (DST_AOS) {
      tile_store(lds_d, g_dst, d_mis, cnt * h.dst_stride);
      __syncthreads();
    }
  }
}

}  // namespace

namespace pstk {

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

// Plan entries are uploaded into a small ring of device slots; hipMemcpyAsync from pageable host memory stages the
// bytes before returning, so the caller's ConvertPlan may die immediately.  Slot reuse is stream-ordered for the
// common single-stream case and 256 launches deep otherwise.
static const PlanEntry* upload_entries(const ConvertPlan& plan, hipStream_t stream) {
  constexpr int kSlots = 256;
  static PlanEntry* ring = nullptr;
  static unsigned next = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!ring) {
    if (hipMalloc((void**)&ring, sizeof(PlanEntry) * PST_PLAN_MAX_ENTRIES * kSlots) != hipSuccess) return nullptr;
  }
  PlanEntry* slot = ring + (size_t)(next++ % kSlots) * PST_PLAN_MAX_ENTRIES;
  if (hipMemcpyAsync(slot, plan.e, sizeof(PlanEntry) * plan.h.n_entries, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
  return slot;
}

bool launch_convert(const ConvertPlan& plan, bool src_aos, bool dst_aos, bool use_lds, hipStream_t stream) {
  const ConvertHeader& h = plan.h;
  if (h.n == 0 || h.n_entries == 0) return true;
  const PlanEntry* entries = upload_entries(plan, stream);
  if (!entries) return false;
  const int cus = device_cus();
  if (use_lds && (src_aos || dst_aos)) {
    const uint32_t T = h.tile;
    const uint64_t n_tiles = (h.n + T - 1) / T;
    size_t lds_bytes = 0;
    if (src_aos) lds_bytes += ((size_t)T * h.src_stride + 32 + 15) & ~(size_t)15;
    if (dst_aos) lds_bytes += ((size_t)T * h.dst_stride + 32 + 15) & ~(size_t)15;
    // resident blocks per CU are LDS-limited (160 KiB / CU); a grid-stride loop covers the rest
    const uint64_t per_cu = lds_bytes ? std::max<uint64_t>(1, std::min<uint64_t>(8, (160 * 1024) / lds_bytes)) : 8;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, per_cu * cus);
#define PST_LAUNCH_TILE(SA, DA)                                                                                \
  do {                                                                                                         \
    auto kfn = convert_tile_kernel<SA, DA>;                                                                    \
    if (lds_bytes > 64 * 1024)                                                                                 \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), lds_bytes, stream, h, entries);                          \
  } while (0)
    if (src_aos && dst_aos) PST_LAUNCH_TILE(true, true);
    else if (src_aos) PST_LAUNCH_TILE(true, false);
    else PST_LAUNCH_TILE(false, true);
#undef PST_LAUNCH_TILE
    return hipGetLastError() == hipSuccess;
  }
  uint64_t max_comp = 1;
  for (uint32_t m = 0; m < h.n_entries; ++m) max_comp = std::max<uint64_t>(max_comp, plan.e[m].ncomp);
  const uint64_t work = h.n * max_comp;
  const unsigned grid = (unsigned)std::min<uint64_t>((work + kBlock - 1) / kBlock, (uint64_t)cus * 8);
  if (src_aos && dst_aos) hipLaunchKernelGGL((convert_direct_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (src_aos) hipLaunchKernelGGL((convert_direct_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else if (dst_aos) hipLaunchKernelGGL((convert_direct_kernel<false, true>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  else hipLaunchKernelGGL((convert_direct_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, h, entries);
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
