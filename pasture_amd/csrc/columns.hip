// K2 / K5 — columnar -> columnar conversion of ONE attribute: wide vector loads, Rust-`as` per component, optional
// closed-set transformation, wide vector stores (gfx950).
//
// Replaces convert_columnar_to_columnar (buffer_conversion.rs:418-487): the same-type arm (:461-469, one bulk
// copy_from_slice) and the converting arm (:438-460, one function-pointer call per value).  Columns are independent
// streams, so every mapping is its own launch.  Each lane moves up to 16 bytes per access on the wider side
// (f64 -> f32 narrowing: 16 B in, 8 B out; u8 -> u32 widening: 4 B in, 16 B out; plain copies: 16 B in, 16 B out as raw
// bytes), four accesses in flight per lane, ONE tile per block (staggered blocks keep HBM reads and writes interleaved —
// measured +8 % over a persistent grid, see stream.hip).  HBM-bound; no MFMA, no LDS.
#include "device_common.hpp"
#include "kernels.hpp"

#include <algorithm>

using namespace pstd;

namespace {

struct ColumnArgs {
  uint64_t src, dst;   // device addresses of component 0
  uint64_t total;      // components
  uint32_t ncomp;      // components per value (for the per-component scale / offset)
  uint32_t xf_kind, xf_pre, shift;
  uint64_t mask;
  double s0, s1, s2, o0, o1, o2;
  uint64_t bounds_partials;  // 0 or address of gridDim.x {min xyz, max xyz} records (D = f64, ncomp = 3)
};

template <typename S, typename D> struct VecOf { static constexpr int value = 16 / (int)(sizeof(S) > sizeof(D) ? sizeof(S) : sizeof(D)); };
constexpr int kUnroll = 4;

template <typename T, int N> struct Pack { T v[N]; };  // N consecutive components, moved as one unaligned vector access
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> load_pack(cgptr_t p) {
  typedef T vec_t __attribute__((ext_vector_type(N)));
  typedef vec_t vec_un __attribute__((aligned(1)));
  Pack<T, N> r;
  if constexpr (N == 1) {
    r.v[0] = load_un<T>(p);
  } else {
    const vec_t x = __builtin_nontemporal_load(reinterpret_cast<const PST_AS_GLOBAL vec_un*>(p));
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = x[i];
  }
  return r;
}
template <typename T, int N>
__device__ __forceinline__ void store_pack(gptr_t p, const Pack<T, N>& r) {
  typedef T vec_t __attribute__((ext_vector_type(N)));
  typedef vec_t vec_un __attribute__((aligned(1)));
  if constexpr (N == 1) {
    store_un<T>(p, r.v[0]);
  } else {
    vec_t x;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = r.v[i];
    __builtin_nontemporal_store(x, reinterpret_cast<PST_AS_GLOBAL vec_un*>(p));
  }
}

template <typename S, typename D>
__device__ __forceinline__ D convert_one(S v, const ColumnArgs& a, uint32_t c) {
  const double sc = pick3(c, a.s0, a.s1, a.s2), of = pick3(c, a.o0, a.o1, a.o2);
  if (a.xf_kind != 0 && a.xf_pre != 0) v = apply_xf<S>(v, a.xf_kind, sc, of, a.shift, a.mask);
  D w = rust_as<D, S>(v);
  if (a.xf_kind != 0 && a.xf_pre == 0) w = apply_xf<D>(w, a.xf_kind, sc, of, a.shift, a.mask);
  return w;
}

template <typename S, typename D>
__global__ __launch_bounds__(kBlock) void column_convert_kernel(const ColumnArgs a) {
  constexpr int VEC = VecOf<S, D>::value;
  constexpr uint64_t kTile = (uint64_t)kBlock * VEC * kUnroll;
  double mn[3] = {kF64Max, kF64Max, kF64Max}, mx[3] = {-kF64Max, -kF64Max, -kF64Max};
  const uint64_t n_tiles = (a.total + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t base = tile * kTile + (uint64_t)threadIdx.x * VEC;
    if ((tile + 1) * kTile <= a.total) {
      Pack<S, VEC> in[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) in[u] = load_pack<S, VEC>(as_global(a.src) + (base + (uint64_t)u * kBlock * VEC) * sizeof(S));
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const uint64_t k0 = base + (uint64_t)u * kBlock * VEC;
        uint32_t c = a.ncomp == 1 ? 0u : (uint32_t)(k0 % a.ncomp);
        Pack<D, VEC> out;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          out.v[i] = convert_one<S, D>(in[u].v[i], a, c);
          if constexpr (std::is_same<D, double>::value) {
            if (a.bounds_partials) {
#pragma unroll
              for (uint32_t cc = 0; cc < 3; ++cc) {
                mn[cc] = __builtin_fmin(mn[cc], cc == c ? out.v[i] : kF64Max);
                mx[cc] = __builtin_fmax(mx[cc], cc == c ? out.v[i] : -kF64Max);
              }
            }
          }
          c = c + 1 == a.ncomp ? 0u : c + 1;
        }
        store_pack<D, VEC>(as_global(a.dst) + k0 * sizeof(D), out);
      }
    } else {
      for (int u = 0; u < kUnroll; ++u) {
        for (int i = 0; i < VEC; ++i) {
          const uint64_t k = base + (uint64_t)u * kBlock * VEC + i;
          if (k < a.total) {
            const uint32_t c = a.ncomp == 1 ? 0u : (uint32_t)(k % a.ncomp);
            const D w = convert_one<S, D>(load_un<S>(as_global(a.src) + k * sizeof(S)), a, c);
            store_un<D>(as_global(a.dst) + k * sizeof(D), w);
            if constexpr (std::is_same<D, double>::value) {
              if (a.bounds_partials) {
                for (uint32_t cc = 0; cc < 3; ++cc) {
                  mn[cc] = __builtin_fmin(mn[cc], cc == c ? w : kF64Max);
                  mx[cc] = __builtin_fmax(mx[cc], cc == c ? w : -kF64Max);
                }
              }
            }
          }
        }
      }
    }
  }
  if constexpr (std::is_same<D, double>::value) {
    if (a.bounds_partials) {
      __shared__ double scratch[(kBlock / 64) * 6];
      block_reduce_minmax<double, 3>(mn, mx, scratch);
      if (threadIdx.x == 0) {
        double* o = (double*)a.bounds_partials + (uint64_t)blockIdx.x * 6;
        o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2]; o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
      }
    }
  }
}

template <typename S, typename D>
unsigned column_grid(uint64_t total) {
  constexpr uint64_t kTile = (uint64_t)kBlock * VecOf<S, D>::value * kUnroll;
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((total + kTile - 1) / kTile, 1u << 22));
}

template <typename S, typename D>
unsigned launch_typed(const ColumnArgs& a, bool launch, hipStream_t stream) {
  const unsigned grid = column_grid<S, D>(a.total);
  if (launch) hipLaunchKernelGGL((column_convert_kernel<S, D>), dim3(grid), dim3(kBlock), pstk::lds_with_resident_cap(0, sizeof(S) >= sizeof(D) ? pstk::kResidentColumn : 0), stream, a);
  return grid;
}

template <typename S>
unsigned launch_src(const ColumnArgs& a, uint32_t dst_ct, bool launch, hipStream_t stream) {
  switch (dst_ct) {
    case CT_U8: return launch_typed<S, uint8_t>(a, launch, stream);
    case CT_I8: return launch_typed<S, int8_t>(a, launch, stream);
    case CT_U16: return launch_typed<S, uint16_t>(a, launch, stream);
    case CT_I16: return launch_typed<S, int16_t>(a, launch, stream);
    case CT_U32: return launch_typed<S, uint32_t>(a, launch, stream);
    case CT_I32: return launch_typed<S, int32_t>(a, launch, stream);
    case CT_U64: return launch_typed<S, uint64_t>(a, launch, stream);
    case CT_I64: return launch_typed<S, int64_t>(a, launch, stream);
    case CT_F32: return launch_typed<S, float>(a, launch, stream);
    default: return launch_typed<S, double>(a, launch, stream);
  }
}

unsigned dispatch_column(const ColumnArgs& a, uint32_t src_ct, uint32_t dst_ct, bool launch, hipStream_t stream) {
  switch (src_ct) {
    case CT_U8: return launch_src<uint8_t>(a, dst_ct, launch, stream);
    case CT_I8: return launch_src<int8_t>(a, dst_ct, launch, stream);
    case CT_U16: return launch_src<uint16_t>(a, dst_ct, launch, stream);
    case CT_I16: return launch_src<int16_t>(a, dst_ct, launch, stream);
    case CT_U32: return launch_src<uint32_t>(a, dst_ct, launch, stream);
    case CT_I32: return launch_src<int32_t>(a, dst_ct, launch, stream);
    case CT_U64: return launch_src<uint64_t>(a, dst_ct, launch, stream);
    case CT_I64: return launch_src<int64_t>(a, dst_ct, launch, stream);
    case CT_F32: return launch_src<float>(a, dst_ct, launch, stream);
    default: return launch_src<double>(a, dst_ct, launch, stream);
  }
}

ColumnArgs make_args(const PlanEntry& e, uint64_t n, uint64_t bounds_partials) {
  ColumnArgs a{};
  a.src = e.src_col;
  a.dst = e.dst_col;
  a.total = n * e.ncomp;
  a.ncomp = e.ncomp;
  a.xf_kind = e.xf_kind;
  a.xf_pre = e.xf_on_source;
  a.shift = e.shift;
  a.mask = e.mask;
  a.s0 = e.scale[0]; a.s1 = e.scale[1]; a.s2 = e.scale[2];
  a.o0 = e.offset[0]; a.o1 = e.offset[1]; a.o2 = e.offset[2];
  a.bounds_partials = bounds_partials;
  return a;
}

}  // namespace

namespace pstk {

// Same datatype and no transformation: move the bytes (any datatype, including the opaque ones) 16 at a time.
static bool is_plain_copy(const PlanEntry& e) { return !e.convert && e.xf_kind == 0; }

unsigned column_launch_grid(const PlanEntry& e, uint64_t n, bool with_bounds) {
  if (is_plain_copy(e) && !with_bounds) {
    ColumnArgs a = make_args(e, n, 0);
    a.total = n * e.src_size;
    return dispatch_column(a, CT_U8, CT_U8, false, nullptr);
  }
  return dispatch_column(make_args(e, n, 0), e.src_ct, e.convert ? e.dst_ct : e.src_ct, false, nullptr);
}

bool launch_column(const PlanEntry& e, uint64_t n, double* bounds_partials, hipStream_t stream) {
  if (n == 0) return true;
  if (is_plain_copy(e) && !bounds_partials) {
    ColumnArgs a = make_args(e, n, 0);
    a.total = n * e.src_size;  // bytes
    a.ncomp = 1;
    dispatch_column(a, CT_U8, CT_U8, true, stream);
  } else {
    dispatch_column(make_args(e, n, (uint64_t)(uintptr_t)bounds_partials), e.src_ct, e.convert ? e.dst_ct : e.src_ct, true, stream);
  }
  return hipGetLastError() == hipSuccess;
}

}  // namespace pstk
