// Streaming compaction with the attribute list as a compile-time constant (gfx950): HashMapBuffer::filter_into (point_buffer.rs:1082-1136) for
// ANY columnar source whose attributes sum to at most 64 bytes, into columns or into packed records.
//
// filter.hip's gather kernels issue one load per granule of every SELECTED point.  At the densities a filter runs at every cache line of the
// source columns is touched anyway, so the columns can be read like a conversion reads them (jit_quad.hpp): a lane owns four consecutive points
// and fetches their 4 x size bytes per attribute with 16-byte vector loads, all in flight before the ranks are known; selected points go to LDS
// at their rank -- one span per target column (naturally aligned pieces), or a record tile (a record image in registers, shifted to the dword
// grid of its slot) -- and leave with 16-byte stores.  `filter_stream_body<P>` is instantiated over a plan type P: in-tree for the layouts of the
// reference's filter bench and of typed LAS-0 points (filter.hip), and from source text compiled by hipRTC at run time for every other layout
// (jit.cpp).  The kernel covers FULL 2048-point tiles; the host sends the ragged last tile through the gather kernel.
//
// P:  static constexpr int n;                         // attributes
//     static constexpr bool dst_columns;              // columnar target (else records)
//     static constexpr bool covered;                  // record target: every byte of a record is written by some attribute (else the target span
//                                                     // is read into the LDS tile first and only the attributes' bytes are replaced: padding survives)
//     static constexpr uint32_t dst_stride, cap;      // record size of an interleaved target; points per LDS round (multiple of 16)
//     static constexpr uint32_t size(int k), dst_off(int k);
//     static constexpr uint32_t piece(int k);          // LDS store width of attribute k's values (8 / 4 / 2 / 1): divides the size and the
//                                                     // alignment the host found for the target (column address; record base, stride, offset)
//     static constexpr bool has_pred;                 // the predicate is a device function of the plan's translation unit (round 6: filter's
//                                                     // closure, point_buffer.rs:1064-1136, evaluated on the values the kernel holds anyway): no byte mask
//     template <int W> static uint32_t pred_mask(const uint32_t (&w)[W], uint64_t i0, const double* const (&p)[4]);
//                                                     // has_pred: byte t of the result != 0 <=> the predicate holds for the lane's point t (index i0 + t)
#pragma once
#include "jit_quad.hpp"

namespace pstf {

using namespace pstd;

constexpr int kMaxFilterAttrs = 32;

struct FilterAttr {
  uint64_t src;         // address of the attribute of source point 0
  uint64_t dst;         // columnar target: address of the attribute of target point 0; interleaved: unused
  uint32_t src_stride;  // bytes between consecutive source points
  uint32_t dst_off;     // interleaved target: offset inside the record
  uint32_t unit;        // copy granule: largest of 16/8/4/2/1 dividing the attribute size
  uint32_t cnt;         // granules per value
};

struct FilterArgs {
  const uint8_t* mask;
  const uint32_t* counts;
  const unsigned long long* offsets;
  uint64_t n;
  uint64_t limit;       // never write target points >= limit (num_matches of the reference)
  uint64_t dst_aos;     // interleaved target: address of record 0
  uint32_t dst_stride;  // interleaved target: record size
  uint32_t tile;
  uint32_t n_attrs;
  uint32_t dst_covered;  // interleaved target: attributes cover every byte of the record (no read-modify-write needed)
  uint32_t chunk;        // interleaved target: records per LDS chunk (multiple of 16)
  uint32_t tile0;        // the launch's block 0 is tile `tile0` (the ragged last tile after the streaming kernel took the full ones)
  FilterAttr attrs[kMaxFilterAttrs];
  const double* p[4];    // plans with a fused predicate (P::has_pred): the arrays the expression names p0 .. p3 (null otherwise)
};

constexpr uint32_t kStreamThreads = 512, kStreamTile = 2048;

template <typename P>
__host__ __device__ constexpr uint32_t words_before(int k) {  // a lane's four values of attribute j are size(j) dwords
  uint32_t o = 0;
  for (int j = 0; j < k; ++j) o += P::size(j);
  return o;
}
template <typename P>
__host__ __device__ constexpr uint32_t span_before(int k) {  // LDS offset of column span k (multiples of 16; 16 bytes of slack for the span's phase)
  uint32_t o = 0;
  for (int j = 0; j < k; ++j) o += P::cap * P::size(j) + 16u;
  return o;
}
template <typename P>
__host__ __device__ constexpr uint32_t stream_lds_bytes() {  // columns: the spans + one 16-byte descriptor per span (the chunk list's lookup table)
  return P::dst_columns ? span_before<P>(P::n) + 16u * (uint32_t)P::n : P::cap * P::dst_stride + 64u;
}
// the widest naturally aligned piece a value of `size` bytes splits into when its column starts on a multiple of that piece
__host__ __device__ constexpr uint32_t piece_of(uint32_t size) { return size % 8u == 0 ? 8u : size % 4u == 0 ? 4u : size % 2u == 0 ? 2u : 1u; }

template <uint32_t NB> struct PieceType;
template <> struct PieceType<1> { typedef uint8_t type; };
template <> struct PieceType<2> { typedef uint16_t type; };
template <> struct PieceType<4> { typedef uint32_t type; };
template <> struct PieceType<8> { typedef uint64_t type; };

// BATCH: 16-byte chunks a lane has in flight when the spans leave (4; 2 frees the registers an instantiation needs to stay within 80 -- see
// StreamOccupancy in filter.hip)
template <typename P, uint32_t BATCH = 4>
__device__ __forceinline__ void filter_stream_body(const FilterArgs& a) {
  using pstq::static_for;
  extern __shared__ __attribute__((aligned(16))) uint8_t pstf_lds[];
  lptr_t lds = (lptr_t)pstf_lds;
  __shared__ uint32_t wave_tot[kStreamThreads / 64];
  constexpr int W = (int)words_before<P>(P::n);
  static_assert(P::cap % 16u == 0 && P::cap >= 16u, "column spans start on 16-byte boundaries");
  const uint32_t tile = blockIdx.x;
  const uint64_t first = (uint64_t)tile * kStreamTile;
  // block-uniform by construction: on the scalar unit, so that everything derived from them (span addresses, phases, chunk ranges, the round
  // loop) stays in scalar registers
  const unsigned long long out0_v = a.offsets[tile];
  const uint64_t out0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(out0_v >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)out0_v);
  uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.counts[tile]);
  if (m == 0 || out0 >= a.limit) return;
  if (out0 + m > a.limit) m = (uint32_t)(a.limit - out0);
  const uint32_t p0 = threadIdx.x * 4u;
  uint32_t mw = 0;
  if constexpr (!P::has_pred) mw = load_un<uint32_t>((cgptr_t)((uint64_t)(uintptr_t)a.mask + first) + p0);
  uint32_t w[W];  // the lane's four points, attribute after attribute: 4 x size(k) bytes = size(k) dwords each
  static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
    constexpr int k = decltype(K)::value;
    constexpr uint32_t S = P::size(k), WB = words_before<P>(k);  // (constexpr locals: a plan's functions may be loops the optimiser would not fold)
    pstq::load_words<S, false>((cgptr_t)as_global(a.attrs[k].src) + (first + p0) * S, w, WB);
  });
  if constexpr (P::has_pred) mw = P::template pred_mask<W>(w, first + p0, a.p);  // the same predicate the count pass evaluated, on the values in registers
  // ranks: matches before this lane's points, within the tile
  uint32_t c = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) c += ((mw >> (8u * i)) & 0xFFu) != 0u;
  uint32_t incl = c;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t r0 = incl - c;
  for (uint32_t v = 0; v < wave; ++v) r0 += wave_tot[v];
  for (uint32_t base = 0; base < m; base += P::cap) {
    const uint32_t cm = (m - base) < P::cap ? (m - base) : P::cap;
    if constexpr (P::dst_columns) {
      uint64_t ga[P::n];
      uint32_t mis[P::n];
      static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        constexpr uint32_t S = P::size(k);
        ga[k] = a.attrs[k].dst + (out0 + base) * S;
        mis[k] = (uint32_t)(ga[k] & 15u);
      });
      uint32_t r = r0;
      static_for<0, 4>([&](auto I) __attribute__((always_inline)) {
        constexpr uint32_t i = (uint32_t) decltype(I)::value;
        const bool on = ((mw >> (8u * i)) & 0xFFu) != 0u;
        if (on && r >= base && r < base + cm) {
          const uint32_t j = r - base;
          static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            constexpr uint32_t S = P::size(k), U = P::piece(k), SB = span_before<P>(k);
            lptr_t q = lds + (SB + mis[k] + j * S);
            static_for<0, (int)(S / U)>([&](auto C) __attribute__((always_inline)) {
              constexpr uint32_t cidx = (uint32_t) decltype(C)::value;
              typedef typename PieceType<U>::type PT;
              store_un<PT>(q + cidx * U, (PT)pstq::img_get<4u * words_before<P>(k) + i * S + cidx * U, U>(w));
            });
          });
        }
        r += on ? 1u : 0u;
      });
      // chunk list (below): span k's whole chunks are the list entries [pre[k], pre[k + 1]); entry c of span k lies at LDS byte A_k + 16 c and goes
      // to address B_k + 16 c -- lane k leaves {A_k, B_k} in a table behind the spans, so that a lane finds its span by counting and one LDS read
      uint32_t pre[P::n + 1], vf[P::n];
      pre[0] = 0;
      static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        constexpr uint32_t S = P::size(k), SB = span_before<P>(k), TB = span_before<P>(P::n);
        const uint32_t end = mis[k] + cm * S;
        vf[k] = (mis[k] + 15u) >> 4;
        const uint32_t vl = end >> 4;
        pre[k + 1] = pre[k] + (vl > vf[k] ? vl - vf[k] : 0u);
        if (threadIdx.x == (uint32_t)k) {
          const uint64_t bk = ga[k] - mis[k] + (uint64_t)(vf[k] << 4) - ((uint64_t)pre[k] << 4);
          u32x4 d;
          d.x = SB + (vf[k] << 4) - (pre[k] << 4);
          d.y = (uint32_t)bk;
          d.z = (uint32_t)(bk >> 32);
          d.w = 0;
          *reinterpret_cast<l4ptr_t>(lds + (TB + 16u * (uint32_t)k)) = d;
        }
      });
      __syncthreads();
      // The spans leave as ONE list of 16-byte chunks spread over the block (a span of one-byte values is 64 chunks: a tile_store per span would
      // keep 448 of 512 lanes idle, twelve times over for a LAS layout; even the bench layout's five long spans gain 2 % from the list).  The
      // ragged ends of every span (the bytes before its first and after its last whole chunk) go out byte by byte, one lane per byte -- 32
      // lanes of the first wave per span --, so that no byte outside the target range is written.
      constexpr uint32_t kBatch = BATCH;
      for (uint32_t c0 = threadIdx.x; c0 < pre[P::n]; c0 += kBatch * kStreamThreads) {
        u32x4 v[kBatch];
        uint64_t g[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
          const uint32_t c = c0 + u * kStreamThreads;
          uint32_t k = 0;
          static_for<1, P::n>([&](auto K) __attribute__((always_inline)) { k += c >= pre[decltype(K)::value] ? 1u : 0u; });
          constexpr uint32_t TB = span_before<P>(P::n);
          const u32x4 d = *reinterpret_cast<cl4ptr_t>(lds + (TB + 16u * k));
          g[u] = (((uint64_t)d.z << 32) | d.y) + ((uint64_t)c << 4);
          if (c < pre[P::n]) v[u] = *reinterpret_cast<cl4ptr_t>(lds + (d.x + (c << 4)));
        }
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u)
          if (c0 + u * kStreamThreads < pre[P::n]) __builtin_nontemporal_store(v[u], reinterpret_cast<g4ptr_t>(as_global(g[u])));
      }
      if (threadIdx.x < 32u) {
        const uint32_t t = threadIdx.x & 15u;
        static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
          constexpr int k = decltype(K)::value;
          constexpr uint32_t S = P::size(k), SB = span_before<P>(k);
          const uint32_t end = mis[k] + cm * S;
          const uint32_t head_end = (vf[k] << 4) < end ? (vf[k] << 4) : end;              // bytes [mis, head_end)
          const uint32_t tail_begin = ((end >> 4) << 4) > head_end ? ((end >> 4) << 4) : head_end;  // bytes [tail_begin, end)
          const uint32_t b = threadIdx.x < 16u ? ((mis[k] >> 4) << 4) + t : tail_begin + t;
          const bool mine = threadIdx.x < 16u ? (b >= mis[k] && b < head_end) : (b < end);
          if (mine) as_global(ga[k] - mis[k])[b] = lds[SB + b];
        });
      }
      if (base + P::cap < m) __syncthreads();
    } else {
      constexpr uint32_t STRIDE = P::dst_stride;
      const uint64_t ga = a.dst_aos + (out0 + base) * STRIDE;
      const uint32_t mis = (uint32_t)(ga & 15u);
      if constexpr (!P::covered) {  // padding bytes (and attributes the source lacks) of the target records must survive
        tile_load<(int)kStreamThreads>(lds, as_global(ga - mis), (mis + cm * STRIDE + 15u) & ~15u);
        wait_tile_loads();
        __syncthreads();
      }
      uint32_t r = r0;
      static_for<0, 4>([&](auto I) __attribute__((always_inline)) {
        constexpr uint32_t i = (uint32_t) decltype(I)::value;
        const bool on = ((mw >> (8u * i)) & 0xFFu) != 0u;
        if constexpr (!P::covered) {
          if (on && r >= base && r < base + cm) {
            lptr_t rec = lds + (mis + (r - base) * STRIDE);
            static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
              constexpr int k = decltype(K)::value;
              constexpr uint32_t S = P::size(k), U = P::piece(k);
              static_for<0, (int)(S / U)>([&](auto C) __attribute__((always_inline)) {
                constexpr uint32_t cidx = (uint32_t) decltype(C)::value;
                typedef typename PieceType<U>::type PT;
                constexpr uint32_t DO = P::dst_off(k);
                store_un<PT>(rec + (DO + cidx * U), (PT)pstq::img_get<4u * words_before<P>(k) + i * S + cidx * U, U>(w));
              });
            });
          }
        } else
        if (on && r >= base && r < base + cm) {
          RecordImage<(int)STRIDE> img;
          static_for<0, P::n>([&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            constexpr uint32_t S = P::size(k);
            static_for<0, (int)((S + 7u) / 8u)>([&](auto C) __attribute__((always_inline)) {
              constexpr uint32_t o = 8u * (uint32_t) decltype(C)::value, nb = (S - o) < 8u ? (S - o) : 8u;
              pstq::img_put<false, P::dst_off(k) + o, nb>(img.w, pstq::img_get<4u * words_before<P>(k) + i * S + o, nb>(w));
            });
          });
          img.store_aligned(lds + (mis + (r - base) * STRIDE));
        }
        r += on ? 1u : 0u;
      });
      __syncthreads();
      tile_store<(int)kStreamThreads>(lds, as_global(ga - mis), mis, cm * STRIDE);
      if (base + P::cap < m) __syncthreads();
    }
  }
}

}  // namespace pstf
