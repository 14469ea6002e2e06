// Compile-time conversion plans (convert_tile_static_kernel, convert_kernels.hpp): the mapping lists the reference's own benches and
// tests build, spelled out as constants.  A run-time plan takes the static kernel only if it equals one of these FIELD BY FIELD
// (pstk::match_static_plan, convert_static.hip) -- strides, tile, lane mapping, every entry, the wave that owns each narrow attribute.
#pragma once
#include "convert_kernels.hpp"

namespace pststatic {

using pstd::CT_F32; using pstd::CT_F64; using pstd::CT_I16; using pstd::CT_U16; using pstd::CT_U32; using pstd::CT_U8;

// CustomPointTypeBig (pasture-core/src/test_utils.rs:19-31; buffer_filter_bench.rs, layout_conversion tests), identity plan,
// HashMapBuffer -> VectorBuffer: GpsTime F64 @0 | ColorRGB Vec3u16 @8 | Position3D Vec3f64 @14 | Classification U8 @38 | Intensity I16 @39
template <uint32_t TILE>
struct BigColumnsToRecords_T {
  static constexpr int n = 5;
  static constexpr uint32_t src_stride = 41, dst_stride = 41, tile = TILE, quad = 1, covered = 1;
  __host__ __device__ static constexpr StaticEntry entry(int m) {
    constexpr StaticEntry t[n] = {
        {0, 0, 8, 8, 1, CT_F64, CT_F64, 0, 0},     {8, 8, 6, 6, 3, CT_U16, CT_U16, 0, 0}, {14, 14, 24, 24, 3, CT_F64, CT_F64, 0, 0},
        {38, 38, 1, 1, 1, CT_U8, CT_U8, 0, 1},     {39, 39, 2, 2, 1, CT_I16, CT_I16, 0, 2},
    };
    return t[m];
  }
};

using BigColumnsToRecords = BigColumnsToRecords_T<1024>;  // same-box sweep: 512 / 896 / 1024 within 1 % (0.74-0.75 of peak), 2048 0.45

// the same layout, VectorBuffer -> HashMapBuffer
struct BigRecordsToColumns {
  static constexpr int n = 5;
  static constexpr uint32_t src_stride = 41, dst_stride = 41, tile = 896, quad = 1, covered = 0;
  __host__ __device__ static constexpr StaticEntry entry(int m) { return BigColumnsToRecords::entry(m); }
};

// typed LAS-1 records (LasPointFormat1::layout(), las_types.rs: 43 bytes packed) -> packed records {Position3D, Intensity, Classification}
// (27 bytes): `BufferLayoutConverter::for_layouts(las1, custom)` between two VectorBuffers -- the reader-side "different layout" case
template <uint32_t TILE>
struct Las1RecordsToXyzIC_T {
  static constexpr int n = 3;
  static constexpr uint32_t src_stride = 43, dst_stride = 27, tile = TILE, quad = 1, covered = 1;
  __host__ __device__ static constexpr StaticEntry entry(int m) {
    constexpr StaticEntry t[n] = {
        {0, 0, 24, 24, 3, CT_F64, CT_F64, 0, 0}, {24, 24, 2, 2, 1, CT_U16, CT_U16, 0, 1}, {30, 26, 1, 1, 1, CT_U8, CT_U8, 0, 2},
    };
    return t[m];
  }
};

// layout_conversion_bench.rs:15-39: PointTypeSource {Position3D Vec3f64 @0, Classification U8 @24, Intensity U16 @25, GpsTime F64 @27} (35 bytes)
// -> PointTypeTarget {GpsTime F64 @0, Position3D Vec3f32 @8, Classification U32 @20, Intensity U8 @24} (25 bytes): three `as` casts.
// PAIRING: 0 = records -> columns, 1 = columns -> records, 2 = records -> records (the wave that owns a narrow attribute is decided by its
// columnar-side size, converter.cpp)
template <int PAIRING, uint32_t TILE>
struct BenchSourceToTarget {
  static constexpr int n = 4;
  static constexpr uint32_t src_stride = 35, dst_stride = 25, tile = TILE, quad = 1, covered = PAIRING == 0 ? 0 : 1;
  __host__ __device__ static constexpr StaticEntry entry(int m) {
    constexpr StaticEntry t[n] = {
        {27, 0, 8, 8, 1, CT_F64, CT_F64, 0, 0}, {0, 8, 24, 12, 3, CT_F64, CT_F32, 1, 0}, {24, 20, 1, 4, 1, CT_U8, CT_U32, 1, 1}, {25, 24, 2, 1, 1, CT_U16, CT_U8, 1, 2},
    };
    return t[m];
  }
};

using Las1RecordsToXyzIC = Las1RecordsToXyzIC_T<1024>;  // same-box sweep: 384 0.721, 512 0.713, 768 0.638, 896 0.677, 1024 0.734, 1152 0.699 of peak

}  // namespace pststatic
