// Kernel argument structures shared by the host runtime and the HIP kernels.
// A ConvertPlan is the flattened form of the reference's Vec<AttributeMapping> (buffer_conversion.rs:41-55, 98-102)
// for one convert_into_range call.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define PST_PLAN_MAX_ENTRIES 30
// PlanEntry::xf_kind of a mapping whose transformation is a device expression (expr.cpp): internal -- pst_transform descriptors stop at
// PST_XF_BITFIELD -- and understood by the plan-specialised kernels only, whose translation unit carries the expression's text
// (PlanEntry::mask = index into the plan's expression table, host side).
#define PST_XF_EXPR 3u

struct PlanEntry {
  uint64_t src_col;  // columnar source: device address of element 0 of the source RANGE; interleaved: unused
  uint64_t dst_col;  // columnar target: device address of element 0 of the target RANGE
  uint32_t src_off;  // interleaved source: attribute offset inside the point record
  uint32_t dst_off;
  uint32_t src_size;  // attribute size in bytes
  uint32_t dst_size;
  uint32_t ncomp;     // components per value: 1 scalar, 3 Vec3, size for opaque (byte-wise)
  uint8_t src_ct;     // pst::CompType of one component
  uint8_t dst_ct;
  uint8_t xf_kind;    // PST_XF_*
  uint8_t xf_on_source;
  double scale[3];  // NB: keep 8-byte alignment: 4 x uint8 + ncomp precede
  double offset[3];
  uint64_t mask;
  uint32_t shift;
  uint16_t convert;   // 1: datatypes differ => Rust `as` per component; 0: same type
  uint16_t bounds;    // 1: fold the written Vec3f64 values into the launch's AABB record (fused calculate_bounds)
};

// Header: passed BY VALUE (kernarg segment, statically indexed => plain s_load).  The entries live in a small device
// buffer and are read through the constant address space (wave-uniform s_load, dynamic index).
struct ConvertHeader {
  uint64_t src_aos;  // interleaved source: device address of point s0
  uint64_t dst_aos;  // interleaved target: device address of point t0
  uint64_t n;        // points in the range
  uint32_t src_stride;
  uint32_t dst_stride;
  uint32_t n_entries;
  uint32_t tile;              // points per LDS tile (tile kernels)
  uint32_t dst_fully_covered; // interleaved target: every byte of the record is written by some mapping
  uint32_t in_place;          // tile kernels, interleaved -> interleaved with src == dst: ONE record tile, transformed in LDS
  uint64_t bounds_partials;   // 0, or device address of gridDim.x records {min xyz, max xyz} (f64) for entries with .bounds
  uint32_t quad;              // tile kernels, columnar -> interleaved: four consecutive points per lane (wave-uniform LDS alignment classes)
  uint32_t reserved;
  uint64_t first_index;       // index of the range's first point in the SOURCE buffer: the `i` of a fused expression (plan-specialised kernels)
  uint64_t expr_params[4];    // device addresses of the f64 arrays a fused expression names p0 .. p3 (0: none; conversions capture nothing)
};
struct ConvertPlan {
  ConvertHeader h;
  PlanEntry e[PST_PLAN_MAX_ENTRIES];
  // tile kernels: masks[0] = entries every wave of a block works on; masks[1 + w] = entries owned by wave w (mod 16)
  uint32_t masks[20];
  // HOST side only (never uploaded: the kernels get h by value and e + masks by copy): the texts of the plan's PST_XF_EXPR entries, as a
  // `const std::vector<std::string>*`
  const void* expr_texts;
};

// SoA Vec3f64 streaming kernel (copy / affine / bounds in one pass)
struct StreamParams {
  const double* src;   // first double of the source range (x of point s0)
  double* dst;         // first double of the target range (may equal src for in-place)
  uint64_t n_doubles;  // 3 * points
  uint64_t vec_first;  // first double covered by the vector body (0 or 1)
  uint64_t n_vec;      // number of W-wide vectors in the body
  double scale[3];
  double offset[3];
  double* partials;    // [gridDim.x][6] = {min xyz, max xyz}
  uint32_t xcd_chunk;  // 0 = tile = blockIdx; else tiles per XCD: workgroups are dealt round-robin to the 8 XCDs, tile = (b % 8) * xcd_chunk + b / 8
  uint32_t xcd_block;  // with xcd_chunk: 0 = every XCD owns ONE contiguous eighth of the stream; B > 0 = runs of B tiles dealt round-robin to the XCDs
};

// generic strided min/max
struct ReduceParams {
  const uint8_t* base;  // address of component 0 of element 0
  uint64_t stride;      // bytes between consecutive elements
  uint64_t n;           // elements
  void* partials;       // [gridDim.x][2*NCOMP] of ACC
};

// Compile-time part of one mapping for the plan-specialised kernels (jit_quad.hpp; written out as source text by jit.cpp).
namespace pstq {
struct QEntry {
  uint32_t src_off, dst_off;    // interleaved side: offset of the attribute in its record; columnar side: 0
  uint32_t src_size, dst_size;  // attribute sizes in bytes
  uint32_t ncomp;               // components per value (1, 3; `size` for opaque byte strings)
  uint32_t src_ct, dst_ct;      // component types (pstd::CT_*)
  uint32_t convert;             // datatypes differ: Rust `as` per component
  uint32_t xf_kind, xf_pre;     // PST_XF_* and "applied to the source value" (buffer_conversion.rs:446-456)
  uint32_t bounds;              // fold the written Vec3f64 values into the launch's AABB record
  uint32_t src_img;             // columnar source: first dword of this attribute's quad image among the lane's source words
  uint32_t src_load;            // columnar source: this entry loads the image (0: an earlier entry reads the same column)
  uint32_t src_wide, dst_wide;  // columnar side: the column's tile is staged in LDS (values of >= 8 bytes, 16-byte aligned column)
  uint32_t src_stage, dst_stage;  // ... at LDS byte T * stage (T = points per tile)
};
}  // namespace pstq
