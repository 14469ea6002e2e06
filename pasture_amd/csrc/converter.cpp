// BufferLayoutConverter on the device: mapping construction (host), plan flattening, kernel selection.
// Reference: pasture-core/src/layout/conversion/buffer_conversion.rs:98-663 and attribute_conversion.rs:184-271.
#include <atomic>
#include <cstdlib>
#include <optional>

#include "jit.hpp"
#include "las_layouts.hpp"
#include "runtime.hpp"

namespace pst {

// The `as` table of attribute_conversion.rs:188-264: every ordered pair of distinct scalar primitives, and every
// ordered pair of distinct Vec3 datatypes (Vec3 exists for u8,u16,i32,f32,f64 only).  Everything else — Vec4u8,
// ByteArray, Custom, scalar<->Vec3 — panics with "Invalid conversion X -> Y" (:267-269).
static bool convertible(const DataType& from, const DataType& to) {
  if (from.is_scalar() && to.is_scalar()) return from.kind != to.kind;
  if (from.is_vec3() && to.is_vec3()) return from.kind != to.kind;
  return false;
}
static void require_convertible(const DataType& from, const DataType& to) {
  if (!convertible(from, to)) throw Error(PST_ERR_INVALID_CONVERSION, "Invalid conversion " + from.display() + " -> " + to.display());
}

struct XfDesc {
  uint32_t kind = PST_XF_NONE;
  DataType datatype;
  double scale[3] = {1, 1, 1};
  double offset[3] = {0, 0, 0};
  uint32_t shift = 0;
  uint64_t mask = ~0ull;
};

// AttributeMapping, buffer_conversion.rs:41-55
struct Mapping {
  Member source, target;
  bool has_converter = false;
  std::optional<XfDesc> xf;
  bool apply_to_source = false;
  std::string expr;  // a device expression instead of a descriptor (expr.cpp): this mapping is its own strided launch
};
// expr.cpp
void launch_expression_mapping(const DataType& src_dt, const DataType& dst_dt, bool apply_to_source, const std::string& expr, uint64_t src, uint64_t sstride, uint64_t dst,
                               uint64_t dstride, uint64_t n, uint64_t first_index, hipStream_t stream);
void validate_expression_mapping(const DataType& src_dt, const DataType& dst_dt, bool apply_to_source, const char* expr);

static Mapping make_default_mapping(const Member& from, const Member& to) {  // :368-396
  Mapping m;
  m.source = from;
  m.target = to;
  if (from.def.datatype != to.def.datatype) {
    require_convertible(from.def.datatype, to.def.datatype);
    m.has_converter = true;
  }
  return m;
}

// Which closures the kernels can evaluate (see pst_transform in include/pasture_amd.h)
static void validate_transform(const XfDesc& x) {
  const uint32_t k = x.datatype.kind;
  if (x.kind == PST_XF_AFFINE) {
    if (k == PST_F64 || k == PST_F32 || k == PST_VEC3F64 || k == PST_VEC3F32) return;
  } else if (x.kind == PST_XF_BITFIELD) {
    if ((k == PST_U8 || k == PST_U16 || k == PST_U32 || k == PST_U64) && x.shift < 64) return;
  }
  throw Error(PST_ERR_UNSUPPORTED_TRANSFORM,
              "Unsupported transformation descriptor for datatype " + x.datatype.display() +
                  ": only AFFINE on F32/F64/Vec3f32/Vec3f64 and BITFIELD on U8/U16/U32/U64 run on the device");
}

}  // namespace pst

// RawPointConverter, attribute_conversion.rs:62-109: one converter per attribute present in both layouts (matched by name, in the order
// of from_layout) whose datatypes DIFFER; same-datatype attributes get none and are therefore skipped, not copied (:73-90)
struct pst_point_converter {
  pst::Layout from, to;
  std::vector<PlanEntry> entries;
};

struct pst_converter {
  pst::Layout from, to;
  std::vector<pst::Mapping> mappings;
  // plan recognition cache for the specialised LAS record decoder (las_decode.hip): -2 = not examined yet, -1 = generic plan,
  // 0..10 = "raw LAS records of this format -> its typed default layout with the mappings of get_default_las_converter"
  // (atomics: two threads may share one converter, each on its own stream; both compute the same value, either store wins)
  mutable std::atomic<int> las_decode_format{-2};
  mutable std::atomic<int> identity_records{-2};
  // Which kernel family a conversion of LAS-shaped interleaved records takes when TWO are at hand -- the format-specialised LAS kernels
  // (las_transpose.hip / las_decode.hip) or the plan-specialised quad kernel (convert_static.hip / hipRTC) -- is MEASURED, once per converter and
  // target storage, on the first call of at least 2^22 points (convert_range: family_autotune): the two trade places from box to box by +-5 %
  // (round-4 review, item 6).  [dst columnar][with bounds]: -1 not measured, 0 = LAS family, 1 = plan-specialised, 2 = nothing to choose.
  mutable std::atomic<int> family_choice[3][2] = {{{-1}, {-1}}, {{-1}, {-1}}, {{-1}, {-1}}};
  mutable std::atomic<float> family_ms[3][2][2] = {};  // [storage pairing, family_slot()][with bounds][family]: what the measurement saw (pst_converter_family_choice)
  mutable std::atomic<int> las_typed_format{-2};  // -2 not examined, -1 no, 0..10: identity plan over LasPointFormatN::layout() (las_transpose.hip)  // -2 not examined, 1 = every byte of every record is copied to the same offset (same packed layout)
};

namespace pst {

static PlanEntry entry_from_mapping(const Mapping& m) {
  PlanEntry e = identity_entry(m.source, m.target);
  e.dst_ct = (uint8_t)m.target.def.datatype.comp_type();
  e.convert = m.has_converter ? 1u : 0u;
  if (m.xf) {
    e.xf_kind = (uint8_t)m.xf->kind;
    e.xf_on_source = m.apply_to_source ? 1 : 0;
    for (int c = 0; c < 3; ++c) { e.scale[c] = m.xf->scale[c]; e.offset[c] = m.xf->offset[c]; }
    e.shift = m.xf->shift;
    e.mask = m.xf->mask;
  }
  return e;
}

// LDS tile: ~36 KiB of records per block (4 resident blocks per CU; measured best on MI355X: larger tiles amortise the
// per-entry interpretation cost, smaller ones raise occupancy), multiple of 64 points.  PST_TILE_POINTS overrides (tuning).
static uint32_t pick_tile(bool src_aos, uint32_t src_stride, bool dst_aos, uint32_t dst_stride) {
  const uint64_t per_point = (src_aos ? src_stride : 0) + (dst_aos ? dst_stride : 0);
  if (per_point == 0) return 0;
  static const long forced = [] { const char* v = std::getenv("PST_TILE_POINTS"); return v && *v ? std::strtol(v, nullptr, 10) : 0L; }();
  static const long budget_env = [] { const char* v = std::getenv("PST_TILE_LDS_BYTES"); return v && *v ? std::strtol(v, nullptr, 10) : 0L; }();
  const uint64_t hard_cap = 160 * 1024 - 256;
  uint64_t t;
  if (forced > 0) {
    t = (uint64_t)forced;
  } else {
    const uint64_t budget = budget_env > 0 ? (uint64_t)budget_env : 36 * 1024;
    t = budget / per_point;
    t = (t / 64) * 64;
    if (t < 64 && 64 * per_point <= hard_cap) t = 64;  // big records: fall back to the smallest tile that still fits
    if (t > 4096) t = 4096;
  }
  if (t * per_point > hard_cap) t = 0;
  return (uint32_t)t;  // 0 => records too large for LDS staging
}

// One launch's plan: header, entries, covered flag, wave scheduling of the interpreted tile kernels.
static ConvertPlan build_plan(bool src_aos, uint64_t src_base, uint32_t src_stride, bool dst_aos, uint64_t dst_base, uint32_t dst_stride, uint64_t n,
                              const PlanEntry* entries, size_t cnt, uint32_t tile, bool in_place, bool with_bounds, bool* wants_bounds) {
  ConvertPlan plan{};
  plan.h.src_aos = src_base;
  plan.h.dst_aos = dst_base;
  plan.h.n = n;
  plan.h.src_stride = src_stride;
  plan.h.dst_stride = dst_stride;
  plan.h.n_entries = (uint32_t)cnt;
  plan.h.tile = tile;
  plan.h.in_place = in_place ? 1u : 0u;
  // Four points per lane put the lanes of a wave `stride` DWORDS apart in the record tile: strides that are multiples of 32 bytes
  // pile them onto the same LDS banks (measured 32 B: 5.96 -> 4.55 TB/s), every other stride gains (41 B: 4.20 -> 5.87).
  // PST_TILE_QUAD=0 / 1 forces the mapping (tuning).
  static const int quad_env = [] { const char* v = std::getenv("PST_TILE_QUAD"); return v && *v ? (*v == '0' ? 0 : 1) : -1; }();
  plan.h.quad = quad_env >= 0 ? (uint32_t)quad_env : ((dst_stride % 32u != 0 && !(src_aos && src_stride % 32u == 0)) ? 1u : 0u);
  std::vector<uint8_t> covered(dst_aos ? dst_stride : 0, 0);
  *wants_bounds = false;
  for (size_t i = 0; i < cnt; ++i) {
    plan.e[i] = entries[i];
    if (!with_bounds) plan.e[i].bounds = 0;
    *wants_bounds = *wants_bounds || plan.e[i].bounds;
    if (dst_aos)
      for (uint32_t b = 0; b < plan.e[i].dst_size; ++b) covered[plan.e[i].dst_off + b] = 1;
  }
  plan.h.dst_fully_covered = dst_aos && std::all_of(covered.begin(), covered.end(), [](uint8_t c) { return c != 0; });
  // wave scheduling of the tile kernels (convert.hip): narrow attributes are owned by single waves, balanced by bytes
  {
    const long nwaves = 4;  // convert_tile_kernel runs 256-thread blocks
    static const long own_max = [] { const char* v = std::getenv("PST_TILE_OWN_MAX_BYTES"); return v && *v ? std::strtol(v, nullptr, 10) : 4L; }();
    uint64_t load[16] = {0};
    for (size_t i = 0; i < cnt; ++i) {
      const uint32_t col_bytes = (src_aos && !dst_aos) ? plan.e[i].dst_size : (!src_aos && dst_aos) ? plan.e[i].src_size
                                                                                                  : std::max(plan.e[i].src_size, plan.e[i].dst_size);
      if ((long)col_bytes > own_max || plan.e[i].bounds || nwaves == 1) { plan.masks[0] |= 1u << i; continue; }
      long best = 0;
      for (long w = 1; w < nwaves; ++w) if (load[w] < load[best]) best = w;
      load[best] += col_bytes + 2;  // + fixed interpretation cost
      plan.masks[1 + best] |= 1u << i;
    }
  }
  return plan;
}

// Whether a plan-specialised kernel (convert.hip: in-tree or run-time compiled) is at hand for ONE launch over these entries; a missing one is
// queued for the compiler thread like a launch would.
bool specialised_kernel_ready(bool src_aos, uint64_t src_base, uint32_t src_stride, bool dst_aos, uint64_t dst_base, uint32_t dst_stride, uint64_t n,
                              const std::vector<PlanEntry>& entries, bool with_bounds) {
  if (entries.empty() || entries.size() > PST_PLAN_MAX_ENTRIES || !(src_aos || dst_aos)) return false;
  const uint32_t tile = pick_tile(src_aos, src_stride, dst_aos, dst_stride);
  if (tile < 1) return false;
  bool wants_bounds = false;
  ConvertPlan plan = build_plan(src_aos, src_base, src_stride, dst_aos, dst_base, dst_stride, n, entries.data(), entries.size(), tile, false, with_bounds, &wants_bounds);
  if (wants_bounds) plan.h.bounds_partials = 0x30000000ull;  // (a placeholder: only whether there is one enters the kernel's signature)
  return pstk::convert_specialised_ready(plan, src_aos, dst_aos);
}

// whole_records_in_place: source and target are the SAME interleaved range and the entries copy every byte of a record (identity entries
// around the transformed one): the plan runs as an ordinary records -> records conversion -- every tile is read into LDS before it is
// written and tiles are disjoint -- which lets the plan-specialised kernels take it (transform_attribute on a packed VectorBuffer)
void execute_entries(bool src_aos, uint64_t src_base, uint32_t src_stride, bool dst_aos, uint64_t dst_base, uint32_t dst_stride,
                     uint64_t n, const std::vector<PlanEntry>& entries, bool allow_lds, hipStream_t stream, double* bounds_out6, bool whole_records_in_place) {
  if (n == 0 || entries.empty()) return;
  ensure_device();
  // interleaved in place (transform_attribute on a VectorBuffer): one record tile, transformed in LDS
  const bool in_place = src_aos && dst_aos && src_base == dst_base && src_stride == dst_stride && !whole_records_in_place;
  const uint32_t tile = pick_tile(src_aos && !in_place, src_stride, dst_aos, dst_stride);
  const bool use_lds = allow_lds && (src_aos || dst_aos) && tile >= 1;
  for (size_t begin = 0; begin < entries.size(); begin += PST_PLAN_MAX_ENTRIES) {
    const size_t cnt = std::min<size_t>(PST_PLAN_MAX_ENTRIES, entries.size() - begin);
    bool wants_bounds = false;
    ConvertPlan plan = build_plan(src_aos, src_base, src_stride, dst_aos, dst_base, dst_stride, n, entries.data() + begin, cnt, tile, in_place,
                                  bounds_out6 != nullptr, &wants_bounds);
    unsigned records = 0;
    double* partials = nullptr;
    if (wants_bounds) {
      partials = (double*)workspace().partials(pstk::bounds_partials_bytes(pstk::convert_max_records(plan, src_aos, dst_aos, use_lds)));
      plan.h.bounds_partials = (uint64_t)(uintptr_t)partials;
    }
    if (!pstk::launch_convert(plan, src_aos, dst_aos, use_lds, stream, &records))
      throw hip_failure("conversion kernel launch failed: ");
    if (wants_bounds) pstk::launch_finalize_bounds(partials, records, bounds_out6, stream);
  }
}

// transform_attribute with the closure as a device expression, IN PLACE on the records of an interleaved buffer without padding (round 6): the
// transformation as a records -> records plan whose other attributes are identity copies and whose transformed attribute is a PST_XF_EXPR entry
// -- the plan-specialised kernel reads every tile into registers before it writes it, tiles are disjoint --, the expression's index `i` is the
// point's index, p[0 .. 3] the arrays it names.  Returns the points covered (full tiles; 0: no such kernel for this layout / PST_JIT=0 -- the caller's
// strided kernel takes everything); throws when the kernel with the expression in it does not compile.
uint64_t transform_records_with_expression(const pst_buffer& b, int slot, const std::string& expr, const double* const p[4], hipStream_t stream) {
  static const bool fuse_env = [] { const char* v = std::getenv("PST_EXPR_FUSE"); return !(v && *v == '0'); }();
  uint64_t attr_bytes = 0;
  for (const Member& mm : b.layout.members) attr_bytes += mm.size;
  if (!fuse_env || b.columnar || b.len == 0 || attr_bytes != b.layout.size || b.layout.members.size() > PST_PLAN_MAX_ENTRIES || pstjit::mode() == pstjit::Mode::Off) return 0;
  std::vector<PlanEntry> all;
  for (size_t a = 0; a < b.layout.members.size(); ++a) {
    PlanEntry e = identity_entry(b.layout.members[a], b.layout.members[a]);
    if ((int)a == slot) { e.xf_kind = (uint8_t)PST_XF_EXPR; e.xf_on_source = 0; e.mask = 0; }
    all.push_back(e);
  }
  const uint32_t stride = (uint32_t)b.layout.size;
  const uint32_t tile = pick_tile(true, stride, true, stride);
  if (tile < 1) return 0;
  const std::vector<std::string> texts{expr};
  bool wants_bounds = false;
  const uint64_t base = aos_addr(b, 0);
  ConvertPlan plan = build_plan(true, base, stride, true, base, stride, b.len, all.data(), all.size(), tile, false, false, &wants_bounds);
  plan.expr_texts = &texts;
  plan.h.first_index = 0;
  for (int q = 0; q < 4; ++q) plan.h.expr_params[q] = p ? (uint64_t)(uintptr_t)p[q] : 0;
  uint64_t done = 0;
  std::string err;
  pstk::reset_plan_kinds();
  if (!pstk::launch_convert_fused_expressions(plan, true, true, stream, &done, &err))
    throw hip_failure("transformation kernel launch failed: ");
  if (!err.empty()) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "transformation expression: the kernel with the expression in it does not compile:\n" + err);
  return done;
}

// Identity between two buffers of the same layout in which the mappings cover every byte of the record (no padding, no
// unmapped attribute): interleaved -> interleaved is then a plain byte copy of the records (what the reference's per-attribute
// loops add up to, buffer_conversion.rs:606-662).
// Round 4: the plan-specialised kernels (jit.cpp / convert_static.hip) have overtaken two of the format-specialised LAS kernels -- typed
// LAS records -> columns (same box: 0.81 against 0.78 of peak) and raw LAS records -> typed records (0.74 against 0.65).  Those two plans take
// the generic path WHEN a specialised kernel for them is at hand (in-tree, compiled, or PST_JIT=sync), and the LAS kernel otherwise
// (PST_JIT=0, hipRTC missing, the first calls while the compiler thread works).  The mappings of such a plan, as the generic path lists them:
// Slot of a storage pairing in pst_converter::family_choice: 0 = records -> records, 1 = records -> columns, 2 = columns -> records (columns -> columns
// has one family).  Keyed by the SOURCE storage too since round 6: an identity LAS converter used records -> records (one family: choice 2) no longer
// shares -- and blocks -- the slot of its columns -> records direction.
static inline int family_slot(bool src_columnar, bool dst_columnar) { return dst_columnar ? 1 : (src_columnar ? 2 : 0); }
static std::vector<PlanEntry> interleaved_source_entries(const pst_converter& c, const pst_buffer* dst, size_t t0, bool dst_columnar, int pos_slot, bool with_bounds) {
  std::vector<PlanEntry> out;
  bool bounds_done = false;
  for (const Mapping& m : c.mappings) {
    PlanEntry e = entry_from_mapping(m);
    const int tslot = c.to.index_of(m.target.def);
    if (dst_columnar) e.dst_col = dst ? col_addr(*dst, (size_t)tslot, t0) : 0x100000ull * (uint64_t)(tslot + 1);
    if (with_bounds && tslot == pos_slot && m.target.def.datatype.kind == PST_VEC3F64 && !bounds_done) { e.bounds = 1; bounds_done = true; }
    out.push_back(e);
  }
  return out;
}
// force_family: -1 = by the converter's measured choice (default: plan-specialised), 0 = the LAS family, 1 = plan-specialised if at hand
static bool las_plan_prefers_generic(const pst_converter& c, const pst_buffer& src, size_t s0, const pst_buffer& dst, size_t t0, uint64_t n, int pos_slot,
                                     bool with_bounds, int force_family) {
  static const bool on = [] { const char* v = std::getenv("PST_LAS_PREFER_SPECIALISED"); return !(v && *v == '0'); }();  // the A/B switch
  if (!on || force_family == 0) return false;
  const int choice = c.family_choice[family_slot(src.columnar, dst.columnar)][with_bounds ? 1 : 0];
  if (force_family < 0 && choice == 0) return false;
  if (src.columnar) {
    // columns -> typed LAS records: the LAS transposer stays the default; the plan-specialised kernel runs when the measurement chose it (or forces it)
    if (dst.columnar || (force_family < 0 && choice != 1)) return false;
    std::vector<PlanEntry> entries;
    bool bounds_done = false;
    for (const Mapping& m : c.mappings) {
      if (!m.expr.empty()) return false;
      PlanEntry e = entry_from_mapping(m);
      const int sslot = c.from.index_of(m.source.def), tslot = c.to.index_of(m.target.def);
      e.src_col = col_addr(src, (size_t)sslot, s0);
      if (with_bounds && tslot == pos_slot && m.target.def.datatype.kind == PST_VEC3F64 && !bounds_done) { e.bounds = 1; bounds_done = true; }
      entries.push_back(e);
    }
    if (entries.empty() || entries.size() > PST_PLAN_MAX_ENTRIES) return false;
    return specialised_kernel_ready(false, 0, (uint32_t)c.from.size, true, aos_addr(dst, t0), (uint32_t)c.to.size, n, entries, with_bounds);
  }
  const std::vector<PlanEntry> entries = interleaved_source_entries(c, &dst, t0, dst.columnar, pos_slot, with_bounds);
  if (entries.empty() || entries.size() > PST_PLAN_MAX_ENTRIES) return false;
  return specialised_kernel_ready(true, aos_addr(src, s0), (uint32_t)c.from.size, !dst.columnar, dst.columnar ? 0 : aos_addr(dst, t0), (uint32_t)c.to.size, n, entries,
                                  with_bounds);
}
static bool match_identity_records(const pst_converter& c) {
  if (!(c.from == c.to) || c.mappings.size() != c.to.members.size()) return false;
  uint64_t covered = 0;
  std::vector<uint8_t> seen(c.to.members.size(), 0);
  for (const Mapping& m : c.mappings) {
    const int tslot = c.to.index_of(m.target.def), sslot = c.from.index_of(m.source.def);
    if (tslot < 0 || tslot != sslot || seen[(size_t)tslot] || m.xf || m.has_converter || !m.expr.empty()) return false;
    seen[(size_t)tslot] = 1;
    covered += m.target.size;
  }
  return covered == c.to.size;
}

// Is this converter exactly the plan get_default_las_converter (raw_readers.rs:31-167) builds for (raw records of format N ->
// LasPointFormatN::layout())?  Then las_decode.hip runs it with the format as a compile-time parameter.
static int match_las_decode_plan(const pst_converter& c) {
  using namespace laslayout;
  int format = -1;
  for (uint32_t f = 0; f <= 10; ++f)
    if (c.from == raw_layout(f) && c.to == typed_layout(f)) { format = (int)f; break; }
  if (format < 0 || c.mappings.size() != c.to.members.size()) return -1;
  const bool ext = format >= 6;
  const char* flags = ext ? "LASExtendedFlags" : "LASBasicFlags";
  struct Bits { const char* target; uint32_t shift; uint64_t mask; };
  static const Bits basic[] = {{"ReturnNumber", 0, 7}, {"NumberOfReturns", 3, 7}, {"ScanDirectionFlag", 6, 1}, {"EdgeOfFlightLine", 7, 1}};
  static const Bits extended[] = {{"ReturnNumber", 0, 15},    {"NumberOfReturns", 4, 15},   {"ClassificationFlags", 8, 15},
                                  {"ScannerChannel", 12, 3},  {"ScanDirectionFlag", 14, 1}, {"EdgeOfFlightLine", 15, 1}};
  std::vector<uint8_t> seen(c.to.members.size(), 0);
  for (const Mapping& m : c.mappings) {
    const int tslot = c.to.index_of(m.target.def);
    if (tslot < 0 || seen[(size_t)tslot] || !m.expr.empty()) return -1;
    seen[(size_t)tslot] = 1;
    const std::string& tn = m.target.def.name;
    if (tn == "Position3D") {
      if (m.source.def.name != "LASLocalPosition" || !m.xf || m.xf->kind != PST_XF_AFFINE || m.xf->datatype.kind != PST_VEC3F64 || m.apply_to_source)
        return -1;
      continue;
    }
    const Bits* b = nullptr;
    const Bits* table = ext ? extended : basic;
    for (size_t i = 0; i < (ext ? 6u : 4u); ++i)
      if (tn == table[i].target) b = &table[i];
    if (b) {  // bit fields of the flags attribute, applied to the SOURCE value (raw_readers.rs:61-164)
      if (m.source.def.name != flags || !m.xf || m.xf->kind != PST_XF_BITFIELD || !m.apply_to_source || m.xf->shift != b->shift ||
          (m.xf->mask & 0xFFFFull) != b->mask)
        return -1;
      continue;
    }
    if (m.source.def.name != tn || m.xf || m.has_converter || !m.expr.empty()) return -1;  // plain copy of the same-named attribute
  }
  return format;
}

// force_family: -1 = the converter's choice, 0 / 1 = that family (the measurement's own passes); kMeasureFamilies = -1 plus: measure the two families first if
// this converter has not yet (synchronous entry points only)
constexpr int kMeasureFamilies = -2;
static void convert_range(const pst_converter& c, pst_buffer& src, size_t s0, size_t s1, pst_buffer& dst, size_t t0, size_t t1, double* bounds_out6, hipStream_t stream,
                          int force_family = -1);

// Do the byte ranges the conversion reads and writes overlap?  (Two slice handles of one parent are different pst_buffer objects over the same
// memory: the measurement below repeats the conversion, which is only idempotent when the source is not written.)
static bool storage_overlaps(const pst_buffer& src, size_t s0, size_t s1, const pst_buffer& dst, size_t t0, size_t t1) {
  struct Span { uint64_t lo, hi; };
  auto spans = [](const pst_buffer& b, size_t p0, size_t p1) {
    std::vector<Span> v;
    if (!b.columnar) v.push_back({aos_addr(b, p0), aos_addr(b, p1)});
    else
      for (size_t a = 0; a < b.layout.members.size(); ++a) v.push_back({col_addr(b, a, p0), col_addr(b, a, p1)});
    return v;
  };
  for (const Span& a : spans(src, s0, s1))
    for (const Span& b : spans(dst, t0, t1))
      if (a.lo < b.hi && b.lo < a.hi) return true;
  return false;
}

// The measurement behind pst_converter::family_choice.  Runs on the caller's buffers and range (both families write the same bytes, the source is
// only read): per family one untimed pass and THREE timed ones, each between its own pair of HIP events on the caller's stream (the median decides),
// then ONE host wait.  Since round 6 it runs from the SYNCHRONOUS conversion entry points and from pst_converter_measure_families only: a
// stream-ordered (`_async`) call never blocks the host for it, nor repeats the caller's conversion.  Skipped (and left for a later call) while the
// stream is being captured into a graph, below 2^22 points, when source and target memory overlap, and when the plan-specialised kernel is not
// compiled yet.  PST_FAMILY_AUTOTUNE=0 switches it off (the default order of preference then stands: plan-specialised first).
static void family_autotune(const pst_converter& c, pst_buffer& src, size_t s0, size_t s1, pst_buffer& dst, size_t t0, size_t t1, double* bounds_out6, hipStream_t stream) {
  static const bool on = [] { const char* v = std::getenv("PST_FAMILY_AUTOTUNE"); return !(v && *v == '0'); }();
  const uint64_t n = s1 - s0;
  const int slot = family_slot(src.columnar, dst.columnar), bslot = bounds_out6 ? 1 : 0;
  std::atomic<int>& choice = c.family_choice[slot][bslot];
  if (!on || choice != -1 || n < ((uint64_t)1 << 22) || (src.columnar && dst.columnar) || storage_overlaps(src, s0, s1, dst, t0, t1)) return;
  // is this one of the two LAS-shaped plans at all?
  if (c.las_typed_format == -2) {
    int f = -1;
    if (match_identity_records(c))
      for (uint32_t k = 0; k <= 10; ++k)
        if (c.to == laslayout::typed_layout(k)) { f = (int)k; break; }
    c.las_typed_format = f;
  }
  if (c.las_decode_format == -2) c.las_decode_format = match_las_decode_plan(c);
  const bool las_shaped = (dst.columnar || src.columnar) ? c.las_typed_format >= 0 : c.las_decode_format >= 0;  // (columns <-> typed records; raw -> typed records)
  if (!las_shaped) { choice = 2; return; }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
  int pos_slot = -1;
  if (bounds_out6) { const Member* pm = c.to.find_by_name("Position3D"); if (!pm) return; pos_slot = (int)(pm - c.to.members.data()); }
  if (!las_plan_prefers_generic(c, src, s0, dst, t0, n, pos_slot, bounds_out6 != nullptr, 1)) return;  // (not compiled yet: queued; measured on a later call)
  constexpr int kTimed = 3;
  hipEvent_t ev[2][kTimed + 1] = {};
  bool ok = true;
  for (auto& row : ev) for (hipEvent_t& e : row) ok = ok && hipEventCreate(&e) == hipSuccess;
  float med[2] = {0.f, 0.f};
  if (ok) {
    for (int fam = 0; fam < 2; ++fam) {
      convert_range(c, src, s0, s1, dst, t0, t1, bounds_out6, stream, fam);
      ok = ok && hipEventRecord(ev[fam][0], stream) == hipSuccess;
      for (int r = 0; r < kTimed; ++r) {
        convert_range(c, src, s0, s1, dst, t0, t1, bounds_out6, stream, fam);
        ok = ok && hipEventRecord(ev[fam][r + 1], stream) == hipSuccess;
      }
    }
    ok = ok && hipEventSynchronize(ev[1][kTimed]) == hipSuccess;
    for (int fam = 0; fam < 2 && ok; ++fam) {
      float t[kTimed];
      for (int r = 0; r < kTimed && ok; ++r) ok = hipEventElapsedTime(&t[r], ev[fam][r], ev[fam][r + 1]) == hipSuccess;
      std::sort(t, t + kTimed);
      med[fam] = t[kTimed / 2];
    }
  }
  for (auto& row : ev) for (hipEvent_t& e : row) if (e) (void)hipEventDestroy(e);
  if (!ok) { (void)hipGetLastError(); return; }
  for (int fam = 0; fam < 2; ++fam) c.family_ms[slot][bslot][fam] = med[fam];
  choice = med[1] <= med[0] ? 1 : 0;
}

// ---- convert_into_range, buffer_conversion.rs:292-359 -----------------------------------------------------
// bounds_out6: when non-null, {min xyz, max xyz} of the TARGET's POSITION_3D over the target range is written there
// (device-accessible memory), fused into the conversion pass when possible.
static void convert_range(const pst_converter& c, pst_buffer& src, size_t s0, size_t s1, pst_buffer& dst, size_t t0, size_t t1,
                          double* bounds_out6, hipStream_t stream, int force_family) {
  if (src.layout != c.from) throw Error(PST_ERR_LAYOUT_MISMATCH, "assertion `left == right` failed: source_buffer.point_layout() != from_layout");
  if (dst.layout != c.to) throw Error(PST_ERR_LAYOUT_MISMATCH, "assertion `left == right` failed: target_buffer.point_layout() != to_layout");
  if (s1 < s0 || t1 < t0 || (s1 - s0) != (t1 - t0)) throw Error(PST_ERR_RANGE, "assertion failed: source_range.len() == target_range.len()");
  if (s1 > src.len) throw Error(PST_ERR_RANGE, "assertion failed: source_range.end <= source_buffer.len()");
  if (t1 > dst.len) throw Error(PST_ERR_RANGE, "assertion failed: target_range.end <= target_buffer.len()");
  const uint64_t n = s1 - s0;
  ensure_device();
  if (force_family == kMeasureFamilies) { family_autotune(c, src, s0, s1, dst, t0, t1, bounds_out6, stream); force_family = -1; }
  pstk::reset_plan_kinds();

  // position attribute of the target for the fused / trailing bounds
  int pos_slot = -1;
  if (bounds_out6) {
    const Member* pm = c.to.find_by_name("Position3D");
    if (!pm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "target PointLayout has no Position3D attribute");
    pos_slot = (int)(pm - c.to.members.data());
  }
  bool bounds_done = false;

  if (n > 0 && !src.columnar && !dst.columnar && !bounds_out6) {
    if (c.identity_records == -2) c.identity_records = match_identity_records(c) ? 1 : 0;
    if (c.identity_records == 1) {  // one wide-vector byte copy of the record range (columns.hip)
      PlanEntry e{};
      e.src_col = aos_addr(src, s0);
      e.dst_col = aos_addr(dst, t0);
      e.src_size = e.dst_size = (uint32_t)c.to.size;
      e.ncomp = 1;
      if (!pstk::launch_column(e, n, nullptr, stream))
        throw hip_failure("record copy launch failed: ");
      pstk::note_plan_kind(PST_PLAN_COPY);
      return;
    }
  }
  static const bool las_fast = [] { const char* v = std::getenv("PST_LAS_DECODE"); return !(v && *v == '0'); }();
  if (las_fast && n > 0 && src.columnar != dst.columnar) {
    if (c.las_typed_format == -2) {
      c.las_typed_format = -1;
      if (match_identity_records(c))
        for (uint32_t f = 0; f <= 10; ++f)
          if (c.to == laslayout::typed_layout(f)) { c.las_typed_format = (int)f; break; }
    }
    if (c.las_typed_format >= 0 && !las_plan_prefers_generic(c, src, s0, dst, t0, n, pos_slot, bounds_out6 != nullptr, force_family)) {
      // typed LAS points, columns <-> packed records: format-specialised transposition
      const pst_buffer& soa = src.columnar ? src : dst;
      const size_t p0 = src.columnar ? s0 : t0;
      std::vector<uint64_t> cols(c.to.members.size());
      for (size_t a = 0; a < cols.size(); ++a) cols[a] = col_addr(soa, a, p0);
      const uint64_t aos = src.columnar ? aos_addr(dst, t0) : aos_addr(src, s0);
      const unsigned grid = pstk::las_transpose_grid(n);
      double* partials = bounds_out6 ? (double*)workspace().partials(pstk::bounds_partials_bytes(grid)) : nullptr;
      if (!pstk::launch_las_transpose(c.las_typed_format, src.columnar, aos, cols.data(), (int)cols.size(), n, partials, stream))
        throw hip_failure("LAS transposition launch failed: ");
      if (bounds_out6) pstk::launch_finalize_bounds(partials, grid, bounds_out6, stream);
      pstk::note_plan_kind(PST_PLAN_LAS);
      return;
    }
  }
  std::vector<PlanEntry> generic;
  if (las_fast && n > 0 && !src.columnar && c.las_decode_format == -2) c.las_decode_format = match_las_decode_plan(c);
  if (las_fast && n > 0 && !src.columnar && c.las_decode_format >= 0 &&
      !(!dst.columnar && las_plan_prefers_generic(c, src, s0, dst, t0, n, pos_slot, bounds_out6 != nullptr, force_family))) {
    // the production plan of the LAS readers: format-specialised kernel (las_decode.hip)
    const Mapping* pos = nullptr;
    for (const Mapping& m : c.mappings)
      if (m.target.def.name == "Position3D") pos = &m;
    if (!dst.columnar) {  // VectorBuffer of LasPointFormatN: one lane per point, records assembled in LDS
      const unsigned grid = pstk::las_decode_aos_grid(c.las_decode_format, n);
      double* partials = bounds_out6 ? (double*)workspace().partials(pstk::bounds_partials_bytes(grid)) : nullptr;
      if (!pstk::launch_las_decode_aos(c.las_decode_format, aos_addr(src, s0), aos_addr(dst, t0), n, pos->xf->scale, pos->xf->offset, partials, stream))
        throw hip_failure("LAS decode launch failed: ");
      if (bounds_out6) pstk::launch_finalize_bounds(partials, grid, bounds_out6, stream);
      pstk::note_plan_kind(PST_PLAN_LAS);
      return;
    }
    std::vector<uint64_t> cols(c.to.members.size());
    for (size_t a = 0; a < cols.size(); ++a) cols[a] = col_addr(dst, a, t0);
    double* partials = nullptr;
    if (bounds_out6) partials = (double*)workspace().partials(pstk::bounds_partials_bytes(pstk::las_decode_grid(n)));
    if (!pstk::launch_las_decode(c.las_decode_format, aos_addr(src, s0), n, cols.data(), (int)cols.size(), pos->xf->scale, pos->xf->offset, partials,
                                 stream))
      throw hip_failure("LAS decode launch failed: ");
    if (bounds_out6) pstk::launch_finalize_bounds(partials, pstk::las_decode_grid(n), bounds_out6, stream);
    pstk::note_plan_kind(PST_PLAN_LAS);
    return;
  }
  if (!c.mappings.empty() && n > 0) {  // no mappings => silent no-op (:308-313)
    // User-written transformations (expr.cpp).  The reference applies the closure INSIDE the conversion loop (buffer_conversion.rs:569-590, 471-484).
    // Round 6: where a side is interleaved -- where a pass of its own re-reads and re-writes whole records for one attribute -- the expressions become
    // part of the plan's specialised kernel (PST_XF_EXPR entries; jit.cpp writes their text into the translation unit): ONE kernel, the plan's
    // algorithmic bytes.  Columns -> columns is one launch per mapping anyway (the expression's kernel IS that mapping's pass: source read once,
    // target written once).  The fused AABB is not combined with expressions (bounds_of_range follows).  Fall-back -- no specialised form for this
    // plan (unaligned record bases, records beyond the register images, PST_JIT=0 refused earlier) and the ragged tail: one strided launch per
    // expression mapping, the index the expression sees is the point's index in the SOURCE buffer either way.
    bool any_expr = false;
    for (const Mapping& m : c.mappings) any_expr = any_expr || !m.expr.empty();
    static const bool fuse_env = [] { const char* v = std::getenv("PST_EXPR_FUSE"); return !(v && *v == '0'); }();  // the A/B switch
    const bool fuse_expr = any_expr && fuse_env && !(src.columnar && dst.columnar) && c.mappings.size() <= PST_PLAN_MAX_ENTRIES && pstjit::mode() != pstjit::Mode::Off;
    struct ExprPass { const Mapping* m; uint64_t src, sstride, dst, dstride; };
    std::vector<ExprPass> expr_passes;
    std::vector<std::string> expr_texts;
    std::vector<PlanEntry> fused;  // fuse_expr: every mapping, in order
    auto run_expr_pass = [&](const ExprPass& x, uint64_t first, uint64_t count) {
      launch_expression_mapping(x.m->source.def.datatype, x.m->target.def.datatype, x.m->apply_to_source, x.m->expr, x.src + first * x.sstride, x.sstride,
                                x.dst + first * x.dstride, x.dstride, count, s0 + first, stream);
    };
    for (const Mapping& m : c.mappings) {
      PlanEntry e = entry_from_mapping(m);
      const int sslot = c.from.index_of(m.source.def), tslot = c.to.index_of(m.target.def);
      if (src.columnar) e.src_col = col_addr(src, (size_t)sslot, s0);
      if (dst.columnar) e.dst_col = col_addr(dst, (size_t)tslot, t0);
      if (!m.expr.empty()) {
        const ExprPass x{&m, src.columnar ? e.src_col : aos_addr(src, s0) + m.source.offset, src.columnar ? m.source.size : c.from.size,
                         dst.columnar ? e.dst_col : aos_addr(dst, t0) + m.target.offset, dst.columnar ? m.target.size : c.to.size};
        if (fuse_expr) {
          e.xf_kind = (uint8_t)PST_XF_EXPR;
          e.xf_on_source = m.apply_to_source ? 1 : 0;
          e.mask = expr_texts.size();
          expr_texts.push_back(m.expr);
          expr_passes.push_back(x);
          fused.push_back(e);
        } else {
          run_expr_pass(x, 0, n);
        }
        continue;
      }
      if (fuse_expr) { fused.push_back(e); generic.push_back(e); continue; }  // (generic: the same plan without its expressions, for the fall-back and the tail)
      if (src.columnar && dst.columnar) {
        const bool same_type = !m.has_converter;
        const bool vec3f64 = m.source.def.datatype.kind == PST_VEC3F64;
        const bool want_bounds = bounds_out6 && tslot == pos_slot && vec3f64 && same_type;
        const bool affine = m.xf && m.xf->kind == PST_XF_AFFINE;
        const bool aligned = (e.src_col % 8 == 0) && (e.dst_col % 8 == 0) && (e.src_col % 16 == e.dst_col % 16);
        if (same_type && vec3f64 && aligned && (affine || want_bounds)) {
          // K2 fast path: one coalesced pass: copy (+ affine) (+ AABB)
          Workspace& ws = workspace();
          unsigned mode = 2u | (affine ? 1u : 0u) | (want_bounds ? 4u : 0u);
          pstk::launch_vec3f64_stream((const double*)(uintptr_t)e.src_col, (double*)(uintptr_t)e.dst_col, n, e.scale, e.offset, mode,
                                      (double*)ws.partials(pstk::stream_partials_bytes(n, mode)), bounds_out6, stream);
          if (want_bounds) bounds_done = true;
          pstk::note_plan_kind(PST_PLAN_STREAM);
          continue;
        }
        // every other columnar -> columnar mapping is its own wide-vector launch (columns.hip): plain copies move raw
        // bytes (== set_attribute_range / copy_from_slice, buffer_conversion.rs:465-469), the rest converts per component
        const bool fuse_bounds = bounds_out6 && tslot == pos_slot && m.target.def.datatype.kind == PST_VEC3F64 && !bounds_done;
        double* partials = nullptr;
        unsigned grid = 0;
        if (fuse_bounds) {
          grid = pstk::column_launch_grid(e, n, true);
          partials = (double*)workspace().partials(pstk::bounds_partials_bytes(grid));
        }
        if (!pstk::launch_column(e, n, partials, stream))
          throw hip_failure("column conversion launch failed: ");
        pstk::note_plan_kind(PST_PLAN_COLUMN);
        if (fuse_bounds) {
          pstk::launch_finalize_bounds(partials, grid, bounds_out6, stream);
          bounds_done = true;
        }
        continue;
      }
      if (bounds_out6 && tslot == pos_slot && m.target.def.datatype.kind == PST_VEC3F64 && !bounds_done) {
        e.bounds = 1;  // fused: the kernel folds the Vec3f64 values it writes into the AABB record
        bounds_done = true;
      }
      generic.push_back(e);
    }
    uint64_t fused_done = 0;
    if (fuse_expr) {
      const bool sa = !src.columnar, da = !dst.columnar;
      const uint32_t tile = pick_tile(sa, (uint32_t)c.from.size, da, (uint32_t)c.to.size);
      if (tile >= 1) {
        bool wants_bounds = false;
        ConvertPlan plan = build_plan(sa, sa ? aos_addr(src, s0) : 0, (uint32_t)c.from.size, da, da ? aos_addr(dst, t0) : 0, (uint32_t)c.to.size, n, fused.data(), fused.size(), tile,
                                      false, false, &wants_bounds);
        plan.expr_texts = &expr_texts;
        plan.h.first_index = s0;
        std::string err;
        if (!pstk::launch_convert_fused_expressions(plan, sa, da, stream, &fused_done, &err))
          throw hip_failure("conversion kernel launch failed: ");
        if (!err.empty()) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "transformation expression: the conversion kernel with the expression(s) in it does not compile:\n" + err);
      }
      // what the fused kernel did not cover -- the ragged tail (less than one tile), or everything: the expressions' own strided launches ...
      if (fused_done < n)
        for (const ExprPass& x : expr_passes) run_expr_pass(x, fused_done, n - fused_done);
      // ... and the other mappings of those points through the generic path
      if (fused_done > 0)
        for (PlanEntry& e : generic) {
          if (src.columnar) e.src_col += fused_done * e.src_size;
          if (dst.columnar) e.dst_col += fused_done * e.dst_size;
        }
    }
    if (!generic.empty() && fused_done < n)
      execute_entries(!src.columnar, src.columnar ? 0 : aos_addr(src, s0 + fused_done), (uint32_t)c.from.size, !dst.columnar,
                      dst.columnar ? 0 : aos_addr(dst, t0 + fused_done), (uint32_t)c.to.size, n - fused_done, generic, true, stream, fuse_expr ? nullptr : bounds_out6);
  }
  PST_HIP_CHECK(hipGetLastError());
  if (bounds_out6 && !bounds_done) {
    bounds_of_range(dst, t0, n, bounds_out6, stream);
  }
}

// The plan a conversion between buffers of the given storage kinds would hand to the generic tile kernels, with placeholder addresses
// (only their equality pattern matters to the specialised kernels): what pst_converter_prepare compiles ahead of the first call.
// Returns the PST_PLAN_* family the call would take WITHOUT a specialised kernel, and fills `plan` when that family is the generic one.
static uint32_t plan_for_storage(const pst_converter& c, bool src_columnar, bool dst_columnar, bool with_bounds, ConvertPlan* plan, std::vector<std::string>* expr_texts) {
  static const bool las_fast = [] { const char* v = std::getenv("PST_LAS_DECODE"); return !(v && *v == '0'); }();
  if (c.mappings.empty()) return PST_PLAN_NONE;
  if (!src_columnar && !dst_columnar && !with_bounds && match_identity_records(c)) return PST_PLAN_COPY;
  // (typed LAS records -> columns and raw LAS records -> typed records prefer a plan-specialised kernel when the run-time compiler is on:
  //  las_plan_prefers_generic above; pst_converter_prepare then compiles the generic plan, and the LAS kernel stays the stand-in)
  static const bool prefer_generic = [] { const char* v = std::getenv("PST_LAS_PREFER_SPECIALISED"); return !(v && *v == '0'); }();
  const bool jit_on = prefer_generic && pstjit::mode() != pstjit::Mode::Off;
  if (las_fast && src_columnar != dst_columnar && match_identity_records(c) && !jit_on)  // (with the compiler on: the generic plan is compiled for BOTH directions, family_autotune picks)
    for (uint32_t f = 0; f <= 10; ++f)
      if (c.to == laslayout::typed_layout(f)) return PST_PLAN_LAS;
  if (las_fast && !src_columnar && match_las_decode_plan(c) >= 0 && !(jit_on && !dst_columnar)) return PST_PLAN_LAS;
  if (src_columnar && dst_columnar) return PST_PLAN_COLUMN;
  bool any_expr = false;
  for (const Mapping& m : c.mappings) any_expr = any_expr || !m.expr.empty();
  const Member* pm = (with_bounds && !any_expr) ? c.to.find_by_name("Position3D") : nullptr;  // (a plan with expressions does not fuse the AABB: convert_range)
  std::vector<PlanEntry> generic;
  bool bounds_done = false;
  for (const Mapping& m : c.mappings) {
    PlanEntry e = entry_from_mapping(m);
    if (!m.expr.empty()) {  // fused into the plan's kernel, as convert_range does
      e.xf_kind = (uint8_t)PST_XF_EXPR;
      e.xf_on_source = m.apply_to_source ? 1 : 0;
      e.mask = expr_texts->size();
      expr_texts->push_back(m.expr);
    }
    const int sslot = c.from.index_of(m.source.def), tslot = c.to.index_of(m.target.def);
    if (src_columnar) e.src_col = 0x100000ull * (uint64_t)(sslot + 1);
    if (dst_columnar) e.dst_col = 0x100000ull * (uint64_t)(tslot + 1);
    if (pm && &c.to.members[(size_t)tslot] == pm && m.target.def.datatype.kind == PST_VEC3F64 && !bounds_done) { e.bounds = 1; bounds_done = true; }
    generic.push_back(e);
  }
  if (generic.size() > PST_PLAN_MAX_ENTRIES) return PST_PLAN_INTERPRETED;  // several launches: interpreted
  const uint32_t ss = (uint32_t)c.from.size, ds = (uint32_t)c.to.size;
  const uint32_t tile = pick_tile(!src_columnar, ss, !dst_columnar, ds);
  if (tile < 1) return PST_PLAN_DIRECT;
  bool wants_bounds = false;
  *plan = build_plan(!src_columnar, src_columnar ? 0 : 0x10000000ull, ss, !dst_columnar, dst_columnar ? 0 : 0x20000000ull, ds, (uint64_t)1 << 30,
                     generic.data(), generic.size(), tile, false, with_bounds, &wants_bounds);
  if (wants_bounds) plan->h.bounds_partials = 0x30000000ull;
  plan->expr_texts = expr_texts->empty() ? nullptr : expr_texts;
  return PST_PLAN_INTERPRETED;
}

}  // namespace pst

using namespace pst;

static AttributeDef def_from(const char* name, const pst_datatype* dt) { return AttributeDef{not_null(name, "name"), DataType::from_c(dt)}; }

static void install_mapping(pst_converter& c, Mapping&& m, const AttributeDef& to_attribute) {
  c.las_decode_format = -2;
  c.identity_records = -2;
  c.las_typed_format = -2;
  for (auto& prev : c.mappings)
    if (prev.target.def == to_attribute) { prev = std::move(m); return; }  // replace the mapping for this target (:168-176)
  c.mappings.push_back(std::move(m));
}

extern "C" {

int pst_point_converter_create(const pst_layout* from, const pst_layout* to, pst_point_converter** out) {
  PST_API_BEGIN
  auto c = std::make_unique<pst_point_converter>();
  c->from = not_null(from, "from")->l;
  c->to = not_null(to, "to")->l;
  check_layout_fits_kernels(c->from, "pst_point_converter_create");
  check_layout_fits_kernels(c->to, "pst_point_converter_create");
  for (const Member& from_attr : c->from.members) {  // :70-91
    const Member* to_attr = c->to.find_by_name(from_attr.def.name);
    if (!to_attr) continue;                                       // .filter(has_attribute_with_name)
    if (from_attr.def.datatype == to_attr->def.datatype) continue;  // get_converter_for_attributes -> None -> filter_map drops it
    require_convertible(from_attr.def.datatype, to_attr->def.datatype);  // "Invalid conversion X -> Y" :267-269
    PlanEntry e = identity_entry(from_attr, *to_attr);
    e.dst_ct = (uint8_t)to_attr->def.datatype.comp_type();
    e.convert = 1u;
    c->entries.push_back(e);
  }
  *not_null(out, "out") = c.release();
  PST_API_END
}
int pst_point_converter_destroy(pst_point_converter* c) {
  delete c;
  return PST_OK;
}
int pst_point_converter_num_converters(const pst_point_converter* c, size_t* out) {
  PST_API_BEGIN
  *not_null(out, "out") = not_null(c, "converter")->entries.size();
  PST_API_END
}
// RawPointConverter::convert (:104-108) applied to `count` points: target bytes no converter writes keep their values
int pst_point_converter_convert(const pst_point_converter* c, const pst_buffer* src, size_t src_first, pst_buffer* dst, size_t dst_first, size_t count) {
  PST_API_BEGIN
  not_null(c, "converter");
  not_null(src, "src");
  not_null(dst, "dst");
  if (src->columnar || dst->columnar) throw Error(PST_ERR_INVALID_ARGUMENT, "RawPointConverter::convert works on the bytes of one interleaved point");
  // the reference's safety contract ("source_point must ... have the exact same PointLayout as the one passed to from_to") is checked here
  if (src->layout != c->from) throw Error(PST_ERR_LAYOUT_MISMATCH, "source point does not have the PointLayout passed to RawPointConverter::from_to");
  if (dst->layout != c->to) throw Error(PST_ERR_LAYOUT_MISMATCH, "target point does not have the PointLayout passed to RawPointConverter::from_to");
  if (src_first + count < src_first || src_first + count > src->len || dst_first + count < dst_first || dst_first + count > dst->len)
    throw Error(PST_ERR_RANGE, "point range out of bounds");
  if (count && !c->entries.empty()) {
    hipStream_t s = current_stream();
    execute_entries(true, aos_addr(*src, src_first), (uint32_t)c->from.size, true, aos_addr(*dst, dst_first), (uint32_t)c->to.size, count, c->entries, true, s);
    stream_sync(s);
  }
  PST_API_END
}

int pst_converter_create(const pst_layout* from, const pst_layout* to, int with_default, pst_converter** out) {
  PST_API_BEGIN
  auto c = std::make_unique<pst_converter>();
  c->from = not_null(from, "from")->l;
  c->to = not_null(to, "to")->l;
  check_layout_fits_kernels(c->from, "pst_converter_create");
  check_layout_fits_kernels(c->to, "pst_converter_create");
  for (const Member& to_attr : c->to.members) {  // one default mapping per TARGET attribute, matched BY NAME (:112-143)
    const Member* from_attr = c->from.find_by_name(to_attr.def.name);
    if (!from_attr) {
      if (with_default) continue;
      throw Error(PST_ERR_MISSING_ATTRIBUTE,
                  "Attribute not found in `from_layout`! When calling `BufferLayoutConverter::for_layouts`, the source PointLayout must "
                  "contain all attributes from the target PointLayout. If you want to use default values for attributes that are not "
                  "present in the source layout, use `BufferLayoutConverter::for_layouts_with_default` instead!");
    }
    c->mappings.push_back(make_default_mapping(*from_attr, to_attr));
  }
  *not_null(out, "out") = c.release();
  PST_API_END
}
int pst_converter_destroy(pst_converter* c) { delete c; return PST_OK; }

int pst_converter_set_custom_mapping(pst_converter* c, const char* from_name, const pst_datatype* from_dt, const char* to_name,
                                     const pst_datatype* to_dt) {
  PST_API_BEGIN
  not_null(c, "converter");
  const AttributeDef fa = def_from(from_name, from_dt), ta = def_from(to_name, to_dt);
  const Member* fm = c->from.find(fa);
  if (!fm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "from_attribute not found in source PointLayout");
  const Member* tm = c->to.find(ta);
  if (!tm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "to_attribute not found in target PointLayout");
  install_mapping(*c, make_default_mapping(*fm, *tm), ta);
  PST_API_END
}
int pst_converter_set_custom_mapping_with_transformation(pst_converter* c, const char* from_name, const pst_datatype* from_dt,
                                                         const char* to_name, const pst_datatype* to_dt, const pst_transform* xf,
                                                         int apply_to_source) {
  PST_API_BEGIN
  not_null(c, "converter");
  not_null(xf, "transform");
  const AttributeDef fa = def_from(from_name, from_dt), ta = def_from(to_name, to_dt);
  const Member* fm = c->from.find(fa);
  if (!fm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "from_attribute not found in source PointLayout");
  const Member* tm = c->to.find(ta);
  if (!tm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "to_attribute not found in target PointLayout");
  XfDesc x;
  x.kind = xf->kind;
  x.datatype = DataType::from_c(&xf->datatype);
  for (int i = 0; i < 3; ++i) { x.scale[i] = xf->scale[i]; x.offset[i] = xf->offset[i]; }
  x.shift = xf->shift;
  x.mask = xf->mask;
  const DataType& expect = apply_to_source ? fm->def.datatype : tm->def.datatype;  // :209-213
  if (x.datatype != expect)
    throw Error(PST_ERR_TRANSFORM_TYPE_MISMATCH,
                "assertion `left == right` failed: T::data_type() is " + x.datatype.display() + " but the attribute is " + expect.display());
  Mapping m = make_default_mapping(*fm, *tm);
  validate_transform(x);
  m.xf = x;
  m.apply_to_source = apply_to_source != 0;
  install_mapping(*c, std::move(m), ta);
  PST_API_END
}
// set_custom_mapping_with_transformation with the closure given as a device expression (expr.cpp: names v, x, y, z, c, i; C++ expression
// syntax; the result converted to T with Rust `as`).  T is the source attribute's datatype when apply_to_source, the target's otherwise
// (buffer_conversion.rs:209-213) -- there is no separate T to mismatch.  Shape errors (components, datatype kinds) are reported here, syntax
// errors by the first conversion (PST_ERR_UNSUPPORTED_TRANSFORM with the compiler's log).
int pst_converter_set_custom_mapping_with_expression(pst_converter* c, const char* from_name, const pst_datatype* from_dt, const char* to_name,
                                                     const pst_datatype* to_dt, const char* expr, int apply_to_source) {
  PST_API_BEGIN
  not_null(c, "converter");
  not_null(expr, "expr");
  const AttributeDef fa = def_from(from_name, from_dt), ta = def_from(to_name, to_dt);
  const Member* fm = c->from.find(fa);
  if (!fm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "from_attribute not found in source PointLayout");
  const Member* tm = c->to.find(ta);
  if (!tm) throw Error(PST_ERR_MISSING_ATTRIBUTE, "to_attribute not found in target PointLayout");
  Mapping m = make_default_mapping(*fm, *tm);
  validate_expression_mapping(fm->def.datatype, tm->def.datatype, apply_to_source != 0, expr);
  m.expr = expr;
  m.apply_to_source = apply_to_source != 0;
  install_mapping(*c, std::move(m), ta);
  PST_API_END
}
// Ahead-of-time specialisation: compiles (hipRTC, cached) the kernel a conversion between buffers of these storage kinds will take, so
// that the first call already runs it.  *plan_kind = the PST_PLAN_* family such a call will use.
int pst_converter_prepare(const pst_converter* c, int src_columnar, int dst_columnar, int with_bounds, uint32_t* plan_kind) {
  PST_API_BEGIN
  not_null(c, "converter");
  ConvertPlan plan{};
  std::vector<std::string> texts;
  uint32_t kind = plan_for_storage(*c, src_columnar != 0, dst_columnar != 0, with_bounds != 0, &plan, &texts);
  if (kind == PST_PLAN_INTERPRETED && plan.h.n_entries) {
    std::string err;
    bool in_tree = false;
    if (pstk::prepare_convert(plan, !src_columnar, !dst_columnar, &err, &in_tree)) kind = in_tree ? PST_PLAN_STATIC : PST_PLAN_JIT;
    else if (!err.empty() && err.find("not eligible") == std::string::npos && err != "PST_JIT=0") set_last_error(err);
  }
  if (plan_kind) *plan_kind = kind;
  PST_API_END
}
int pst_converter_family_choice(const pst_converter* c, int dst_columnar, int with_bounds, int* choice, float ms2[2]) {
  PST_API_BEGIN
  not_null(c, "converter");
  // dst_columnar: 0 = records from records, 1 = columns (from records), 2 = records from COLUMNS (its own slot since round 6)
  const int d = dst_columnar == 2 ? 2 : (dst_columnar ? 1 : 0), b = with_bounds ? 1 : 0;
  *not_null(choice, "choice") = c->family_choice[d][b];
  if (ms2) { ms2[0] = c->family_ms[d][b][0]; ms2[1] = c->family_ms[d][b][1]; }
  PST_API_END
}
// The explicit form of the measurement (stream-ordered callers: once, before their loop).  Converts the range like pst_converter_convert_into_range
// (same bytes, several times) and waits for the stream; a no-op when the pairing was measured already or has one family.
int pst_converter_measure_families(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1, int with_bounds) {
  PST_API_BEGIN
  not_null(c, "converter");
  Workspace& ws = workspace();
  double* dev_rec = (double*)(ws.dev + Workspace::kWorkspaceBytes - 64);
  const bool has_pos = c->to.find_by_name("Position3D") != nullptr;
  convert_range(*c, *not_null(src, "src"), s0, s1, *not_null(dst, "dst"), t0, t1, (with_bounds && has_pos && t1 > t0) ? dev_rec : nullptr, current_stream(), kMeasureFamilies);
  stream_sync(current_stream());
  PST_API_END
}
// The translation unit the run-time compiler is given for this converter and storage pairing (empty when the plan takes another family).
int pst_converter_jit_source(const pst_converter* c, int src_columnar, int dst_columnar, int with_bounds, char* buf, size_t cap, size_t* needed) {
  PST_API_BEGIN
  not_null(c, "converter");
  ConvertPlan plan{};
  std::string src;
  std::vector<std::string> texts;
  if (plan_for_storage(*c, src_columnar != 0, dst_columnar != 0, with_bounds != 0, &plan, &texts) == PST_PLAN_INTERPRETED && plan.h.n_entries) {
    pstjit::QuadSpec spec;
    if (pstjit::spec_from_plan(plan, !src_columnar, !dst_columnar, &spec)) src = pstjit::spec_source(spec);
  }
  if (needed) *needed = src.size() + 1;
  if (buf && cap) {
    const size_t k = std::min(cap - 1, src.size());
    memcpy(buf, src.data(), k);
    buf[k] = 0;
  }
  PST_API_END
}
// hipRTC compilation of a translation unit against the embedded device headers, for gfx950, WITHOUT a device (CPU check of the generator).
int pst_jit_compile_source(const char* source, void* code_buf, size_t code_cap, size_t* code_bytes, char* log, size_t log_cap) {
  PST_API_BEGIN
  std::string err;
  const std::vector<char> code = pstjit::compile_source(not_null(source, "source"), "gfx950", &err);
  if (code_bytes) *code_bytes = code.size();
  if (code_buf && code_cap) memcpy(code_buf, code.data(), std::min(code_cap, code.size()));
  if (log && log_cap) {
    const size_t k = std::min(log_cap - 1, err.size());
    memcpy(log, err.data(), k);
    log[k] = 0;
  }
  if (code.empty()) throw Error(PST_ERR_UNSUPPORTED, err.empty() ? "hipRTC produced no code" : err);
  PST_API_END
}
int pst_jit_get_stats(pst_jit_stats* out) {
  PST_API_BEGIN
  const pstjit::Stats s = pstjit::stats();
  not_null(out, "out");
  out->compiled = s.compiled; out->disk_hits = s.disk_hits; out->memory_hits = s.memory_hits; out->failures = s.failures; out->launches = s.launches;
  out->compile_seconds = s.compile_seconds;
  PST_API_END
}
int pst_jit_set_mode(int mode) {
  PST_API_BEGIN
  if (mode < -1 || mode > 2) throw Error(PST_ERR_INVALID_ARGUMENT, "mode must be -1 (environment), 0 (off), 1 (async) or 2 (sync)");
  pstjit::set_mode(mode);
  PST_API_END
}
int pst_last_plan_kinds(uint32_t* mask) {
  PST_API_BEGIN
  *not_null(mask, "mask") = pstk::plan_kinds();
  PST_API_END
}

int pst_converter_num_mappings(const pst_converter* c, size_t* out) { PST_API_BEGIN *not_null(out, "out") = not_null(c, "converter")->mappings.size(); PST_API_END }
int pst_converter_get_mapping(const pst_converter* c, size_t index, pst_mapping_info* out) {
  PST_API_BEGIN
  const auto& ms = not_null(c, "converter")->mappings;
  if (index >= ms.size()) throw Error(PST_ERR_RANGE, "index out of bounds");
  const Mapping& m = ms[index];
  not_null(out, "out")->source_name = m.source.def.name.c_str();
  out->target_name = m.target.def.name.c_str();
  out->source_datatype = m.source.def.datatype.to_c();
  out->target_datatype = m.target.def.datatype.to_c();
  out->source_offset = m.source.offset;
  out->target_offset = m.target.offset;
  out->has_converter = m.has_converter;
  out->transform_kind = m.xf ? m.xf->kind : 0u;
  out->apply_to_source = m.xf ? (int)m.apply_to_source : 0;
  out->reserved = 0;
  PST_API_END
}

int pst_converter_convert_into_range_async(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1) {
  PST_API_BEGIN
  convert_range(*not_null(c, "converter"), *not_null(src, "src"), s0, s1, *not_null(dst, "dst"), t0, t1, nullptr, current_stream());
  PST_API_END
}
int pst_converter_convert_into_range(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1) {
  PST_API_BEGIN
  convert_range(*not_null(c, "converter"), *not_null(src, "src"), s0, s1, *not_null(dst, "dst"), t0, t1, nullptr, current_stream(), kMeasureFamilies);
  stream_sync(current_stream());
  PST_API_END
}

// convert :242-259.  The reference zero-fills the whole target (`resize`) and then overwrites most of it.  Here only the
// bytes NO mapping writes are zero-filled (unmapped columns; interleaved records that are not fully covered) — the
// resulting buffer is byte-identical, one full write pass over the target cheaper.
int pst_converter_convert(const pst_converter* c, pst_buffer* src, uint32_t out_storage, pst_buffer** out) {
  PST_API_BEGIN
  not_null(c, "converter");
  not_null(src, "src");
  if (src->layout != c->from) throw Error(PST_ERR_LAYOUT_MISMATCH, "assertion `left == right` failed: source_buffer.point_layout() != from_layout");
  pst_layout tl{c->to};
  pst_buffer* target = nullptr;
  int rc = pst_buffer_create(&tl, out_storage, PST_MEM_DEVICE, &target);
  if (rc != PST_OK) return rc;
  std::unique_ptr<pst_buffer> guard(target);
  const size_t n = src->len;
  if (n > 0) {
    resize_buffer(*target, n, /*zero_fill=*/false);
    hipStream_t s = current_stream();
    if (target->columnar) {
      for (size_t a = 0; a < c->to.members.size(); ++a) {
        bool written = false;
        for (const Mapping& m : c->mappings) written = written || (m.target.def == c->to.members[a].def);
        if (!written && c->to.members[a].size) PST_HIP_CHECK(hipMemsetAsync(target->columns[a], 0, n * c->to.members[a].size, s));
      }
    } else {
      std::vector<uint8_t> covered(c->to.size, 0);
      for (const Mapping& m : c->mappings)
        for (uint64_t b = 0; b < m.target.size; ++b) covered[m.target.offset + b] = 1;
      const bool full = std::all_of(covered.begin(), covered.end(), [](uint8_t v) { return v != 0; });
      if (!full && c->to.size) PST_HIP_CHECK(hipMemsetAsync(target->data, 0, n * c->to.size, s));
    }
    convert_range(*c, *src, 0, n, *target, 0, n, nullptr, s, kMeasureFamilies);
    stream_sync(s);
  }
  *not_null(out, "out") = guard.release();
  PST_API_END
}

int pst_converter_convert_into_range_with_bounds_async(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst,
                                                       size_t t0, size_t t1, double* device_out6) {
  PST_API_BEGIN
  convert_range(*not_null(c, "converter"), *not_null(src, "src"), s0, s1, *not_null(dst, "dst"), t0, t1, not_null(device_out6, "device_out6"),
                current_stream());
  PST_API_END
}
int pst_converter_convert_into_range_with_bounds(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0,
                                                 size_t t1, double out_min[3], double out_max[3], int* has_value) {
  PST_API_BEGIN
  not_null(dst, "dst");
  not_null(has_value, "has_value");
  Workspace& ws = workspace();
  hipStream_t s = current_stream();
  double* host_rec = (double*)ws.pinned;
  double* dev_rec = results_to_host() ? host_rec : (double*)(ws.dev + Workspace::kWorkspaceBytes - 64);
  const bool has_pos = not_null(c, "converter")->to.find_by_name("Position3D") != nullptr;
  convert_range(*c, *not_null(src, "src"), s0, s1, *dst, t0, t1, (has_pos && t1 > t0) ? dev_rec : nullptr, s, kMeasureFamilies);
  // calculate_bounds(target): None for an empty range or a layout without Position3D (bounds.rs:12-21)
  if (!has_pos || t1 <= t0) {
    stream_sync(s);
    *has_value = 0;
    return PST_OK;
  }
  if (dev_rec != host_rec) PST_HIP_CHECK(hipMemcpyAsync(host_rec, dev_rec, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  stream_sync(s);
  check_bounds_record(host_rec, out_min, out_max);
  *has_value = 1;
  PST_API_END
}

}  // extern "C"

// view_attribute_with_conversion::<T>(attribute) collected, point_buffer.rs:322-330 / buffer_views.rs:533-650: the attribute is looked up
// BY NAME (:549-552, expect), the stored datatype is converted to T with the `as` table (:553-561: convert_unit for equal datatypes, an
// unlisted pair is Err("Conversion between attribute types is impossible")).  One launch of the column kernels (columnar source) or of
// the interleaved -> columnar tile kernel into a dense array of T.
namespace pst {
static void read_attribute_converted(const pst_buffer& b, const char* name, const pst_datatype* target_dt, size_t first, size_t count, void* device_dst,
                                     hipStream_t s) {
  const DataType t = DataType::from_c(target_dt);
  const Member* m = b.layout.find_by_name(not_null(name, "name"));
  if (!m) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  if (m->def.datatype != t && !convertible(m->def.datatype, t))
    throw Error(PST_ERR_INVALID_CONVERSION, "Conversion between attribute types is impossible (" + m->def.datatype.display() + " -> " + t.display() + ")");
  if (first + count < first || first + count > b.len)
    throw Error(PST_ERR_RANGE, "range end index " + std::to_string(first + count) + " out of range for buffer of length " + std::to_string(b.len));
  if (count == 0 || t.size() == 0) return;
  ensure_device();
  Member target{AttributeDef{m->def.name, t}, 0, t.size()};
  PlanEntry e = identity_entry(*m, target);
  e.dst_ct = (uint8_t)t.comp_type();
  e.convert = m->def.datatype != t ? 1u : 0u;
  e.dst_col = (uint64_t)(uintptr_t)not_null(device_dst, "dst");
  const size_t slot = (size_t)(m - b.layout.members.data());
  if (b.columnar) {
    e.src_col = col_addr(b, slot, first);
    execute_entries(false, 0, 0, false, 0, 0, count, {e}, false, s);
  } else {
    execute_entries(true, aos_addr(b, first), (uint32_t)b.layout.size, false, 0, 0, count, {e}, true, s);
  }
  PST_HIP_CHECK(hipGetLastError());
}
}  // namespace pst

extern "C" {
int pst_buffer_read_attribute_converted_device(const pst_buffer* b, const char* name, const pst_datatype* target_dt, size_t first, size_t count,
                                               void* device_dst) {
  PST_API_BEGIN
  read_attribute_converted(*not_null(b, "buffer"), name, target_dt, first, count, device_dst, current_stream());
  PST_API_END
}
int pst_buffer_read_attribute_converted(const pst_buffer* b, const char* name, const pst_datatype* target_dt, size_t first, size_t count, void* host_dst) {
  PST_API_BEGIN
  not_null(b, "buffer");
  const size_t bytes = count * DataType::from_c(target_dt).size();
  hipStream_t s = current_stream();
  uint8_t* tmp = bytes ? dev_alloc(bytes, PST_MEM_DEVICE) : nullptr;
  struct Free { uint8_t* p; ~Free() { if (p) dev_free(p, PST_MEM_DEVICE); } } guard{tmp};
  static uint8_t dummy;
  read_attribute_converted(*b, name, target_dt, first, count, tmp ? (void*)tmp : (void*)&dummy, s);
  if (bytes) {
    PST_HIP_CHECK(hipMemcpyAsync(not_null(host_dst, "host_dst"), tmp, bytes, hipMemcpyDeviceToHost, s));
    PST_HIP_CHECK(hipStreamSynchronize(s));
  }
  PST_API_END
}
}  // extern "C"
