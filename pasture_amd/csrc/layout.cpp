// PointLayout model + its C ABI (host only; needs no device).
// Reference: pasture-core/src/layout/point_layout.rs:23-127 (datatypes), :648-997 (PointLayout).
#include "core.hpp"

namespace pst {

static thread_local std::string t_last_error;
void set_last_error(const std::string& m) { t_last_error = m; }
const char* last_error_cstr() { return t_last_error.c_str(); }

DataType DataType::from_c(const pst_datatype* d) {
  not_null(d, "datatype");
  if (d->kind > PST_CUSTOM) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid datatype kind " + std::to_string(d->kind));
  DataType t;
  t.kind = d->kind;
  if (d->kind == PST_BYTEARRAY || d->kind == PST_CUSTOM) t.size_param = d->size_param;
  if (d->kind == PST_CUSTOM) {
    t.align_param = d->align_param;
    std::memcpy(t.uuid.data(), d->uuid, 16);
  }
  return t;
}
pst_datatype DataType::to_c() const {
  pst_datatype d{};
  d.kind = kind;
  d.size_param = size_param;
  d.align_param = align_param;
  std::memcpy(d.uuid, uuid.data(), 16);
  return d;
}
// sizes: point_layout.rs:72-95
uint64_t DataType::size() const {
  static const uint64_t fixed[] = {1, 1, 2, 2, 4, 4, 8, 8, 4, 8, /*Vec3u8*/ 3, /*Vec3u16*/ 6, /*Vec3f32*/ 12, /*Vec3i32*/ 12,
                                   /*Vec3f64*/ 24, /*Vec4u8*/ 4};
  return kind <= PST_VEC4U8 ? fixed[kind] : size_param;
}
// alignments: point_layout.rs:98-126 = align_of the Rust types (values listed at pasture-derive/src/lib.rs:36-56)
uint64_t DataType::min_alignment() const {
  static const uint64_t fixed[] = {1, 1, 2, 2, 4, 4, 8, 8, 4, 8, 1, 2, 4, 4, 8, 1};
  if (kind <= PST_VEC4U8) return fixed[kind];
  return kind == PST_BYTEARRAY ? 1 : align_param;
}
std::string DataType::display() const {
  static const char* names[] = {"U8", "I8", "U16", "I16", "U32", "I32", "U64", "I64", "F32", "F64",
                                "Vec3<u8>", "Vec3<u16>", "Vec3<f32>", "Vec3<i32>", "Vec3<f64>", "Vec4<u8>"};
  if (kind <= PST_VEC4U8) return names[kind];
  if (kind == PST_BYTEARRAY) return "ByteArray[" + std::to_string(size_param) + "]";
  return "Custom";
}
CompType DataType::comp_type() const {
  switch (kind) {
    case PST_U8: case PST_VEC3U8: return CT_U8;
    case PST_I8: return CT_I8;
    case PST_U16: case PST_VEC3U16: return CT_U16;
    case PST_I16: return CT_I16;
    case PST_U32: return CT_U32;
    case PST_I32: case PST_VEC3I32: return CT_I32;
    case PST_U64: return CT_U64;
    case PST_I64: return CT_I64;
    case PST_F32: case PST_VEC3F32: return CT_F32;
    case PST_F64: case PST_VEC3F64: return CT_F64;
    default: return CT_U8;  // Vec4u8 / ByteArray / Custom: raw bytes
  }
}
uint32_t DataType::num_components() const {
  if (is_scalar()) return 1;
  if (is_vec3()) return 3;
  return (uint32_t)size();  // opaque: one U8 "component" per byte
}
bool DataType::operator==(const DataType& o) const {
  if (kind != o.kind) return false;
  if (kind == PST_BYTEARRAY) return size_param == o.size_param;
  if (kind == PST_CUSTOM) return size_param == o.size_param && align_param == o.align_param && uuid == o.uuid;
  return true;
}

// std::alloc::Layout::from_size_align: align must be a non-zero power of two
static void check_size_align(uint64_t size, uint64_t align) {
  if (align == 0 || (align & (align - 1)) != 0 || size > (uint64_t)INT64_MAX - (align - 1))
    throw Error(PST_ERR_INVALID_LAYOUT, "Could not create memory layout for PointLayout");
}

// add_attribute, point_layout.rs:778-822
void Layout::add_attribute(const AttributeDef& def, bool packed, uint64_t max_alignment) {
  if (find_by_name(def.name))
    throw Error(PST_ERR_DUPLICATE_ATTRIBUTE, "Point attribute " + def.name + " is already present in this PointLayout!");
  const uint64_t type_align = def.datatype.min_alignment();
  const uint64_t field_align = packed ? std::min(max_alignment, type_align) : type_align;
  const uint64_t next = members.empty() ? 0 : members.back().offset + members.back().size;  // :985-996
  const uint64_t offset = align_up(next, field_align);
  const uint64_t new_align = packed ? std::min(max_alignment, align) : std::max(align, type_align);
  const uint64_t new_size = align_up(std::max(size, offset + def.datatype.size()), new_align);
  check_size_align(new_size, new_align);
  members.push_back(Member{def, offset, def.datatype.size()});
  size = new_size;
  align = new_align;
}

// from_members_and_alignment, point_layout.rs:719-759
Layout Layout::from_members_and_alignment(const std::vector<Member>& ms, uint64_t type_alignment) {
  for (size_t i = 0; i < ms.size(); ++i)
    for (size_t j = i + 1; j < ms.size(); ++j)
      if (ms[i].def.name == ms[j].def.name)
        throw Error(PST_ERR_INVALID_LAYOUT, "PointLayout::from_attributes_and_offsets: All attributes must have unique names!");
  std::vector<std::pair<uint64_t, uint64_t>> spans;
  for (auto& m : ms) spans.emplace_back(m.offset, m.offset + m.size);
  std::sort(spans.begin(), spans.end());
  for (size_t i = 1; i < spans.size(); ++i)
    if (spans[i - 1].second > spans[i].first)
      throw Error(PST_ERR_INVALID_LAYOUT,
                  "PointLayout::from_attributes_and_offsets: All attributes must span non-overlapping memory regions!");
  uint64_t end = 0, max_off = 0;
  for (auto& m : ms)
    if (m.offset >= max_off) { max_off = m.offset; end = m.offset + m.size; }
  Layout l;
  l.members = ms;
  l.size = align_up(end, type_alignment);
  l.align = type_alignment;
  check_size_align(l.size, l.align);
  return l;
}
const Member* Layout::find(const AttributeDef& d) const {
  for (auto& m : members) if (m.def == d) return &m;
  return nullptr;
}
const Member* Layout::find_by_name(const std::string& n) const {
  for (auto& m : members) if (m.def.name == n) return &m;
  return nullptr;
}
int Layout::index_of(const AttributeDef& d) const {
  for (size_t i = 0; i < members.size(); ++i) if (members[i].def == d) return (int)i;
  return -1;
}

const char* last_error_cstr();
}  // namespace pst

using namespace pst;

extern "C" {

const char* pst_last_error(void) { return pst::last_error_cstr(); }

int pst_layout_create(pst_layout** out) { PST_API_BEGIN *not_null(out, "out") = new pst_layout(); PST_API_END }
int pst_layout_destroy(pst_layout* l) { delete l; return PST_OK; }
int pst_layout_clone(const pst_layout* l, pst_layout** out) { PST_API_BEGIN *not_null(out, "out") = new pst_layout{not_null(l, "layout")->l}; PST_API_END }
int pst_layout_add_attribute(pst_layout* l, const char* name, const pst_datatype* dt, uint32_t packed, uint64_t max_alignment) {
  PST_API_BEGIN
  not_null(l, "layout")->l.add_attribute(AttributeDef{not_null(name, "name"), DataType::from_c(dt)}, packed != 0, max_alignment);
  PST_API_END
}
int pst_layout_from_members(const pst_member* members, size_t n, uint64_t type_alignment, pst_layout** out) {
  PST_API_BEGIN
  std::vector<Member> ms;
  for (size_t i = 0; i < n; ++i) {
    DataType t = DataType::from_c(&members[i].datatype);
    ms.push_back(Member{AttributeDef{not_null(members[i].name, "name"), t}, members[i].offset, t.size()});
  }
  *not_null(out, "out") = new pst_layout{Layout::from_members_and_alignment(ms, type_alignment)};
  PST_API_END
}
int pst_layout_num_attributes(const pst_layout* l, size_t* out) { PST_API_BEGIN *not_null(out, "out") = not_null(l, "layout")->l.members.size(); PST_API_END }
int pst_layout_get_member(const pst_layout* l, size_t index, pst_member* out) {
  PST_API_BEGIN
  const auto& ms = not_null(l, "layout")->l.members;
  if (index >= ms.size()) throw Error(PST_ERR_RANGE, "index out of bounds: the len is " + std::to_string(ms.size()) + " but the index is " + std::to_string(index));
  not_null(out, "out")->name = ms[index].def.name.c_str();
  out->datatype = ms[index].def.datatype.to_c();
  out->offset = ms[index].offset;
  out->size = ms[index].size;
  PST_API_END
}
int pst_layout_size_of_point_entry(const pst_layout* l, uint64_t* out) { PST_API_BEGIN *not_null(out, "out") = not_null(l, "layout")->l.size; PST_API_END }
int pst_layout_alignment(const pst_layout* l, uint64_t* out) { PST_API_BEGIN *not_null(out, "out") = not_null(l, "layout")->l.align; PST_API_END }
int pst_layout_equals(const pst_layout* a, const pst_layout* b, int* out) { PST_API_BEGIN *not_null(out, "out") = not_null(a, "a")->l == not_null(b, "b")->l; PST_API_END }

}  // extern "C"
