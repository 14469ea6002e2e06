// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
// CPU restatement (C++17, gcc) of the reference's per-point attribute-transform path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, link
// or execute this code, and only as the checker / reported baseline.
//
// Faithful shape on purpose (attribute-major loops, one function-pointer call per value,
// one hash lookup per point on columnar buffers, zero-fill on resize): the same code is the
// parity oracle and the timed single-thread CPU baseline, so it is NOT optimised.
//
// Parity pin: the reference is Rust and cannot be compiled here (no rustc/cargo, no crates,
// no network).  This restatement is pinned against the reference's own golden data instead:
//   * LAS fixtures pasture-io/resources/test/10_points_format_{0..10}.las with the expected
//     columns of pasture-io/src/las/test_util.rs:46-188 (tests/test_oracle_las_golden.py)
//   * layout offsets/sizes: point_layout.rs doc-tests :684-691,:713-717,:773-776,:924-927;
//     las_layout.rs:278; las_types.rs:37,93,...
//   * AABB known answer math/bounds.rs:305-315; normals known answers normal_estimation.rs:503-610
//   * align_to known answers math/arithmetic.rs:78-85, MinMax doc-tests math/minmax.rs:17-32
// Parity UNPINNED (third-party arithmetic absent from /root/reference): kd-tree 0.3.0 tie-breaking
// among equidistant neighbours, nalgebra 0.32 `DMatrix::diagonal()` copy semantics
// (normal_estimation.rs:446-449 restated as a no-op; alternative behind ORC_DIAGONAL_SUBTRACT),
// Rust `as` semantics (language-defined; pinned by SURVEY Appendix C table in tests).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// =====================================================================================
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace orc {

// ----------------------------------------------------------------------------------
// Panics.  The reference signals precondition violations with panic!/assert!/expect.
// The oracle throws; the C API turns this into a status code + message.
// ----------------------------------------------------------------------------------
enum Status : int {
  OK = 0,
  ERR_INVALID_ARGUMENT = 1,
  ERR_LAYOUT_MISMATCH = 2,
  ERR_RANGE = 3,
  ERR_MISSING_ATTRIBUTE = 4,
  ERR_INVALID_CONVERSION = 5,
  ERR_TRANSFORM_TYPE_MISMATCH = 6,
  ERR_UNSUPPORTED_TRANSFORM = 7,
  ERR_DUPLICATE_ATTRIBUTE = 8,
  ERR_INVALID_LAYOUT = 9,
  ERR_BOUNDS_INVALID = 10,
  ERR_TOO_FEW_POINTS = 11,
  ERR_K_TOO_SMALL = 12,
  ERR_NOT_ENOUGH_NEIGHBOURS = 13,
  ERR_UNSUPPORTED_ATTRIBUTE = 14,
  ERR_UNSUPPORTED = 23,
};

struct Panic : std::runtime_error {
  int code;
  Panic(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

// ----------------------------------------------------------------------------------
// PointAttributeDataType — pasture-core/src/layout/point_layout.rs:23-127
// Kind codes follow the enum declaration order at :25-50.
// ----------------------------------------------------------------------------------
enum Kind : uint32_t {
  U8 = 0, I8, U16, I16, U32, I32, U64, I64, F32, F64,
  Vec3u8, Vec3u16, Vec3f32, Vec3i32, Vec3f64, Vec4u8, ByteArray, Custom
};

struct DataType {
  Kind kind = U8;
  uint64_t size_param = 0;   // ByteArray(length) / Custom.size
  uint64_t align_param = 0;  // Custom.min_alignment
  std::array<uint8_t, 16> uuid{};

  // point_layout.rs:72-95
  uint64_t size() const {
    switch (kind) {
      case U8: case I8: return 1;
      case U16: case I16: return 2;
      case U32: case I32: case F32: return 4;
      case U64: case I64: case F64: return 8;
      case Vec3u8: return 3;
      case Vec3u16: return 6;
      case Vec3i32: case Vec3f32: return 12;
      case Vec3f64: return 24;
      case Vec4u8: return 4;
      case ByteArray: return size_param;
      case Custom: return size_param;
    }
    return 0;
  }
  // point_layout.rs:98-126 (align_of the Rust types; numeric values spelled out at
  // pasture-derive/src/lib.rs:36-56)
  uint64_t min_alignment() const {
    switch (kind) {
      case U8: case I8: return 1;
      case U16: case I16: return 2;
      case U32: case I32: case F32: return 4;
      case U64: case I64: case F64: return 8;
      case Vec3u8: return 1;
      case Vec3u16: return 2;
      case Vec3i32: case Vec3f32: return 4;
      case Vec3f64: return 8;
      case Vec4u8: return 1;
      case ByteArray: return 1;
      case Custom: return align_param;
    }
    return 1;
  }
  bool operator==(const DataType& o) const {
    if (kind != o.kind) return false;
    if (kind == ByteArray) return size_param == o.size_param;
    if (kind == Custom) return size_param == o.size_param && align_param == o.align_param && uuid == o.uuid;
    return true;
  }
  bool operator!=(const DataType& o) const { return !(*this == o); }
  // Display impl point_layout.rs:129-158
  std::string display() const {
    static const char* names[] = {"U8", "I8", "U16", "I16", "U32", "I32", "U64", "I64", "F32", "F64",
                                  "Vec3<u8>", "Vec3<u16>", "Vec3<f32>", "Vec3<i32>", "Vec3<f64>", "Vec4<u8>"};
    if (kind <= Vec4u8) return names[kind];
    if (kind == ByteArray) return "ByteArray[" + std::to_string(size_param) + "]";
    return "Custom";
  }
  static DataType of(Kind k) { DataType d; d.kind = k; return d; }
};

// math/arithmetic.rs:8-66 (Alignable::align_to; alignment 0 returns the value itself)
inline uint64_t align_to(uint64_t v, uint64_t alignment) {
  if (alignment == 0) return v;
  return ((v + alignment - 1) / alignment) * alignment;
}

// PointAttributeDefinition — point_layout.rs:261-341 (name + datatype, derive(PartialEq, Hash))
struct AttributeDef {
  std::string name;
  DataType datatype;
  uint64_t size() const { return datatype.size(); }
  bool operator==(const AttributeDef& o) const { return name == o.name && datatype == o.datatype; }
  bool operator!=(const AttributeDef& o) const { return !(*this == o); }
  std::string display() const { return "[" + name + ";" + datatype.display() + "]"; }
};
struct AttributeDefHash {
  size_t operator()(const AttributeDef& d) const {
    // The reference hashes name + datatype with SipHash-1-3; any hash over the same key
    // material keeps the "one hash of the name string per lookup" cost shape.
    size_t h = std::hash<std::string>()(d.name);
    h ^= (size_t)d.datatype.kind * 0x9E3779B97F4A7C15ull + (size_t)d.datatype.size_param;
    return h;
  }
};

// PointAttributeMember — point_layout.rs:353-431
struct AttributeMember {
  AttributeDef def;
  uint64_t offset = 0;
  uint64_t size = 0;
  bool operator==(const AttributeMember& o) const { return def == o.def && offset == o.offset && size == o.size; }
};

enum class FieldAlignmentKind { Default, Packed };
struct FieldAlignment {
  FieldAlignmentKind kind = FieldAlignmentKind::Default;
  uint64_t max_alignment = 0;
  static FieldAlignment Default() { return {FieldAlignmentKind::Default, 0}; }
  static FieldAlignment Packed(uint64_t n) { return {FieldAlignmentKind::Packed, n}; }
};

// std::alloc::Layout::from_size_align preconditions (align non-zero power of two)
inline void check_rust_layout(uint64_t size, uint64_t align) {
  if (align == 0 || (align & (align - 1)) != 0 || size > (uint64_t)INT64_MAX - (align - 1))
    throw Panic(ERR_INVALID_LAYOUT, "Could not create memory layout for PointLayout");
}

// PointLayout — point_layout.rs:648-997
struct PointLayout {
  std::vector<AttributeMember> attributes;
  uint64_t mem_size = 0;   // memory_layout.size()
  uint64_t mem_align = 1;  // memory_layout.align(); Default impl :1011-1023 = (0, 1)

  // :667-669 / FromIterator :1025-1033
  static PointLayout from_attributes(const std::vector<AttributeDef>& defs) {
    PointLayout l;
    for (auto& d : defs) l.add_attribute(d, FieldAlignment::Default());
    return l;
  }
  // :693-702
  static PointLayout from_attributes_packed(const std::vector<AttributeDef>& defs, uint64_t max_alignment) {
    PointLayout l;
    for (auto& d : defs) l.add_attribute(d, FieldAlignment::Packed(max_alignment));
    return l;
  }
  // :719-759
  static PointLayout from_members_and_alignment(const std::vector<AttributeMember>& members, uint64_t type_alignment) {
    for (size_t i = 0; i < members.size(); ++i)
      for (size_t j = i + 1; j < members.size(); ++j)
        if (members[i].def.name == members[j].def.name)
          throw Panic(ERR_INVALID_LAYOUT, "PointLayout::from_attributes_and_offsets: All attributes must have unique names!");
    std::vector<std::pair<uint64_t, uint64_t>> ranges;
    for (auto& m : members) ranges.push_back({m.offset, m.offset + m.size});
    std::sort(ranges.begin(), ranges.end(), [](auto& a, auto& b) { return a.first < b.first; });
    for (size_t i = 1; i < ranges.size(); ++i)
      if (ranges[i - 1].second > ranges[i].first)
        throw Panic(ERR_INVALID_LAYOUT, "PointLayout::from_attributes_and_offsets: All attributes must span non-overlapping memory regions!");
    uint64_t unaligned = 0;
    if (!members.empty()) {
      const AttributeMember* last = &members[0];
      for (auto& m : members) if (m.offset >= last->offset) last = &m;  // max_by keeps the last maximum
      unaligned = last->offset + last->size;
    }
    PointLayout l;
    l.attributes = members;
    check_rust_layout(align_to(unaligned, type_alignment), type_alignment);
    l.mem_size = align_to(unaligned, type_alignment);
    l.mem_align = type_alignment;
    return l;
  }
  // :778-822
  void add_attribute(const AttributeDef& def, FieldAlignment fa) {
    if (get_attribute_by_name(def.name))
      throw Panic(ERR_DUPLICATE_ATTRIBUTE, "Point attribute " + def.name + " is already present in this PointLayout!");
    uint64_t req = fa.kind == FieldAlignmentKind::Default ? def.datatype.min_alignment()
                                                          : std::min(fa.max_alignment, def.datatype.min_alignment());
    uint64_t offset = align_to(packed_offset_of_next_field(), req);
    uint64_t cur = mem_align;
    uint64_t new_align = fa.kind == FieldAlignmentKind::Default ? std::max(cur, def.datatype.min_alignment())
                                                                : std::min(fa.max_alignment, cur);
    attributes.push_back(AttributeMember{def, offset, def.size()});
    uint64_t end = offset + def.size();
    uint64_t new_size_unaligned = std::max(mem_size, end);
    check_rust_layout(align_to(new_size_unaligned, new_align), new_align);
    mem_size = align_to(new_size_unaligned, new_align);
    mem_align = new_align;
  }
  // :834-838
  bool has_attribute_with_name(const std::string& n) const { return get_attribute_by_name(n) != nullptr; }
  // :861-866 / :882-890 (name AND datatype)
  const AttributeMember* get_attribute(const AttributeDef& d) const {
    for (auto& a : attributes) if (a.def.name == d.name && a.def.datatype == d.datatype) return &a;
    return nullptr;
  }
  bool has_attribute(const AttributeDef& d) const { return get_attribute(d) != nullptr; }
  // :892-896
  const AttributeMember* get_attribute_by_name(const std::string& n) const {
    for (auto& a : attributes) if (a.def.name == n) return &a;
    return nullptr;
  }
  uint64_t size_of_point_entry() const { return mem_size; }  // :928-931
  // derive(PartialEq) :646 — attribute list AND memory layout (size, align)
  bool operator==(const PointLayout& o) const {
    return attributes == o.attributes && mem_size == o.mem_size && mem_align == o.mem_align;
  }
  bool operator!=(const PointLayout& o) const { return !(*this == o); }
  // :985-996
  uint64_t packed_offset_of_next_field() const {
    if (attributes.empty()) return 0;
    return attributes.back().offset + attributes.back().size;
  }
};

// ----------------------------------------------------------------------------------
// Buffers — pasture-core/src/containers/point_buffer.rs
// ----------------------------------------------------------------------------------
struct Range { size_t start = 0, end = 0; size_t len() const { return end - start; } };

struct InterleavedBuffer;
struct ColumnarBuffer;

// BorrowedBuffer / BorrowedMutBuffer / OwningBuffer — point_buffer.rs:17-266 (merged; the oracle's
// buffers are all owning)
struct Buffer {
  virtual ~Buffer() = default;
  virtual size_t len() const = 0;
  virtual const PointLayout& point_layout() const = 0;
  virtual void get_attribute_unchecked(const AttributeMember& m, size_t index, uint8_t* out) const = 0;
  virtual void set_attribute(const AttributeDef& d, size_t index, const uint8_t* data) = 0;
  virtual void resize(size_t count) = 0;  // zero-fills (:263, :831-835, :1354-1360)
  virtual InterleavedBuffer* as_interleaved() { return nullptr; }
  virtual ColumnarBuffer* as_columnar() { return nullptr; }
  const InterleavedBuffer* as_interleaved() const { return const_cast<Buffer*>(this)->as_interleaved(); }
  const ColumnarBuffer* as_columnar() const { return const_cast<Buffer*>(this)->as_columnar(); }
};

// RawAttributeView(Mut) — raw_attribute_view.rs:10-134
struct RawAttributeView {
  uint8_t* data; size_t data_len; size_t offset, stride, size_of_attribute;
  uint8_t* at(size_t index) const {
    size_t start = offset + stride * index;
    size_t end = start + size_of_attribute;
    if (end > data_len) throw Panic(ERR_RANGE, "range end index out of range for slice");
    return data + start;
  }
};

struct InterleavedBuffer : Buffer {
  virtual uint8_t* get_point_range_mut(Range r) = 0;  // :555-557
  const uint8_t* get_point_range_ref(Range r) const { return const_cast<InterleavedBuffer*>(this)->get_point_range_mut(r); }
  // :531-536 / raw_attribute_view.rs:19-33
  RawAttributeView view_raw_attribute(const AttributeMember& m) {
    size_t stride = point_layout().size_of_point_entry();
    return RawAttributeView{get_point_range_mut({0, len()}), len() * stride, (size_t)m.offset, stride, (size_t)m.size};
  }
};
struct ColumnarBuffer : Buffer {
  virtual uint8_t* get_attribute_range_mut(const AttributeDef& d, Range r) = 0;  // :636-642
  const uint8_t* get_attribute_range_ref(const AttributeDef& d, Range r) const {
    return const_cast<ColumnarBuffer*>(this)->get_attribute_range_mut(d, r);
  }
  // :218-223 / :1340-1347
  void set_attribute_range(const AttributeDef& d, Range r, const uint8_t* data) {
    std::memcpy(get_attribute_range_mut(d, r), data, r.len() * d.size());
  }
  // raw_attribute_view.rs:35-49
  RawAttributeView view_raw_attribute(const AttributeMember& m) {
    return RawAttributeView{get_attribute_range_mut(m.def, {0, len()}), len() * (size_t)m.def.size(), 0,
                            (size_t)m.def.size(), (size_t)m.def.size()};
  }
};

// VectorBuffer — point_buffer.rs:659-945 (AoS Vec<u8>)
struct VectorBuffer : InterleavedBuffer {
  std::vector<uint8_t> storage;
  PointLayout layout;
  size_t length = 0;
  explicit VectorBuffer(PointLayout l) : layout(std::move(l)) {}  // new_from_layout :699-707
  size_t len() const override { return length; }
  const PointLayout& point_layout() const override { return layout; }
  // :731-739
  void get_attribute_unchecked(const AttributeMember& m, size_t index, uint8_t* out) const override {
    size_t start = index * layout.size_of_point_entry() + m.offset;
    if (start + m.size > storage.size()) throw Panic(ERR_RANGE, "range end index out of range for slice");
    std::memcpy(out, storage.data() + start, m.size);
  }
  // :756-770
  void set_attribute(const AttributeDef& d, size_t index, const uint8_t* data) override {
    const AttributeMember* m = layout.get_attribute(d);
    if (!m) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of this buffer");
    size_t start = index * layout.size_of_point_entry() + m->offset;
    if (start + m->size > storage.size()) throw Panic(ERR_RANGE, "range end index out of range for slice");
    std::memcpy(storage.data() + start, data, m->size);
  }
  void resize(size_t count) override {  // :831-835
    storage.resize(count * layout.size_of_point_entry(), 0);
    length = count;
  }
  uint8_t* get_point_range_mut(Range r) override {  // :873-880
    size_t s = layout.size_of_point_entry();
    if (r.end * s > storage.size() || r.start > r.end) throw Panic(ERR_RANGE, "range end index out of range for slice");
    return storage.data() + r.start * s;
  }
  InterleavedBuffer* as_interleaved() override { return this; }  // :741-743
};

// HashMapBuffer — point_buffer.rs:1031-1474 (SoA HashMap<PointAttributeDefinition, Vec<u8>>)
struct HashMapBuffer : ColumnarBuffer {
  std::unordered_map<AttributeDef, std::vector<uint8_t>, AttributeDefHash> attributes_storage;
  PointLayout layout;
  size_t length = 0;
  explicit HashMapBuffer(PointLayout l) : layout(std::move(l)) {  // new_from_layout :1155-1167
    for (auto& a : layout.attributes) attributes_storage.emplace(a.def, std::vector<uint8_t>());
  }
  size_t len() const override { return length; }
  const PointLayout& point_layout() const override { return layout; }
  // :1222-1235 — one hash lookup PER POINT
  void get_attribute_unchecked(const AttributeMember& m, size_t index, uint8_t* out) const override {
    auto it = attributes_storage.find(m.def);
    if (it == attributes_storage.end()) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of this buffer");
    size_t s = m.def.size();
    if ((index + 1) * s > it->second.size()) throw Panic(ERR_RANGE, "range end index out of range for slice");
    std::memcpy(out, it->second.data() + index * s, s);
  }
  // :1263-1276
  void set_attribute(const AttributeDef& d, size_t index, const uint8_t* data) override {
    auto it = attributes_storage.find(d);
    if (it == attributes_storage.end()) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of this buffer");
    size_t s = d.size();
    if ((index + 1) * s > it->second.size()) throw Panic(ERR_RANGE, "range end index out of range for slice");
    std::memcpy(it->second.data() + index * s, data, s);
  }
  void resize(size_t count) override {  // :1354-1360
    for (auto& kv : attributes_storage) kv.second.resize(count * kv.first.size(), 0);
    length = count;
  }
  uint8_t* get_attribute_range_mut(const AttributeDef& d, Range r) override {  // :1425-1439
    auto it = attributes_storage.find(d);
    if (it == attributes_storage.end()) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of this buffer");
    size_t s = d.size();
    if (r.end * s > it->second.size() || r.start > r.end) throw Panic(ERR_RANGE, "range end index out of range for slice");
    return it->second.data() + r.start * s;
  }
  ColumnarBuffer* as_columnar() override { return this; }  // :1237-1239
};

// BufferSlice / BufferSliceMut — pasture-core/src/containers/slice.rs:76-330: a view of `point_range` of another buffer with the same memory
// layout capabilities (an interleaved buffer's slice is interleaved, a columnar buffer's columnar).  Local indices are checked against
// the slice's own length and shifted by point_range.start (get_and_check_global_point_index :52-55, _range :64-75).
inline size_t slice_global_index(size_t local_index, const Range& point_range) {
  if (!(local_index < point_range.len())) throw Panic(ERR_RANGE, "assertion failed: local_index < point_range.len()");
  return local_index + point_range.start;
}
inline Range slice_global_range(const Range& local_range, const Range& point_range) {
  if (!(local_range.end <= point_range.len())) throw Panic(ERR_RANGE, "assertion failed: local_range.end <= point_range.len()");
  if (local_range.start >= local_range.end) return Range{point_range.end, point_range.end};
  return Range{local_range.start + point_range.start, local_range.end + point_range.start};
}
struct InterleavedSlice : InterleavedBuffer {
  InterleavedBuffer* buffer;
  Range point_range;
  InterleavedSlice(InterleavedBuffer* b, Range r) : buffer(b), point_range(r) {}
  size_t len() const override { return point_range.end - point_range.start; }
  const PointLayout& point_layout() const override { return buffer->point_layout(); }
  void get_attribute_unchecked(const AttributeMember& m, size_t index, uint8_t* out) const override {
    buffer->get_attribute_unchecked(m, slice_global_index(index, point_range), out);
  }
  void set_attribute(const AttributeDef& d, size_t index, const uint8_t* data) override {
    buffer->set_attribute(d, slice_global_index(index, point_range), data);
  }
  void resize(size_t) override { throw Panic(ERR_UNSUPPORTED, "a buffer slice is not an OwningBuffer: it cannot be resized"); }
  uint8_t* get_point_range_mut(Range r) override { return buffer->get_point_range_mut(slice_global_range(r, point_range)); }
  InterleavedBuffer* as_interleaved() override { return this; }
};
struct ColumnarSlice : ColumnarBuffer {
  ColumnarBuffer* buffer;
  Range point_range;
  ColumnarSlice(ColumnarBuffer* b, Range r) : buffer(b), point_range(r) {}
  size_t len() const override { return point_range.end - point_range.start; }
  const PointLayout& point_layout() const override { return buffer->point_layout(); }
  void get_attribute_unchecked(const AttributeMember& m, size_t index, uint8_t* out) const override {
    buffer->get_attribute_unchecked(m, slice_global_index(index, point_range), out);
  }
  void set_attribute(const AttributeDef& d, size_t index, const uint8_t* data) override {
    buffer->set_attribute(d, slice_global_index(index, point_range), data);
  }
  void resize(size_t) override { throw Panic(ERR_UNSUPPORTED, "a buffer slice is not an OwningBuffer: it cannot be resized"); }
  uint8_t* get_attribute_range_mut(const AttributeDef& d, Range r) override {
    return buffer->get_attribute_range_mut(d, slice_global_range(r, point_range));
  }
  ColumnarBuffer* as_columnar() override { return this; }
};
// SliceBuffer::slice :16-29 for VectorBuffer / HashMapBuffer / slices of them ("May panic if `range` is out of bounds": the index
// assertions above fire at the first access; here the range is checked when the slice is taken, like a Rust slice index)
inline std::unique_ptr<Buffer> make_slice(Buffer& parent, Range r) {
  if (r.start > r.end || r.end > parent.len())
    throw Panic(ERR_RANGE, "range end index " + std::to_string(r.end) + " out of range for buffer of length " + std::to_string(parent.len()));
  if (InterleavedBuffer* i = parent.as_interleaved()) return std::make_unique<InterleavedSlice>(i, r);
  if (ColumnarBuffer* c = parent.as_columnar()) return std::make_unique<ColumnarSlice>(c, r);
  throw Panic(ERR_UNSUPPORTED, "buffer is neither interleaved nor columnar");
}

// ----------------------------------------------------------------------------------
// Rust `as` semantics — the language definition of numeric casts, which
// attribute_conversion.rs:310-343 applies via num_traits::AsPrimitive.
// ----------------------------------------------------------------------------------
template <typename To, typename From>
inline To rust_as(From v) {
  if constexpr (std::is_floating_point<From>::value && std::is_integral<To>::value) {
    // float -> int: truncate toward zero, saturate at To::MIN/MAX, NaN -> 0
    if (v != v) return (To)0;
    constexpr int digits = std::numeric_limits<To>::digits;  // value bits (excl. sign)
    const From hi = (From)std::ldexp(1.0, digits);            // 2^digits, exact in f32 and f64
    if (v >= hi) return std::numeric_limits<To>::max();
    if constexpr (std::is_signed<To>::value) {
      if (v < -hi) return std::numeric_limits<To>::min();     // (-hi-1, -hi) truncates to MIN anyway
      if (v <= -hi) return std::numeric_limits<To>::min();
    } else {
      if (v <= (From)-1) return (To)0;
      if (v < (From)0) return (To)0;                          // (-1, 0) truncates to 0
    }
    return (To)v;  // in range: defined behaviour, truncation toward zero
  } else {
    // int->int: two's complement truncate / extend; int->float: round-to-nearest-even;
    // f64->f32: RNE, overflow -> inf, NaN preserved; f32->f64 exact
    return static_cast<To>(v);
  }
}

// AttributeConversionFn — attribute_conversion.rs:112
using AttributeConversionFn = void (*)(const uint8_t* from, uint8_t* to);

// convert_scalar_using_as — attribute_conversion.rs:310-321 (unaligned read, `as`, unaligned write)
template <typename From, typename To>
void convert_scalar_using_as(const uint8_t* from, uint8_t* to) {
  From f; std::memcpy(&f, from, sizeof(From));
  To t = rust_as<To, From>(f);
  std::memcpy(to, &t, sizeof(To));
}
// convert_vec3_using_as — attribute_conversion.rs:332-343
template <typename From, typename To>
void convert_vec3_using_as(const uint8_t* from, uint8_t* to) {
  From f[3]; std::memcpy(f, from, sizeof(f));
  To t[3] = {rust_as<To, From>(f[0]), rust_as<To, From>(f[1]), rust_as<To, From>(f[2])};
  std::memcpy(to, t, sizeof(t));
}

// get_generic_converter — attribute_conversion.rs:184-271.  Returns nullptr for an unlisted pair
// (the reference panics "Invalid conversion X -> Y" :267-269; callers below throw).
inline AttributeConversionFn lookup_generic_converter(Kind from, Kind to) {
  static const std::map<std::pair<Kind, Kind>, AttributeConversionFn> table = [] {
    std::map<std::pair<Kind, Kind>, AttributeConversionFn> m;
#define ORC_SCALAR(PF, PT, KF, KT)                                  \
    m[{KF, KT}] = convert_scalar_using_as<PF, PT>;                  \
    m[{KT, KF}] = convert_scalar_using_as<PT, PF>;
#define ORC_VEC3(PF, PT, KF, KT)                                    \
    m[{KF, KT}] = convert_vec3_using_as<PF, PT>;                    \
    m[{KT, KF}] = convert_vec3_using_as<PT, PF>;
    // :194-246 (45 unordered scalar pairs)
    ORC_SCALAR(uint8_t, uint16_t, U8, U16) ORC_SCALAR(uint8_t, uint32_t, U8, U32) ORC_SCALAR(uint8_t, uint64_t, U8, U64)
    ORC_SCALAR(uint8_t, int8_t, U8, I8) ORC_SCALAR(uint8_t, int16_t, U8, I16) ORC_SCALAR(uint8_t, int32_t, U8, I32)
    ORC_SCALAR(uint8_t, int64_t, U8, I64) ORC_SCALAR(uint8_t, float, U8, F32) ORC_SCALAR(uint8_t, double, U8, F64)
    ORC_SCALAR(uint16_t, uint32_t, U16, U32) ORC_SCALAR(uint16_t, uint64_t, U16, U64) ORC_SCALAR(uint16_t, int8_t, U16, I8)
    ORC_SCALAR(uint16_t, int16_t, U16, I16) ORC_SCALAR(uint16_t, int32_t, U16, I32) ORC_SCALAR(uint16_t, int64_t, U16, I64)
    ORC_SCALAR(uint16_t, float, U16, F32) ORC_SCALAR(uint16_t, double, U16, F64)
    ORC_SCALAR(uint32_t, uint64_t, U32, U64) ORC_SCALAR(uint32_t, int8_t, U32, I8) ORC_SCALAR(uint32_t, int16_t, U32, I16)
    ORC_SCALAR(uint32_t, int32_t, U32, I32) ORC_SCALAR(uint32_t, int64_t, U32, I64) ORC_SCALAR(uint32_t, float, U32, F32)
    ORC_SCALAR(uint32_t, double, U32, F64)
    ORC_SCALAR(uint64_t, int8_t, U64, I8) ORC_SCALAR(uint64_t, int16_t, U64, I16) ORC_SCALAR(uint64_t, int32_t, U64, I32)
    ORC_SCALAR(uint64_t, int64_t, U64, I64) ORC_SCALAR(uint64_t, float, U64, F32) ORC_SCALAR(uint64_t, double, U64, F64)
    ORC_SCALAR(int8_t, int16_t, I8, I16) ORC_SCALAR(int8_t, int32_t, I8, I32) ORC_SCALAR(int8_t, int64_t, I8, I64)
    ORC_SCALAR(int8_t, float, I8, F32) ORC_SCALAR(int8_t, double, I8, F64)
    ORC_SCALAR(int16_t, int32_t, I16, I32) ORC_SCALAR(int16_t, int64_t, I16, I64) ORC_SCALAR(int16_t, float, I16, F32)
    ORC_SCALAR(int16_t, double, I16, F64)
    ORC_SCALAR(int32_t, int64_t, I32, I64) ORC_SCALAR(int32_t, float, I32, F32) ORC_SCALAR(int32_t, double, I32, F64)
    ORC_SCALAR(int64_t, float, I64, F32) ORC_SCALAR(int64_t, double, I64, F64)
    ORC_SCALAR(float, double, F32, F64)
    // :248-260 (10 unordered Vec3 pairs)
    ORC_VEC3(float, double, Vec3f32, Vec3f64)
    ORC_VEC3(uint8_t, uint16_t, Vec3u8, Vec3u16) ORC_VEC3(uint8_t, int32_t, Vec3u8, Vec3i32)
    ORC_VEC3(uint8_t, float, Vec3u8, Vec3f32) ORC_VEC3(uint8_t, double, Vec3u8, Vec3f64)
    ORC_VEC3(uint16_t, int32_t, Vec3u16, Vec3i32) ORC_VEC3(uint16_t, float, Vec3u16, Vec3f32)
    ORC_VEC3(uint16_t, double, Vec3u16, Vec3f64)
    ORC_VEC3(int32_t, float, Vec3i32, Vec3f32) ORC_VEC3(int32_t, double, Vec3i32, Vec3f64)
#undef ORC_SCALAR
#undef ORC_VEC3
    return m;
  }();
  auto it = table.find({from, to});
  return it == table.end() ? nullptr : it->second;
}
inline AttributeConversionFn get_generic_converter(const DataType& from, const DataType& to) {
  AttributeConversionFn f = lookup_generic_converter(from.kind, to.kind);
  if (!f || from.kind >= Vec4u8 || to.kind >= Vec4u8)
    throw Panic(ERR_INVALID_CONVERSION, "Invalid conversion " + from.display() + " -> " + to.display());
  return f;
}
// get_converter_for_attributes — attribute_conversion.rs:122-132: None for equal datatypes (the caller copies or, in RawPointConverter,
// SKIPS the attribute), the generic converter otherwise; panics for an impossible pair.
inline AttributeConversionFn get_converter_for_attributes(const AttributeDef& from, const AttributeDef& to) {
  if (from.name != to.name) throw Panic(ERR_INVALID_ARGUMENT, "assertion `left == right` failed: from_attribute.name() == to_attribute.name()");
  if (from.datatype == to.datatype) return nullptr;
  return get_generic_converter(from.datatype, to.datatype);
}

// RawAttributeConverter + RawPointConverter — attribute_conversion.rs:21-109.  Point-major: converts ONE interleaved point; only
// attributes present in both layouts (matched by name, in the order of `from_layout`) whose datatypes DIFFER get a converter --
// same-datatype attributes are skipped, not copied (:73-90: filter_map over an Option that is None for equal datatypes).
struct RawAttributeConverter {
  AttributeConversionFn conversion_fn;
  uint64_t source_offset, source_size, target_offset, target_size;
  void convert(const uint8_t* source_point, uint8_t* target_point) const { conversion_fn(source_point + source_offset, target_point + target_offset); }  // :45-58
};
struct RawPointConverter {
  std::vector<RawAttributeConverter> attribute_converters;
  static RawPointConverter from_to(const PointLayout& from_layout, const PointLayout& to_layout) {  // :69-96
    RawPointConverter c;
    for (const AttributeMember& from_attribute : from_layout.attributes) {
      const AttributeMember* to_attribute = to_layout.get_attribute_by_name(from_attribute.def.name);
      if (!to_attribute) continue;
      AttributeConversionFn fn = get_converter_for_attributes(from_attribute.def, to_attribute->def);
      if (fn) c.attribute_converters.push_back({fn, from_attribute.offset, from_attribute.def.datatype.size(), to_attribute->offset, to_attribute->def.datatype.size()});
    }
    return c;
  }
  void convert(const uint8_t* source_point, uint8_t* target_point) const {  // :104-108
    for (const RawAttributeConverter& a : attribute_converters) a.convert(source_point, target_point);
  }
};

// convert_unit — attribute_conversion.rs:297-299 (only used through AttributeViewConverting)

// ----------------------------------------------------------------------------------
// Transformations.  The reference takes arbitrary `Fn(T)->T` closures (buffer_conversion.rs:14-31).
// The oracle builds closures from the SAME closed descriptor set the device path accepts; each
// closure body restates one reference call site.
// ----------------------------------------------------------------------------------
enum TransformKind : uint32_t {
  XF_NONE = 0,
  XF_AFFINE = 1,    // f64 / Vec3f64: (p*scale)+offset, two roundings (pasture-io/src/las/raw_readers.rs:42-48);
                    // f32 / Vec3f32: ((p as f64*scale)+offset) as f32 (raw_readers.rs:49-55);
                    // scale=1 gives add_scalar (buffer_conversion.rs:780-782) exactly
  XF_BITFIELD = 2,  // unsigned ints: (v >> shift) & mask (raw_readers.rs:61-164)
};
struct TransformDesc {
  uint32_t kind = XF_NONE;
  DataType datatype;      // the closure's T (T::data_type())
  double scale[3] = {1, 1, 1};
  double offset[3] = {0, 0, 0};
  uint32_t shift = 0;
  uint64_t mask = ~0ull;
};
using AttributeTransformFn = std::function<void(uint8_t*)>;  // buffer_conversion.rs:14

// to_untyped_transform_fn — buffer_conversion.rs:17-31 (read_unaligned, f(T), write_unaligned)
inline AttributeTransformFn make_transform_fn(const TransformDesc& d) {
  const TransformDesc t = d;
  auto no_fma_affine = [](double p, double s, double o) {
    volatile double m = p * s;  // two roundings: the Rust closure is `(pos.x * scale) + offset`, never fused
    return m + o;
  };
  if (t.kind == XF_AFFINE) {
    switch (t.datatype.kind) {
      case Vec3f64:
        return [t, no_fma_affine](uint8_t* mem) {
          double v[3]; std::memcpy(v, mem, 24);
          for (int c = 0; c < 3; ++c) v[c] = no_fma_affine(v[c], t.scale[c], t.offset[c]);
          std::memcpy(mem, v, 24);
        };
      case F64:
        return [t, no_fma_affine](uint8_t* mem) {
          double v; std::memcpy(&v, mem, 8);
          v = no_fma_affine(v, t.scale[0], t.offset[0]);
          std::memcpy(mem, &v, 8);
        };
      case Vec3f32:
        return [t, no_fma_affine](uint8_t* mem) {
          float v[3]; std::memcpy(v, mem, 12);
          for (int c = 0; c < 3; ++c) v[c] = (float)no_fma_affine((double)v[c], t.scale[c], t.offset[c]);
          std::memcpy(mem, v, 12);
        };
      case F32:
        return [t, no_fma_affine](uint8_t* mem) {
          float v; std::memcpy(&v, mem, 4);
          v = (float)no_fma_affine((double)v, t.scale[0], t.offset[0]);
          std::memcpy(mem, &v, 4);
        };
      default: break;
    }
  } else if (t.kind == XF_BITFIELD) {
    switch (t.datatype.kind) {
      case U8: return [t](uint8_t* mem) { uint8_t v = *mem; v = (uint8_t)((v >> t.shift) & t.mask); *mem = v; };
      case U16: return [t](uint8_t* mem) { uint16_t v; std::memcpy(&v, mem, 2); v = (uint16_t)((v >> t.shift) & t.mask); std::memcpy(mem, &v, 2); };
      case U32: return [t](uint8_t* mem) { uint32_t v; std::memcpy(&v, mem, 4); v = (uint32_t)((v >> t.shift) & t.mask); std::memcpy(mem, &v, 4); };
      case U64: return [t](uint8_t* mem) { uint64_t v; std::memcpy(&v, mem, 8); v = (v >> t.shift) & t.mask; std::memcpy(mem, &v, 8); };
      default: break;
    }
  }
  throw Panic(ERR_UNSUPPORTED_TRANSFORM, "Unsupported transformation descriptor for datatype " + t.datatype.display());
}

// ----------------------------------------------------------------------------------
// BufferLayoutConverter — layout/conversion/buffer_conversion.rs:98-663
// ----------------------------------------------------------------------------------
struct Transformation {  // :33-36
  AttributeTransformFn func;
  bool apply_to_source_attribute;
};
struct AttributeMapping {  // :41-55
  AttributeMember target_attribute;
  AttributeMember source_attribute;
  AttributeConversionFn converter = nullptr;  // Option<AttributeConversionFn>
  std::optional<Transformation> transformation;
  size_t required_buffer_size() const { return (size_t)std::max(source_attribute.size, target_attribute.size); }
};

struct BufferLayoutConverter {
  PointLayout from_layout, to_layout;
  std::vector<AttributeMapping> mappings;

  // for_layouts :112-123
  static BufferLayoutConverter for_layouts(const PointLayout& from, const PointLayout& to) {
    BufferLayoutConverter c{from, to, {}};
    for (auto& to_attr : to.attributes) {
      const AttributeMember* from_attr = from.get_attribute_by_name(to_attr.def.name);
      if (!from_attr)
        throw Panic(ERR_MISSING_ATTRIBUTE,
                    "Attribute not found in `from_layout`! When calling `BufferLayoutConverter::for_layouts`, the source "
                    "PointLayout must contain all attributes from the target PointLayout.");
      c.mappings.push_back(make_default_mapping(*from_attr, to_attr));
    }
    return c;
  }
  // for_layouts_with_default :126-143
  static BufferLayoutConverter for_layouts_with_default(const PointLayout& from, const PointLayout& to) {
    BufferLayoutConverter c{from, to, {}};
    for (auto& to_attr : to.attributes) {
      const AttributeMember* from_attr = from.get_attribute_by_name(to_attr.def.name);
      if (from_attr) c.mappings.push_back(make_default_mapping(*from_attr, to_attr));
    }
    return c;
  }
  // set_custom_mapping :156-183
  void set_custom_mapping(const AttributeDef& from_attribute, const AttributeDef& to_attribute) {
    const AttributeMember* fm = from_layout.get_attribute(from_attribute);
    if (!fm) throw Panic(ERR_MISSING_ATTRIBUTE, "from_attribute not found in source PointLayout");
    const AttributeMember* tm = to_layout.get_attribute(to_attribute);
    if (!tm) throw Panic(ERR_MISSING_ATTRIBUTE, "to_attribute not found in target PointLayout");
    AttributeMapping m = make_default_mapping(*fm, *tm);
    for (auto& prev : mappings)
      if (prev.target_attribute.def == to_attribute) { prev = std::move(m); return; }
    mappings.push_back(std::move(m));
  }
  // set_custom_mapping_with_transformation :194-234
  void set_custom_mapping_with_transformation(const AttributeDef& from_attribute, const AttributeDef& to_attribute,
                                              const TransformDesc& xf, bool apply_to_source_attribute) {
    const AttributeMember* fm = from_layout.get_attribute(from_attribute);
    if (!fm) throw Panic(ERR_MISSING_ATTRIBUTE, "from_attribute not found in source PointLayout");
    const AttributeMember* tm = to_layout.get_attribute(to_attribute);
    if (!tm) throw Panic(ERR_MISSING_ATTRIBUTE, "to_attribute not found in target PointLayout");
    const DataType& expect = apply_to_source_attribute ? fm->def.datatype : tm->def.datatype;  // :209-213
    if (xf.datatype != expect)
      throw Panic(ERR_TRANSFORM_TYPE_MISMATCH, "assertion failed: T::data_type() == " + expect.display() + " (got " + xf.datatype.display() + ")");
    AttributeMapping m = make_default_mapping(*fm, *tm);  // make_transformed_mapping :404-416
    m.transformation = Transformation{make_transform_fn(xf), apply_to_source_attribute};
    for (auto& prev : mappings)
      if (prev.target_attribute.def == to_attribute) { prev = std::move(m); return; }
    mappings.push_back(std::move(m));
  }
  // make_default_mapping :368-396
  static AttributeMapping make_default_mapping(const AttributeMember& from, const AttributeMember& to) {
    AttributeMapping m;
    m.target_attribute = to;
    m.source_attribute = from;
    if (!(from.def.datatype == to.def.datatype)) m.converter = get_generic_converter(from.def.datatype, to.def.datatype);
    return m;
  }

  // convert :242-259 (new_from_layout + resize (zero fill) + convert_into)
  template <typename OutBuffer>
  std::unique_ptr<OutBuffer> convert(Buffer& source) const {
    auto target = std::make_unique<OutBuffer>(to_layout);
    target->resize(source.len());
    convert_into(source, *target);
    return target;
  }
  // convert_into :268-283
  void convert_into(Buffer& source, Buffer& target) const {
    Range r{0, source.len()};
    convert_into_range(source, r, target, r);
  }
  // convert_into_range :292-359
  void convert_into_range(Buffer& source, Range source_range, Buffer& target, Range target_range) const {
    if (source.point_layout() != from_layout) throw Panic(ERR_LAYOUT_MISMATCH, "assertion failed: source_buffer.point_layout() == self.from_layout");
    if (target.point_layout() != to_layout) throw Panic(ERR_LAYOUT_MISMATCH, "assertion failed: target_buffer.point_layout() == self.to_layout");
    if (source_range.len() != target_range.len() || source_range.end < source_range.start || target_range.end < target_range.start)
      throw Panic(ERR_RANGE, "assertion failed: source_range.len() == target_range.len()");
    if (source_range.end > source.len()) throw Panic(ERR_RANGE, "assertion failed: source_range.end <= source_buffer.len()");
    if (target_range.end > target.len()) throw Panic(ERR_RANGE, "assertion failed: target_range.end <= target_buffer.len()");
    if (mappings.empty()) return;  // max() of an empty iterator is None :308-313
    size_t max_attribute_size = 0;
    for (auto& m : mappings) max_attribute_size = std::max(max_attribute_size, m.required_buffer_size());
    ColumnarBuffer* sc = source.as_columnar();
    ColumnarBuffer* tc = target.as_columnar();
    if (sc && tc) convert_columnar_to_columnar(*sc, source_range, *tc, target_range);
    else if (sc) {
      InterleavedBuffer* ti = target.as_interleaved();
      if (!ti) throw Panic(ERR_INVALID_ARGUMENT, "Target buffer must either be an interleaved or columnar buffer");
      convert_columnar_to_interleaved(*sc, source_range, *ti, target_range);
    } else if (tc) {
      InterleavedBuffer* si = source.as_interleaved();
      if (!si) throw Panic(ERR_INVALID_ARGUMENT, "Source buffer must either be an interleaved or columnar buffer");
      convert_interleaved_to_columnar(*si, source_range, *tc, target_range, max_attribute_size);
    } else {
      InterleavedBuffer* si = source.as_interleaved();
      InterleavedBuffer* ti = target.as_interleaved();
      if (!si || !ti) throw Panic(ERR_INVALID_ARGUMENT, "buffers must either be interleaved or columnar");
      convert_interleaved_to_interleaved(*si, source_range, *ti, target_range, max_attribute_size);
    }
  }

  // :418-487
  void convert_columnar_to_columnar(ColumnarBuffer& src, Range sr, ColumnarBuffer& dst, Range tr) const {
    for (auto& mapping : mappings) {
      const uint8_t* source_data = src.get_attribute_range_ref(mapping.source_attribute.def, sr);
      if (mapping.converter) {
        uint8_t* target_data = dst.get_attribute_range_mut(mapping.target_attribute.def, tr);
        size_t ss = mapping.source_attribute.size, ts = mapping.target_attribute.size;
        std::vector<uint8_t> tmp(ss, 0);
        for (size_t i = 0; i < sr.len(); ++i) {
          const uint8_t* s = source_data + i * ss;
          uint8_t* t = target_data + i * ts;
          if (mapping.transformation) {
            if (mapping.transformation->apply_to_source_attribute) {
              std::memcpy(tmp.data(), s, ss);
              mapping.transformation->func(tmp.data());
              mapping.converter(tmp.data(), t);
            } else {
              mapping.converter(s, t);
              mapping.transformation->func(t);
            }
          } else {
            mapping.converter(s, t);
          }
        }
      } else {
        dst.set_attribute_range(mapping.target_attribute.def, tr, source_data);  // bulk memcpy :465-469
        if (mapping.transformation) {                                           // in-place sweep :471-484
          uint8_t* target_data = dst.get_attribute_range_mut(mapping.target_attribute.def, tr);
          size_t ts = mapping.target_attribute.size;
          for (size_t i = 0; i < tr.len(); ++i) mapping.transformation->func(target_data + i * ts);
        }
      }
    }
  }
  // :489-544
  void convert_columnar_to_interleaved(ColumnarBuffer& src, Range sr, InterleavedBuffer& dst, Range tr) const {
    for (auto& mapping : mappings) {
      const uint8_t* source_data = src.get_attribute_range_ref(mapping.source_attribute.def, sr);
      RawAttributeView target_view = dst.view_raw_attribute(mapping.target_attribute);
      size_t ss = mapping.source_attribute.size;
      if (mapping.converter) {
        std::vector<uint8_t> tmp(ss, 0);
        for (size_t i = 0; i < sr.len(); ++i) {
          const uint8_t* s = source_data + i * ss;
          uint8_t* t = target_view.at(i + tr.start);
          if (mapping.transformation) {
            if (mapping.transformation->apply_to_source_attribute) {
              std::memcpy(tmp.data(), s, ss);
              mapping.transformation->func(tmp.data());
              mapping.converter(tmp.data(), t);
            } else {
              mapping.converter(s, t);
              mapping.transformation->func(t);
            }
          } else {
            mapping.converter(s, t);
          }
        }
      } else {
        for (size_t i = 0; i < sr.len(); ++i) {
          uint8_t* t = target_view.at(i + tr.start);
          std::memcpy(t, source_data + i * ss, ss);
          if (mapping.transformation) mapping.transformation->func(t);
        }
      }
    }
  }
  // :546-604
  void convert_interleaved_to_columnar(InterleavedBuffer& src, Range sr, ColumnarBuffer& dst, Range tr, size_t max_attribute_size) const {
    std::vector<uint8_t> buffer(max_attribute_size, 0);
    for (auto& mapping : mappings) {
      RawAttributeView source_view = src.view_raw_attribute(mapping.source_attribute);
      uint8_t* target_range = dst.get_attribute_range_mut(mapping.target_attribute.def, tr);
      size_t ts = mapping.target_attribute.size;
      size_t ss = mapping.source_attribute.size;
      for (size_t i = 0; i < tr.len(); ++i) {
        uint8_t* t = target_range + i * ts;
        const uint8_t* s = source_view.at(i + sr.start);
        if (mapping.converter) {
          if (mapping.transformation) {
            if (mapping.transformation->apply_to_source_attribute) {
              std::memcpy(buffer.data(), s, ss);
              mapping.transformation->func(buffer.data());
              mapping.converter(buffer.data(), t);
            } else {
              mapping.converter(s, t);
              mapping.transformation->func(t);
            }
          } else {
            mapping.converter(s, t);
          }
        } else if (mapping.transformation) {
          std::memcpy(buffer.data(), s, ss);
          mapping.transformation->func(buffer.data());
          std::memcpy(t, buffer.data(), ss);
        } else {
          std::memcpy(t, s, ss);
        }
      }
    }
  }
  // :606-662
  void convert_interleaved_to_interleaved(InterleavedBuffer& src, Range sr, InterleavedBuffer& dst, Range tr, size_t max_attribute_size) const {
    std::vector<uint8_t> buffer(max_attribute_size, 0);
    for (auto& mapping : mappings) {
      RawAttributeView source_view = src.view_raw_attribute(mapping.source_attribute);
      RawAttributeView target_view = dst.view_raw_attribute(mapping.target_attribute);
      size_t ss = mapping.source_attribute.size;
      for (size_t k = 0; k < sr.len(); ++k) {
        const uint8_t* s = source_view.at(sr.start + k);
        uint8_t* t = target_view.at(tr.start + k);
        if (mapping.converter) {
          if (mapping.transformation) {
            if (mapping.transformation->apply_to_source_attribute) {
              std::memcpy(buffer.data(), s, ss);
              mapping.transformation->func(buffer.data());
              mapping.converter(buffer.data(), t);
            } else {
              mapping.converter(s, t);
              mapping.transformation->func(t);
            }
          } else {
            mapping.converter(s, t);
          }
        } else if (mapping.transformation) {
          std::memcpy(buffer.data(), s, ss);
          mapping.transformation->func(buffer.data());
          std::memcpy(t, buffer.data(), ss);
        } else {
          std::memcpy(t, s, ss);
        }
      }
    }
  }
};

// ----------------------------------------------------------------------------------
// AABB — pasture-core/src/math/bounds.rs:9-26
// ----------------------------------------------------------------------------------
struct AABB {
  double min[3], max[3];
  // from_min_max :21-26 (panics if min > max on any axis; NaN compares false)
  static AABB from_min_max(const double mn[3], const double mx[3]) {
    if (mn[0] > mx[0] || mn[1] > mx[1] || mn[2] > mx[2])
      throw Panic(ERR_BOUNDS_INVALID, "AABB::from_min_max: Minimum position must be <= maximum position!");
    AABB b; for (int c = 0; c < 3; ++c) { b.min[c] = mn[c]; b.max[c] = mx[c]; }
    return b;
  }
};

static const AttributeDef POSITION_3D{"Position3D", DataType::of(Vec3f64)};  // point_layout.rs:459-462

// calculate_bounds — pasture-algorithms/src/bounds.rs:11-85
inline std::optional<AABB> calculate_bounds(const Buffer& buffer) {
  if (buffer.len() == 0) return std::nullopt;                                                   // :12-14
  const AttributeMember* pos = buffer.point_layout().get_attribute_by_name(POSITION_3D.name);  // :15-21
  if (!pos) return std::nullopt;
  double pos_min[3] = {DBL_MAX, DBL_MAX, DBL_MAX};     // f64::MAX :31 / :59
  double pos_max[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};  // f64::MIN :32 / :60
  auto fold = [&](const double p[3]) {  // six strict compares :34-51 / :65-82
    if (p[0] < pos_min[0]) pos_min[0] = p[0];
    if (p[1] < pos_min[1]) pos_min[1] = p[1];
    if (p[2] < pos_min[2]) pos_min[2] = p[2];
    if (p[0] > pos_max[0]) pos_max[0] = p[0];
    if (p[1] > pos_max[1]) pos_max[1] = p[1];
    if (p[2] > pos_max[2]) pos_max[2] = p[2];
  };
  const size_t n = buffer.len();
  if (pos->def.datatype == POSITION_3D.datatype) {
    // calculate_bounds_from_default_positions :30-54; view_attribute -> get_attribute_unchecked per point
    for (size_t i = 0; i < n; ++i) {
      double p[3];
      buffer.get_attribute_unchecked(*pos, i, (uint8_t*)p);
      fold(p);
    }
  } else {
    // calculate_bounds_from_custom_positions :56-85 via AttributeViewConverting (buffer_views.rs:533-589);
    // `.ok()` turns a failed view construction into None — but get_converter panics for impossible pairs.
    AttributeConversionFn conv = get_generic_converter(pos->def.datatype, POSITION_3D.datatype);
    std::vector<uint8_t> scratch(pos->size);
    for (size_t i = 0; i < n; ++i) {
      double p[3];
      buffer.get_attribute_unchecked(*pos, i, scratch.data());
      conv(scratch.data(), (uint8_t*)p);
      fold(p);
    }
  }
  return AABB::from_min_max(pos_min, pos_max);
}

// ----------------------------------------------------------------------------------
// voxelgrid_filter — pasture-algorithms/src/voxel_grid.rs:21-689
// ----------------------------------------------------------------------------------
namespace voxel {

enum Reduce { AVG_VEC = 1, AVG_NUM = 2, MOST_COMMON = 3, MOST_COMMON_BOOL = 4, MAX_POOL = 5 };
struct Rule { const char* name; Kind kind; Reduce reduce; };
// set_all_attributes :459-689: name AND datatype must equal the builtin definition
static const Rule kRules[] = {
    {"Position3D", Vec3f64, AVG_VEC},      {"Intensity", U16, AVG_NUM},           {"ReturnNumber", U8, MOST_COMMON},
    {"NumberOfReturns", U8, MOST_COMMON},  {"ClassificationFlags", U8, MAX_POOL}, {"ScannerChannel", U8, MOST_COMMON},
    {"ScanDirectionFlag", U8, MOST_COMMON_BOOL}, {"EdgeOfFlightLine", U8, MOST_COMMON_BOOL}, {"Classification", U8, MOST_COMMON},
    {"ScanAngleRank", I8, MOST_COMMON},    {"ScanAngle", I16, MOST_COMMON},       {"UserData", U8, MOST_COMMON},
    {"PointSourceID", U16, MOST_COMMON},   {"ColorRGB", Vec3u16, AVG_VEC},        {"GpsTime", F64, MAX_POOL},
    {"NIR", U16, AVG_NUM},                 {"PointID", U64, MAX_POOL},            {"Normal", Vec3f32, AVG_VEC},
};
static const char* kWaveform[] = {"WaveformDataOffset", "WaveformPacketSize", "WaveformParameters", "WavePacketDescriptorIndex",
                                  "ReturnPointWaveformLocation"};

template <typename T> inline double scalar_as_f64(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof(T)); return rust_as<double, T>(v); }
inline double component_as_f64(Kind scalar, const uint8_t* p) {
  switch (scalar) {
    case U8: return scalar_as_f64<uint8_t>(p);   case I8: return scalar_as_f64<int8_t>(p);
    case U16: return scalar_as_f64<uint16_t>(p); case I16: return scalar_as_f64<int16_t>(p);
    case U32: return scalar_as_f64<uint32_t>(p); case I32: return scalar_as_f64<int32_t>(p);
    case U64: return scalar_as_f64<uint64_t>(p); case I64: return scalar_as_f64<int64_t>(p);
    case F32: return scalar_as_f64<float>(p);    default: return scalar_as_f64<double>(p);
  }
}
inline int64_t scalar_as_isize(Kind k, const uint8_t* p) {  // to_string().parse::<isize>() of an integer value
  switch (k) {
    case U8: { uint8_t v; std::memcpy(&v, p, 1); return v; }   case I8: { int8_t v; std::memcpy(&v, p, 1); return v; }
    case U16: { uint16_t v; std::memcpy(&v, p, 2); return v; } default: { int16_t v; std::memcpy(&v, p, 2); return v; }
  }
}

// find_leaf :21-52
inline size_t find_leaf_axis(double p, const std::vector<double>& markers) {
  size_t index = 0;
  while (!markers.empty() && markers[index] < p) index += 1;
  if (index > 0 && p - markers[index - 1] < markers[index] - p) index -= 1;  // clamp to the better fitting marker
  return index;
}
// create_markers_for_axis :55-83
inline std::vector<double> create_markers(double mn, double mx, double leafsize) {
  std::vector<double> m;
  double curr = mn;
  while (curr < mx) {
    const double next = curr + leafsize;
    if (!(next > curr)) throw Panic(ERR_INVALID_ARGUMENT, "voxelgrid_filter: leaf size does not advance the marker (the reference loops forever)");
    curr = next;
    m.push_back(curr);
  }
  return m;
}

}  // namespace voxel

// voxelgrid_filter :109-166.  Ties of centroid_most_common are broken by HashMap iteration order in the reference (random per
// process, :319-327); the restatement picks the smallest value among the most frequent ones.
inline void voxelgrid_filter(const Buffer& buffer, double leafsize_x, double leafsize_y, double leafsize_z, Buffer& filtered_buffer) {
  using namespace voxel;
  const AttributeMember* pos = buffer.point_layout().get_attribute(POSITION_3D);
  if (!pos)  // :116-122
    throw Panic(ERR_MISSING_ATTRIBUTE, "The PointBuffer does not have the attribute attributes::POSITION_3D which is needed for the creation of the voxel grid.");
  std::optional<AABB> aabb = calculate_bounds(buffer);  // :125
  if (!aabb) throw Panic(ERR_BOUNDS_INVALID, "called `Option::unwrap()` on a `None` value");
  const std::vector<double> mx = create_markers(aabb->min[0], aabb->max[0], leafsize_x), my = create_markers(aabb->min[1], aabb->max[1], leafsize_y),
                            mz = create_markers(aabb->min[2], aabb->max[2], leafsize_z);
  // :131-152: Vec<Voxel> kept sorted by pos with binary_search + insert; points pushed in index order == an ordered map
  std::map<std::tuple<size_t, size_t, size_t>, std::vector<size_t>> voxels;
  for (size_t i = 0; i < buffer.len(); ++i) {
    double p[3];
    buffer.get_attribute_unchecked(*pos, i, (uint8_t*)p);
    voxels[{find_leaf_axis(p[0], mx), find_leaf_axis(p[1], my), find_leaf_axis(p[2], mz)}].push_back(i);
  }
  const PointLayout& layout = filtered_buffer.point_layout();
  // set_all_attributes :459-476
  for (const char* w : kWaveform)
    if (layout.get_attribute_by_name(w)) throw Panic(ERR_UNSUPPORTED_ATTRIBUTE, "Waveform data currently not supported!");
  struct Plan { const AttributeMember* dst; const AttributeMember* src; const Rule* rule; };
  std::vector<Plan> plan;
  for (auto& a : layout.attributes) {
    const Rule* rule = nullptr;
    for (auto& r : kRules)
      if (a.def.name == r.name && a.def.datatype == DataType::of(r.kind)) rule = &r;
    if (!rule) throw Panic(ERR_UNSUPPORTED_ATTRIBUTE, "attribute is non-standard which is not supported currently: " + a.def.display());
    const AttributeMember* src = buffer.point_layout().get_attribute(a.def);  // view_attribute::<T>(&attributes::X)
    if (!src) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
    plan.push_back({&a, src, rule});
  }
  const size_t point_size = layout.size_of_point_entry();
  std::vector<uint8_t> centroid(point_size), value(32);
  for (auto& kv : voxels) {
    const std::vector<size_t>& points = kv.second;
    std::fill(centroid.begin(), centroid.end(), 0);  // UntypedPointBuffer::new: vec![0; size]
    for (auto& pl : plan) {
      uint8_t* out = centroid.data() + pl.dst->offset;
      const Kind kind = pl.rule->kind;
      switch (pl.rule->reduce) {
        case AVG_VEC: {  // centroid_average_vec :332-385: sequential f64 sums in point order
          const Kind sk = kind == Vec3f64 ? F64 : kind == Vec3f32 ? F32 : kind == Vec3u16 ? U16 : U8;
          const size_t cs = DataType::of(sk).size();
          double sum[3] = {0.0, 0.0, 0.0};
          for (size_t p : points) {
            buffer.get_attribute_unchecked(*pl.src, p, value.data());
            for (int c = 0; c < 3; ++c) sum[c] += component_as_f64(sk, value.data() + c * cs);
          }
          const double n = (double)points.size();
          for (int c = 0; c < 3; ++c) {
            const double avg = sum[c] / n;
            if (kind == Vec3f64) std::memcpy(out + 8 * c, &avg, 8);
            else if (kind == Vec3u16) { uint16_t v = rust_as<uint16_t, double>(avg); std::memcpy(out + 2 * c, &v, 2); }  // :626
            else { float v = rust_as<float, double>(avg); std::memcpy(out + 4 * c, &v, 4); }                              // :676
          }
          break;
        }
        case AVG_NUM: {  // centroid_average_num :389-440, `as u16` :489, :638
          double sum = 0.0;
          for (size_t p : points) { buffer.get_attribute_unchecked(*pl.src, p, value.data()); sum += component_as_f64(kind, value.data()); }
          const uint16_t v = rust_as<uint16_t, double>(sum / (double)points.size());
          std::memcpy(out, &v, 2);
          break;
        }
        case MAX_POOL: {  // centroid_max_pool :169-219: starts at 0.0, strict >
          double curr_max = 0.0;
          for (size_t p : points) {
            buffer.get_attribute_unchecked(*pl.src, p, value.data());
            const double v = component_as_f64(kind, value.data());
            if (v > curr_max) curr_max = v;
          }
          if (kind == F64) std::memcpy(out, &curr_max, 8);
          else if (kind == U64) { uint64_t v = rust_as<uint64_t, double>(curr_max); std::memcpy(out, &v, 8); }
          else { uint8_t v = rust_as<uint8_t, double>(curr_max); std::memcpy(out, &v, 1); }
          break;
        }
        case MOST_COMMON:
        case MOST_COMMON_BOOL: {  // centroid_most_common :222-329
          std::map<int64_t, size_t> counts;
          for (size_t p : points) { buffer.get_attribute_unchecked(*pl.src, p, value.data()); counts[scalar_as_isize(kind, value.data())] += 1; }
          size_t highest_count = 0;
          int64_t curr_key = 0;
          for (auto& c : counts) if (c.second > highest_count) { highest_count = c.second; curr_key = c.first; }  // ascending keys: smallest wins ties
          if (pl.rule->reduce == MOST_COMMON_BOOL) { uint8_t v = curr_key != 0; std::memcpy(out, &v, 1); }
          else if (kind == U8 || kind == I8) { uint8_t v = (uint8_t)curr_key; std::memcpy(out, &v, 1); }
          else { uint16_t v = (uint16_t)curr_key; std::memcpy(out, &v, 2); }
          break;
        }
      }
    }
    // filtered_buffer.push_points(centroid) :161-164
    const size_t at = filtered_buffer.len();
    filtered_buffer.resize(at + 1);
    for (auto& a : layout.attributes) filtered_buffer.set_attribute(a.def, at, centroid.data() + a.offset);
  }
}

// ----------------------------------------------------------------------------------
// MinMax + minmax_attribute — pasture-core/src/math/minmax.rs:7-112, pasture-algorithms/src/minmax.rs:13-51
// ----------------------------------------------------------------------------------
template <typename T> inline T infimum(T self, T other) {
  if constexpr (std::is_floating_point<T>::value) return (self < other) ? self : other;  // math/minmax.rs:78-84
  else return std::min(self, other);                                                       // cmp::min :38-40
}
template <typename T> inline T supremum(T self, T other) {
  if constexpr (std::is_floating_point<T>::value) return (self > other) ? self : other;  // :86-92
  else return std::max(self, other);
}
// minmax.rs:13-51 with T == the stored datatype (the only reachable configuration).
// out_min/out_max receive `ncomp` values of T.  Returns false for an empty buffer.
template <typename T>
inline bool minmax_attribute_typed(const Buffer& buffer, const AttributeDef& attribute, int ncomp, T* out_min, T* out_max) {
  if (!buffer.point_layout().has_attribute_with_name(attribute.name))  // :17-26
    throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute " + attribute.display() + " not contained in PointLayout buffer");
  const AttributeMember* m = buffer.point_layout().get_attribute(attribute);  // view_attribute: buffer_views.rs:301-310
  if (!m) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  bool has = false;
  T v[3];
  for (size_t i = 0; i < buffer.len(); ++i) {
    buffer.get_attribute_unchecked(*m, i, (uint8_t*)v);
    if (!has) {  // None => Some((val, val)) :30
      for (int c = 0; c < ncomp; ++c) { out_min[c] = v[c]; out_max[c] = v[c]; }
      has = true;
    } else {     // (val.infimum(&old_min), val.supremum(&old_max)) :31-33
      for (int c = 0; c < ncomp; ++c) { out_min[c] = infimum<T>(v[c], out_min[c]); out_max[c] = supremum<T>(v[c], out_max[c]); }
    }
  }
  return has;
}
inline bool minmax_attribute(const Buffer& buffer, const AttributeDef& attribute, uint8_t* out_min, uint8_t* out_max) {
  switch (attribute.datatype.kind) {
    case U8: return minmax_attribute_typed<uint8_t>(buffer, attribute, 1, (uint8_t*)out_min, (uint8_t*)out_max);
    case I8: return minmax_attribute_typed<int8_t>(buffer, attribute, 1, (int8_t*)out_min, (int8_t*)out_max);
    case U16: return minmax_attribute_typed<uint16_t>(buffer, attribute, 1, (uint16_t*)out_min, (uint16_t*)out_max);
    case I16: return minmax_attribute_typed<int16_t>(buffer, attribute, 1, (int16_t*)out_min, (int16_t*)out_max);
    case U32: return minmax_attribute_typed<uint32_t>(buffer, attribute, 1, (uint32_t*)out_min, (uint32_t*)out_max);
    case I32: return minmax_attribute_typed<int32_t>(buffer, attribute, 1, (int32_t*)out_min, (int32_t*)out_max);
    case U64: return minmax_attribute_typed<uint64_t>(buffer, attribute, 1, (uint64_t*)out_min, (uint64_t*)out_max);
    case I64: return minmax_attribute_typed<int64_t>(buffer, attribute, 1, (int64_t*)out_min, (int64_t*)out_max);
    case F32: return minmax_attribute_typed<float>(buffer, attribute, 1, (float*)out_min, (float*)out_max);
    case F64: return minmax_attribute_typed<double>(buffer, attribute, 1, (double*)out_min, (double*)out_max);
    case Vec3u8: return minmax_attribute_typed<uint8_t>(buffer, attribute, 3, (uint8_t*)out_min, (uint8_t*)out_max);
    case Vec3u16: return minmax_attribute_typed<uint16_t>(buffer, attribute, 3, (uint16_t*)out_min, (uint16_t*)out_max);
    case Vec3f32: return minmax_attribute_typed<float>(buffer, attribute, 3, (float*)out_min, (float*)out_max);
    case Vec3i32: return minmax_attribute_typed<int32_t>(buffer, attribute, 3, (int32_t*)out_min, (int32_t*)out_max);
    case Vec3f64: return minmax_attribute_typed<double>(buffer, attribute, 3, (double*)out_min, (double*)out_max);
    default: throw Panic(ERR_INVALID_ARGUMENT, "MinMax is not implemented for datatype " + attribute.datatype.display());
  }
}

// transform_attribute — point_buffer.rs:391-404 (at / set_at per point) with an affine closure
// (loop shape of reproject_point_cloud_within, pasture-algorithms/src/reprojection.rs:132-146)
inline void transform_attribute(Buffer& buffer, const AttributeDef& attribute, const TransformDesc& xf) {
  if (xf.datatype != attribute.datatype)  // AttributeViewMut::new asserts T::data_type() == attribute.datatype()
    throw Panic(ERR_TRANSFORM_TYPE_MISMATCH, "assertion failed: T::data_type() == attribute.datatype()");
  const AttributeMember* m = buffer.point_layout().get_attribute(attribute);
  if (!m) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  AttributeTransformFn f = make_transform_fn(xf);
  std::vector<uint8_t> v(m->size);
  for (size_t i = 0; i < buffer.len(); ++i) {
    buffer.get_attribute_unchecked(*m, i, v.data());
    f(v.data());
    buffer.set_attribute(attribute, i, v.data());
  }
}

// ----------------------------------------------------------------------------------
// Normal estimation — pasture-algorithms/src/normal_estimation.rs:79-476
// ----------------------------------------------------------------------------------
struct Mat3 { double m[3][3]; };  // m[row][col]

inline bool is_finite3(const double p[3]) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }  // :143-148
inline bool is_dense(const double (*pts)[3], size_t n) {  // :133-140
  for (size_t i = 0; i < n; ++i)
    if (std::isnan(pts[i][0]) || std::isnan(pts[i][1]) || std::isnan(pts[i][2])) return false;
  return true;
}
// compute_centroid :198-237
inline void compute_centroid(const double (*pts)[3], size_t n, double centroid[3]) {
  if (n == 0) throw Panic(ERR_TOO_FEW_POINTS, "The point cloud is empty!");
  double temp[3] = {0, 0, 0};
  if (is_dense(pts, n)) {
    for (size_t i = 0; i < n; ++i) { temp[0] += pts[i][0]; temp[1] += pts[i][1]; temp[2] += pts[i][2]; }
    for (int c = 0; c < 3; ++c) centroid[c] = temp[c] / (double)n;
  } else {
    int64_t cnt = 0;
    for (size_t i = 0; i < n; ++i)
      if (is_finite3(pts[i])) { temp[0] += pts[i][0]; temp[1] += pts[i][1]; temp[2] += pts[i][2]; cnt += 1; }
    for (int c = 0; c < 3; ++c) centroid[c] = temp[c] / (double)cnt;
  }
}
// compute_centroid(&buffer) :198-237 on a point buffer: view_attribute::<Vector3<f64>>(&POSITION_3D) (exact datatype, buffer_views.rs:301-310)
inline void compute_centroid(const Buffer& point_cloud, double centroid[3]) {
  if (point_cloud.len() == 0) throw Panic(ERR_TOO_FEW_POINTS, "The point cloud is empty!");
  const AttributeMember* m = point_cloud.point_layout().get_attribute(POSITION_3D);
  if (!m) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  std::vector<double> pts(point_cloud.len() * 3);
  for (size_t i = 0; i < point_cloud.len(); ++i) point_cloud.get_attribute_unchecked(*m, i, reinterpret_cast<uint8_t*>(&pts[3 * i]));
  compute_centroid(reinterpret_cast<const double (*)[3]>(pts.data()), point_cloud.len(), centroid);
}
// compute_covariance_matrix :240-305.  Returns false for Err("... not enough to span a plane!").
inline bool compute_covariance_matrix(const double (*pts)[3], size_t n, Mat3& cov) {
  for (auto& r : cov.m) for (auto& x : r) x = 0.0;
  size_t point_count = 0;
  double centroid[3];
  compute_centroid(pts, n, centroid);
  auto accumulate = [&](const double p[3]) {
    double d[3] = {p[0] - centroid[0], p[1] - centroid[1], p[2] - centroid[2]};
    cov.m[1][1] += d[1] * d[1];
    cov.m[1][2] += d[1] * d[2];
    cov.m[2][2] += d[2] * d[2];
    double dx = d[0];
    d[0] *= dx; d[1] *= dx; d[2] *= dx;
    cov.m[0][0] += d[0];
    cov.m[0][1] += d[1];
    cov.m[0][2] += d[2];
  };
  if (is_dense(pts, n)) {
    point_count = n;
    for (size_t i = 0; i < n; ++i) accumulate(pts[i]);
  } else {
    for (size_t i = 0; i < n; ++i) {
      if (!is_finite3(pts[i])) continue;
      accumulate(pts[i]);
      point_count += 1;
    }
  }
  if (point_count < 3) return false;
  cov.m[1][0] = cov.m[0][1];
  cov.m[2][0] = cov.m[0][2];
  cov.m[2][1] = cov.m[1][2];
  return true;
}
// solve_polynomial_quadratic :308-325
inline void solve_polynomial_quadratic(double c2, double c1, double ev[3]) {
  ev[0] = 0.0;
  double delta = c2 * c2 - 4.0 * c1;
  if (delta < 0.0) delta = 0.0;
  double sd = std::sqrt(delta);
  ev[2] = 0.5 * (c2 + sd);
  ev[1] = 0.5 * (c2 - sd);
}
// solve_polynomial :328-392
inline void solve_polynomial(const Mat3& C, double ev[3]) {
  const auto& m = C.m;
  double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
              m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
  double c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] -
              m[1][2] * m[1][2];
  double c2 = m[0][0] + m[1][1] + m[2][2];
  if (std::fabs(c0) < DBL_EPSILON) { solve_polynomial_quadratic(c2, c1, ev); return; }
  const double one_third = 1.0 / 3.0;
  const double sqrt_3 = std::sqrt(3.0);
  double c2_third = c2 * one_third;
  double alpha_third = (c1 - c2 * c2_third) * one_third;
  if (alpha_third > 0.0) alpha_third = 0.0;
  double half_beta = 0.5 * (c0 + c2_third * (2.0 * c2_third * c2_third - c1));
  double q = half_beta * half_beta + alpha_third * alpha_third * alpha_third;
  if (q > 0.0) q = 0.0;
  double rho = std::sqrt(-alpha_third);
  double theta = std::atan2(std::sqrt(-q), half_beta) * one_third;
  double ct = std::cos(theta), st = std::sin(theta);
  ev[0] = c2_third + 2.0 * rho * ct;
  ev[1] = c2_third - rho * (ct + sqrt_3 * st);
  ev[2] = c2_third - rho * (ct - sqrt_3 * st);
  std::sort(ev, ev + 3);  // partial_cmp().unwrap(): NaN would panic in the reference
  if (ev[0] <= 0.0) solve_polynomial_quadratic(c2, c1, ev);
}
// get_largest_eigen_vector :395-426 (first maximum wins; result NOT normalised)
inline void get_largest_eigen_vector(const Mat3& S, double out[3]) {
  auto cross = [](const double a[3], const double b[3], double r[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
  };
  double rows[3][3];
  cross(S.m[0], S.m[1], rows[0]);
  cross(S.m[0], S.m[2], rows[1]);
  cross(S.m[1], S.m[2], rows[2]);
  auto norm = [](const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  int best = 0;
  for (int r = 0; r < 3; ++r) if (norm(rows[r]) > norm(rows[best])) best = r;
  for (int c = 0; c < 3; ++c) out[c] = rows[best][c];
}
// eigen_3x3 :429-453
inline void eigen_3x3(const Mat3& C, double& eigen_value, double eigen_vector[3]) {
  double scale = 0.0;  // covariance_matrix.abs().max()
  bool first = true;
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) {  // column-major iteration like nalgebra
    double a = std::fabs(C.m[r][c]);
    if (first || a > scale) { scale = a; first = false; }
  }
  Mat3 S;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S.m[r][c] = C.m[r][c] / scale;
  double ev[3];
  solve_polynomial(C, ev);          // eigenvalues of the UNSCALED matrix :441
  eigen_value = ev[0] * scale;      // "undo scale" :443 (sic)
#ifdef ORC_DIAGONAL_SUBTRACT
  for (int d = 0; d < 3; ++d) S.m[d][d] -= ev[0];  // what :446-449 would do if diagonal() were a view
#endif
  // :446-449 operates on the owned copy returned by DMatrix::diagonal() => no effect on S
  get_largest_eigen_vector(S, eigen_vector);
}
// solve_plane_parameter :456-467
inline void solve_plane_parameter(const Mat3& C, double normal[3], double& curvature) {
  double eigen_value;
  eigen_3x3(C, eigen_value, normal);
  double eigen_sum = C.m[0][0] + C.m[1][1] + C.m[2][2];
  curvature = (eigen_sum != 0.0) ? std::fabs(eigen_value / eigen_sum) : 0.0;
}

// Exact k-nearest-neighbour search standing in for kd-tree 0.3.0 `KdTree::build_by_ordered_float` +
// `nearests` (normal_estimation.rs:103,108; crate not vendored).  Published contract restated: the k items
// with the smallest squared Euclidean distance to the query (the query point itself included), sorted by
// ascending distance; fewer than k items if the tree is smaller.  Tie order is the crate's => unpinned.
struct KdTree {
  struct Node { int32_t left = -1, right = -1; uint32_t idx = 0; uint8_t axis = 0; };
  const double (*pts)[3];
  std::vector<Node> nodes;
  int32_t root = -1;
  KdTree(const double (*p)[3], size_t n) : pts(p) {
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    nodes.reserve(n);
    root = build(idx.data(), n, 0);
  }
  int32_t build(uint32_t* idx, size_t n, int depth) {
    if (n == 0) return -1;
    int axis = depth % 3;
    size_t mid = n / 2;
    std::nth_element(idx, idx + mid, idx + n, [&](uint32_t a, uint32_t b) { return pts[a][axis] < pts[b][axis]; });
    int32_t me = (int32_t)nodes.size();
    nodes.push_back(Node{-1, -1, idx[mid], (uint8_t)axis});
    int32_t l = build(idx, mid, depth + 1);
    int32_t r = build(idx + mid + 1, n - mid - 1, depth + 1);
    nodes[me].left = l; nodes[me].right = r;
    return me;
  }
  using Hit = std::pair<double, uint32_t>;  // (squared distance, index); max-heap on distance
  void search(int32_t ni, const double q[3], size_t k, std::vector<Hit>& heap) const {
    if (ni < 0) return;
    const Node& nd = nodes[ni];
    const double* p = pts[nd.idx];
    double d0 = p[0] - q[0], d1 = p[1] - q[1], d2 = p[2] - q[2];
    double d = d0 * d0 + d1 * d1 + d2 * d2;
    if (!(d == d)) d = INFINITY;  // a NaN coordinate never compares: such points rank last (tie order is the crate's, unpinned)
    if (heap.size() < k) { heap.push_back({d, nd.idx}); std::push_heap(heap.begin(), heap.end()); }
    else if (d < heap.front().first) { std::pop_heap(heap.begin(), heap.end()); heap.back() = {d, nd.idx}; std::push_heap(heap.begin(), heap.end()); }
    double delta = q[nd.axis] - p[nd.axis];
    int32_t near = delta < 0 ? nd.left : nd.right, far = delta < 0 ? nd.right : nd.left;
    search(near, q, k, heap);
    if (heap.size() < k || delta * delta <= heap.front().first) search(far, q, k, heap);
  }
  std::vector<uint32_t> nearests(const double q[3], size_t k) const {
    std::vector<Hit> heap;
    heap.reserve(k + 1);
    search(root, q, k, heap);
    std::sort(heap.begin(), heap.end());
    std::vector<uint32_t> out;
    for (auto& h : heap) out.push_back(h.second);
    return out;
  }
};

// compute_normals :79-130.  Output: n x (normal[3], curvature).  `out_knn` (optional) receives the
// neighbour indices (n x k, -1 padded) so a test can cross-check the neighbour SETS independently.
inline void compute_normals(const Buffer& point_cloud, size_t k_nn, double* out_normals, double* out_curvature,
                            int64_t* out_knn = nullptr) {
  if (point_cloud.len() < 3)  // :86-88
    throw Panic(ERR_TOO_FEW_POINTS, "The point cloud is too small. Please use a point cloud that has 3 or more points!");
  if (k_nn < 3) throw Panic(ERR_K_TOO_SMALL, "The k nearest neigbors attribute is too small!");  // :89-91
  const AttributeMember* m = point_cloud.point_layout().get_attribute(POSITION_3D);  // view_attribute::<Vector3<f64>>
  if (!m) throw Panic(ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  const size_t n = point_cloud.len();
  std::vector<std::array<double, 3>> points(n);  // :97-100
  for (size_t i = 0; i < n; ++i) point_cloud.get_attribute_unchecked(*m, i, (uint8_t*)points[i].data());
  const double (*pts)[3] = reinterpret_cast<const double (*)[3]>(points.data());
  KdTree tree(pts, n);  // :103
  std::vector<std::array<double, 3>> nb;
  for (size_t i = 0; i < n; ++i) {  // :106
    double q[3];
    point_cloud.get_attribute_unchecked(*m, i, (uint8_t*)q);
    std::vector<uint32_t> nearest = tree.nearests(q, k_nn);  // :108
    // :111-120 — a fresh HashMapBuffer with the FULL source layout per point; only positions are set
    HashMapBuffer knn_buffer(point_cloud.point_layout());
    knn_buffer.resize(nearest.size());
    for (size_t j = 0; j < nearest.size(); ++j) knn_buffer.set_attribute(POSITION_3D, j, (const uint8_t*)pts[nearest[j]]);
    nb.resize(nearest.size());
    for (size_t j = 0; j < nearest.size(); ++j) knn_buffer.get_attribute_unchecked(*m, j, (uint8_t*)nb[j].data());
    Mat3 cov;
    if (!compute_covariance_matrix(reinterpret_cast<const double (*)[3]>(nb.data()), nb.size(), cov))  // unwrap :471
      throw Panic(ERR_NOT_ENOUGH_NEIGHBOURS,
                  "The number of valid (finite and non-NaN values) points in a k nearest neighborhood is not enough to span a plane!");
    solve_plane_parameter(cov, out_normals + 3 * i, out_curvature[i]);
    if (out_knn)
      for (size_t j = 0; j < k_nn; ++j) out_knn[i * k_nn + j] = j < nearest.size() ? (int64_t)nearest[j] : -1;
  }
}

// The per-point step of compute_normals (:108-123) for neighbour lists GIVEN by the caller instead of found by the kd-tree: the positions of
// knn[i][0 .. k) (in that order; an index < 0 ends the list) go through compute_covariance_matrix + solve_plane_parameter exactly as above.
// Test infrastructure for clouds with exact distance ties (quantised LAS coordinates), where the un-vendored kd-tree crate's tie order is
// unpinned: the GPU's own lists, once verified as exact k-nearest sets, pin the FIT of every query.
inline void fit_neighbourhoods(const double (*pts)[3], size_t n_queries, const int64_t* knn, size_t k, double* out_normals, double* out_curvature) {
  std::vector<std::array<double, 3>> nb;
  for (size_t i = 0; i < n_queries; ++i) {
    nb.clear();
    for (size_t j = 0; j < k && knn[i * k + j] >= 0; ++j) {
      const double* p = pts[knn[i * k + j]];
      nb.push_back({p[0], p[1], p[2]});
    }
    Mat3 cov;
    if (!compute_covariance_matrix(reinterpret_cast<const double (*)[3]>(nb.data()), nb.size(), cov))
      throw Panic(ERR_NOT_ENOUGH_NEIGHBOURS,
                  "The number of valid (finite and non-NaN values) points in a k nearest neighborhood is not enough to span a plane!");
    solve_plane_parameter(cov, out_normals + 3 * i, out_curvature[i]);
  }
}

// ----------------------------------------------------------------------------------
// Synthetic inputs — SURVEY.md §8(d): u = splitmix64(seed ^ (3*i + c)), f = (u >> 11) * 2^-53
// ----------------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline double synth_unit(uint64_t seed, uint64_t i, uint64_t c) {
  return (double)(splitmix64(seed ^ (3 * i + c)) >> 11) * (1.0 / 9007199254740992.0);
}
inline void synth_position(uint64_t seed, uint64_t i, double p[3]) {
  p[0] = synth_unit(seed, i, 0) * 1000.0;
  p[1] = synth_unit(seed, i, 1) * 1000.0;
  p[2] = synth_unit(seed, i, 2) * 100.0;
}

}  // namespace orc
