// pasture_amd host runtime — shared declarations (C++17, compiled by hipcc as host code).
//
// Mirrors the reference's descriptor types for the hot path:
//   PointAttributeDataType / PointAttributeDefinition / PointAttributeMember / PointLayout
//   (pasture-core/src/layout/point_layout.rs:23-127, 261-431, 648-997)
// and turns the reference's panics into status codes + a thread-local message (include/pasture_amd.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pasture_amd.h"

namespace pst {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

// A launch / allocation helper reported failure: the thread's last HIP error names the cause (and is reset by the read).  Running out of device memory is
// PST_ERR_OUT_OF_MEMORY whichever entry point hits it.
inline Error hip_failure(const std::string& what) {
  const hipError_t e = hipGetLastError();
  return Error(e == hipErrorOutOfMemory ? PST_ERR_OUT_OF_MEMORY : PST_ERR_HIP, what + hipGetErrorString(e));
}

#define PST_HIP_CHECK(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      int _code = (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? PST_ERR_NO_DEVICE        \
                  : (_e == hipErrorOutOfMemory)                           ? PST_ERR_OUT_OF_MEMORY    \
                                                                          : PST_ERR_HIP;             \
      (void)hipGetLastError(); /* reported through the status: the NEXT call must not find it */     \
      throw ::pst::Error(_code, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
    }                                                                                                \
  } while (0)

#define PST_API_BEGIN try {
#define PST_API_END                                                                 \
  }                                                                                 \
  catch (const ::pst::Error& e) { ::pst::set_last_error(e.what()); (void)hipGetLastError(); return e.code; } \
  catch (const std::bad_alloc&) { ::pst::set_last_error("host allocation failed"); return PST_ERR_OUT_OF_MEMORY; } \
  catch (const std::exception& e) { ::pst::set_last_error(e.what()); return PST_ERR_INVALID_ARGUMENT; }           \
  return PST_OK;

template <typename T>
inline T* not_null(T* p, const char* what) {
  if (!p) throw Error(PST_ERR_INVALID_ARGUMENT, std::string(what) + " must not be NULL");
  return p;
}

// ---- current device stream (thread-local; pst_set_stream) ------------------------------------------
hipStream_t current_stream();
void ensure_device();  // throws PST_ERR_NO_DEVICE when no HIP device is usable

// ---- datatypes ---------------------------------------------------------------------------------------
// Scalar component types the kernels compute in.
enum CompType : uint8_t { CT_U8 = 0, CT_I8, CT_U16, CT_I16, CT_U32, CT_I32, CT_U64, CT_I64, CT_F32, CT_F64 };

struct DataType {
  uint32_t kind = PST_U8;
  uint64_t size_param = 0;
  uint64_t align_param = 0;
  std::array<uint8_t, 16> uuid{};

  static DataType from_c(const pst_datatype* d);
  pst_datatype to_c() const;
  uint64_t size() const;           // point_layout.rs:72-95
  uint64_t min_alignment() const;  // point_layout.rs:98-126
  std::string display() const;     // Display impl :129-158
  bool is_vec3() const { return kind >= PST_VEC3U8 && kind <= PST_VEC3F64; }
  bool is_scalar() const { return kind <= PST_F64; }
  // component type + count for the kernels (opaque kinds are moved as raw bytes)
  CompType comp_type() const;
  uint32_t num_components() const;
  bool operator==(const DataType& o) const;
  bool operator!=(const DataType& o) const { return !(*this == o); }
};

inline uint64_t align_up(uint64_t v, uint64_t a) { return a == 0 ? v : ((v + a - 1) / a) * a; }  // math/arithmetic.rs:8-66

struct AttributeDef {
  std::string name;
  DataType datatype;
  bool operator==(const AttributeDef& o) const { return name == o.name && datatype == o.datatype; }
};
struct Member {
  AttributeDef def;
  uint64_t offset = 0;
  uint64_t size = 0;
  bool operator==(const Member& o) const { return def == o.def && offset == o.offset && size == o.size; }
};

struct Layout {
  std::vector<Member> members;
  uint64_t size = 0;   // memory_layout.size()
  uint64_t align = 1;  // memory_layout.align()

  void add_attribute(const AttributeDef& def, bool packed, uint64_t max_alignment);                // :778-822
  static Layout from_members_and_alignment(const std::vector<Member>& ms, uint64_t type_alignment);  // :719-759
  const Member* find(const AttributeDef& d) const;            // get_attribute (name + datatype) :882-890
  const Member* find_by_name(const std::string& n) const;     // get_attribute_by_name :892-896
  int index_of(const AttributeDef& d) const;
  bool operator==(const Layout& o) const { return members == o.members && size == o.size && align == o.align; }
  bool operator!=(const Layout& o) const { return !(*this == o); }
};

// The kernels address a point's bytes with 32-bit strides and sizes (plan.h): a layout whose points (or one attribute: ByteArray(n) takes a u64 in the
// reference) reach 4 GiB is refused where it meets device memory -- buffers and converters -- instead of being truncated there.  The layout itself can be
// built and queried like any other (host-only logic).
inline void check_layout_fits_kernels(const Layout& l, const char* what) {
  bool ok = l.size < (1ull << 32);
  for (const Member& m : l.members) ok = ok && m.size < (1ull << 32) && m.offset < (1ull << 32);
  if (!ok) throw Error(PST_ERR_UNSUPPORTED, std::string(what) + ": points (or attributes) of 4 GiB and more are not supported on the device");
}

}  // namespace pst

// opaque C handles
struct pst_layout { pst::Layout l; };
