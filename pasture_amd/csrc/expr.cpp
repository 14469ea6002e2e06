// Device expressions: user-written transformations and predicates compiled at run time (hipRTC) into the library's kernels.
//
// What they stand in for.  The reference takes ANY closure where this library used to take a closed descriptor set:
//   * BufferLayoutConverter::set_custom_mapping_with_transformation<T, F: Fn(T) -> T>      buffer_conversion.rs:13-36, 194-234
//   * BorrowedMutBufferExt::transform_attribute<T, F: Fn(usize, T) -> T>                    point_buffer.rs:391-404
//   * HashMapBuffer::filter / filter_into<F: Fn(usize) -> bool>                             point_buffer.rs:1064-1136
// A Rust closure cannot cross a C ABI into a GPU kernel; its SOURCE TEXT can.  An expression is C++ expression syntax over a fixed set of
// names; the library wraps it into a kernel (below), compiles it through the same hipRTC pipeline as the plan-specialised conversion kernels
// (jit.cpp: cached in memory per source text and on disk per source hash, compiled with -ffp-contract=off so that `v * s + o` keeps its two
// roundings) and launches it.  A syntax error is PST_ERR_UNSUPPORTED_TRANSFORM with the compiler's log as the message.
//
// Transformation expressions (mapping transformations and transform_attribute).  The closure's argument type T is the attribute's datatype
// (the SOURCE attribute's when apply_to_source, the TARGET's otherwise -- buffer_conversion.rs:209-213).  Names:
//     v          this component of the value, of T's component type (the value itself for scalar attributes)
//     x, y, z    the three components of a Vec3 value (scalar attributes: x = y = z = v)
//     c          the component being computed: 0, 1, 2 (int)
//     i          the point's index in the buffer the call was made on (uint64_t): the `usize` of transform_attribute's closure
//     p0 .. p3   caller-provided device arrays of double (what a closure would capture): `p0[3 * i + c]`
// One expression is evaluated per component; a Vec3 attribute may give three, separated by `;` (x ; y ; z).  The result is converted to T's
// component type with Rust `as` (so integer attributes saturate / truncate as `as` does); the usual C++ arithmetic conversions apply INSIDE
// the expression (a u16 times 2.0 is a double).  <cmath> functions of the HIP device runtime are available (sqrt, fabs, floor, fmin, ...).
//
// Predicate expressions (filter).  Names: every attribute of the buffer's layout whose name is a C identifier -- scalars as values of their
// type, Vec3 attributes as structs with .x .y .z --, plus i and p0 .. p3: `Classification == 2 && Position3D.z < 120.0`.
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include <memory>

#include "jit.hpp"
#include "kernels.hpp"
#include "runtime.hpp"

using namespace pst;

namespace pstexpr {

const char* ct_name(uint32_t ct) {
  static const char* names[10] = {"uint8_t", "int8_t", "uint16_t", "int16_t", "uint32_t", "int32_t", "uint64_t", "int64_t", "float", "double"};
  return ct < 10 ? names[ct] : "uint8_t";
}

// split at top-level ';' (none inside parentheses / brackets)
static std::vector<std::string> split_components(const std::string& expr) {
  std::vector<std::string> out;
  std::string cur;
  int depth = 0;
  for (char ch : expr) {
    if (ch == '(' || ch == '[') ++depth;
    if (ch == ')' || ch == ']') --depth;
    if (depth < 0) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression: a closing parenthesis or bracket without its opening one");
    if (ch == ';' && depth == 0) { out.push_back(cur); cur.clear(); }
    else cur += ch;
  }
  if (depth != 0) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression: unbalanced parentheses or brackets");
  out.push_back(cur);
  auto blank = [](const std::string& s) { for (char c : s) if (!std::isspace((unsigned char)c)) return false; return true; };
  if (out.size() > 1 && blank(out.back())) out.pop_back();  // a trailing ';'
  for (const std::string& s : out)
    if (blank(s)) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression: an empty component expression");
  return out;
}
std::vector<std::string> split_expression_components(const std::string& expr) { return split_components(expr); }  // (jit.cpp: fused into a plan's kernel)

static void check_text(const char* expr) {
  if (!expr || !*expr) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression must not be empty");
  if (std::strlen(expr) > 4096) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression longer than 4096 characters");
  // the text is pasted into a function body: keep it an EXPRESSION (no statements, no preprocessor, no string / character literals)
  for (const char* p = expr; *p; ++p)
    if (*p == '{' || *p == '}' || *p == '#' || *p == '"' || *p == '\'' || *p == '\\' || *p == '\n' || *p == '\r')
      throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, std::string("expression: character '") + *p + "' is not allowed (an expression, not statements)");
  // The text is pasted between `rust_as<T>(` / `(bool)(` and `)`: parentheses and brackets must nest and close, or `x) , (y` would escape the cast
  // that wraps it (round-5 advisor finding).
  std::string open;
  for (const char* p = expr; *p; ++p) {
    if (*p == '(' || *p == '[') open += *p;
    else if (*p == ')' || *p == ']') {
      if (open.empty() || open.back() != (*p == ')' ? '(' : '['))
        throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, std::string("expression: '") + *p + "' closes nothing that is open at that point");
      open.pop_back();
    }
  }
  if (!open.empty()) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, std::string("expression: '") + open.back() + "' is never closed");
}

struct MapSpec {
  uint32_t src_ct = 9, dst_ct = 9, ncomp = 1;
  bool pre = false;  // the expression sees the SOURCE value (apply_to_source), the conversion follows; otherwise the converted value
  std::string expr;
};

// The translation unit of one mapping / transform_attribute kernel.
std::string map_source(const MapSpec& s) {
  std::vector<std::string> comps = split_components(s.expr);
  if (comps.size() != 1 && comps.size() != s.ncomp)
    throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "expression: " + std::to_string(comps.size()) + " component expressions for a value of " + std::to_string(s.ncomp) +
                                                   " component(s): give one, or one per component separated by ';'");
  std::string t = "// pasture_amd device expression (expr.cpp): one mapping, strided source and target\n#include \"device_common.hpp\"\nusing namespace pstd;\n";
  t += std::string("typedef ") + ct_name(s.src_ct) + " TS;\ntypedef " + ct_name(s.dst_ct) + " TD;\ntypedef " + (s.pre ? "TS" : "TD") + " TI;\n";
  t += "#define PST_NC " + std::to_string(s.ncomp) + "\n#define PST_PRE " + (s.pre ? "1" : "0") + "\n";
  for (uint32_t c = 0; c < s.ncomp; ++c) {
    t += "__device__ __forceinline__ TI pst_expr_" + std::to_string(c) +
         "(const TI v, const TI x, const TI y, const TI z, const int c, const uint64_t i, const double* __restrict__ p0, const double* __restrict__ p1, "
         "const double* __restrict__ p2, const double* __restrict__ p3) {\n  (void)v; (void)x; (void)y; (void)z; (void)c; (void)i; (void)p0; (void)p1; (void)p2; (void)p3;\n"
         "  return rust_as<TI>(\n" + comps[comps.size() == 1 ? 0 : c] + "\n  );\n}\n";
  }
  t += R"(extern "C" __global__ __launch_bounds__(256) void pst_jit_expr_map(uint64_t src, uint64_t sstride, uint64_t dst, uint64_t dstride, uint64_t n, uint64_t first,
                                                                     const double* p0, const double* p1, const double* p2, const double* p3) {
  for (uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256u) {
    const uint64_t i = first + e;
    TI in[3];
#pragma unroll
    for (int c = 0; c < PST_NC; ++c) {
      const TS s = load_un<TS>((cgptr_t)as_global(src) + e * sstride + c * sizeof(TS));
      if (PST_PRE) in[c] = (TI)s; else in[c] = (TI)rust_as<TD>(s);
    }
#pragma unroll
    for (int c = PST_NC; c < 3; ++c) in[c] = in[0];
    TI r[3];
)";
  for (uint32_t c = 0; c < s.ncomp; ++c)
    t += "    r[" + std::to_string(c) + "] = pst_expr_" + std::to_string(c) + "(in[" + std::to_string(c) + "], in[0], in[1], in[2], " + std::to_string(c) + ", i, p0, p1, p2, p3);\n";
  t += R"(#pragma unroll
    for (int c = 0; c < PST_NC; ++c) store_un<TD>(as_global(dst) + e * dstride + c * sizeof(TD), rust_as<TD>(r[c]));
  }
}
)";
  return t;
}

struct PredAttr {
  std::string name;
  uint32_t ct = 0, ncomp = 1;
  uint64_t base = 0, stride = 0;
};
constexpr size_t kMaxPredAttrs = 16;
struct PredArgs { uint64_t base[kMaxPredAttrs], stride[kMaxPredAttrs]; };

static bool is_identifier(const std::string& s) {
  if (s.empty() || !(std::isalpha((unsigned char)s[0]) || s[0] == '_')) return false;
  for (char c : s) if (!(std::isalnum((unsigned char)c) || c == '_')) return false;
  return true;
}
// identifiers of `expr` (tokens that start with a letter or '_' and are not preceded by '.': member names like `.x` are not attribute names)
static std::vector<std::string> identifiers(const std::string& expr) {
  std::vector<std::string> out;
  for (size_t p = 0; p < expr.size();) {
    const unsigned char ch = (unsigned char)expr[p];
    if (std::isalpha(ch) || ch == '_') {
      size_t q = p;
      while (q < expr.size() && (std::isalnum((unsigned char)expr[q]) || expr[q] == '_')) ++q;
      size_t b = p;
      while (b > 0 && std::isspace((unsigned char)expr[b - 1])) --b;
      if (!(b > 0 && expr[b - 1] == '.')) out.push_back(expr.substr(p, q - p));
      p = q;
    } else if (std::isdigit(ch)) {  // a number (and its suffix / exponent letters)
      while (p < expr.size() && (std::isalnum((unsigned char)expr[p]) || expr[p] == '.')) ++p;
    } else {
      ++p;
    }
  }
  return out;
}

// `PstV3` and the predicate as a device function of the named attributes (by value), the index and the parameter arrays: shared by the three
// translation units a predicate can be part of (byte-mask kernel, per-tile count kernel, the streaming compaction kernel of filter_stream.hpp)
std::string pred_function_text(const std::vector<PredAttr>& attrs, const std::string& expr) {
  std::string t = "template <typename T> struct PstV3 { T x, y, z; };\n";
  std::string params;
  for (size_t a = 0; a < attrs.size(); ++a) {
    const std::string ty = attrs[a].ncomp == 3 ? std::string("PstV3<") + ct_name(attrs[a].ct) + ">" : std::string(ct_name(attrs[a].ct));
    params += "const " + ty + " " + attrs[a].name + ", ";
  }
  t += "__device__ __forceinline__ bool pst_pred(" + params +
       "const uint64_t i, const double* __restrict__ p0, const double* __restrict__ p1, const double* __restrict__ p2, const double* __restrict__ p3) {\n"
       "  using namespace pstd;\n  (void)i; (void)p0; (void)p1; (void)p2; (void)p3;\n  return (bool)(\n" + expr + "\n  );\n}\n";
  return t;
}
// the loads of point `e`'s named attributes (v0, v1, ...) and the argument list that hands them to pst_pred
static std::string pred_loads(const std::vector<PredAttr>& attrs, std::string* args) {
  std::string t;
  for (size_t a = 0; a < attrs.size(); ++a) {
    const std::string T = ct_name(attrs[a].ct), A = std::to_string(a), q = "(cgptr_t)as_global(a.base[" + A + "]) + e * a.stride[" + A + "]";
    if (attrs[a].ncomp == 3)
      t += "    const PstV3<" + T + "> v" + A + " = {load_un<" + T + ">(" + q + "), load_un<" + T + ">(" + q + " + sizeof(" + T + ")), load_un<" + T + ">(" + q + " + 2 * sizeof(" + T + "))};\n";
    else
      t += "    const " + T + " v" + A + " = load_un<" + T + ">(" + q + ");\n";
    *args += "v" + A + ", ";
  }
  return t;
}
// The count pass of a compaction whose predicate is fused (round 6): one wave per 2048-point tile evaluates the predicate on the columns it names
// and leaves the tile's number of matches where filter.hip's scan expects it -- the mask_count_kernel of a mask that is never written.
std::string pred_count_source(const std::vector<PredAttr>& attrs, const std::string& expr) {
  std::string t = "// pasture_amd device expression (expr.cpp): filter predicate -> matches per 2048-point tile\n#include \"device_common.hpp\"\nusing namespace pstd;\n";
  t += "struct PstPredArgs { uint64_t base[" + std::to_string(kMaxPredAttrs) + "], stride[" + std::to_string(kMaxPredAttrs) + "]; };\n";
  t += pred_function_text(attrs, expr);
  std::string args;
  const std::string loads = pred_loads(attrs, &args);
  t += "extern \"C\" __global__ __launch_bounds__(256) void pst_jit_pred_count(const PstPredArgs a, uint64_t n, uint64_t first, uint32_t* __restrict__ counts,\n"
       "                                                                       const double* p0, const double* p1, const double* p2, const double* p3) {\n"
       "  const uint32_t lane = threadIdx.x & 63u;\n  const uint64_t tile = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);\n  const uint64_t b = tile * 2048u;\n"
       "  if (b >= n) return;\n  const uint64_t end = b + 2048u < n ? b + 2048u : n;\n  uint32_t c = 0;\n"
       "  for (uint64_t e = b + lane; e < end; e += 64u) {\n" + loads +
       "    c += pst_pred(" + args + "first + e, p0, p1, p2, p3) ? 1u : 0u;\n  }\n"
       "#pragma unroll\n  for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);\n  if (lane == 0) counts[tile] = c;\n}\n";
  return t;
}

std::string pred_source(const std::vector<PredAttr>& attrs, const std::string& expr) {
  std::string t = "// pasture_amd device expression (expr.cpp): filter predicate -> byte mask\n#include \"device_common.hpp\"\nusing namespace pstd;\n";
  t += "struct PstPredArgs { uint64_t base[" + std::to_string(kMaxPredAttrs) + "], stride[" + std::to_string(kMaxPredAttrs) + "]; };\n";
  t += pred_function_text(attrs, expr);
  std::string args;
  t += "extern \"C\" __global__ __launch_bounds__(256) void pst_jit_expr_pred(const PstPredArgs a, uint64_t n, uint64_t first, uint8_t* __restrict__ mask,\n"
       "                                                                      const double* p0, const double* p1, const double* p2, const double* p3) {\n"
       "  for (uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256u) {\n";
  t += pred_loads(attrs, &args);
  t += "    mask[e] = pst_pred(" + args + "first + e, p0, p1, p2, p3) ? (uint8_t)1 : (uint8_t)0;\n  }\n}\n";
  return t;
}

static pstjit::Kernel compile(const std::string& source, const char* entry, const char* what) {
  if (pstjit::mode() == pstjit::Mode::Off)
    throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, std::string(what) + ": device expressions need the run-time compiler (PST_JIT=0 switches it off)");
  pstjit::Kernel k;
  std::string error;
  if (!pstjit::acquire_source(source, entry, 256, 0, 0, pstjit::Acquire::Wait, &k, &error))
    throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, std::string(what) + ": the expression does not compile:\n" + error);
  return k;
}
static unsigned grid_for(uint64_t n) {
  const uint64_t blocks = (n + 255) / 256;
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(blocks, (uint64_t)pstk::device_cus() * 16));
}
static void fill_params(const double* const* device_params, size_t n_params, const double* p[4]) {
  if (n_params > 4) throw Error(PST_ERR_INVALID_ARGUMENT, "at most 4 parameter arrays (p0 .. p3)");
  for (size_t a = 0; a < 4; ++a) p[a] = a < n_params ? not_null(device_params, "device_params")[a] : nullptr;
}

void launch_map(const MapSpec& s, uint64_t src, uint64_t sstride, uint64_t dst, uint64_t dstride, uint64_t n, uint64_t first_index, const double* const p[4], hipStream_t stream) {
  const pstjit::Kernel k = compile(map_source(s), "pst_jit_expr_map", "transformation expression");
  if (n == 0) return;
  const double *p0 = p ? p[0] : nullptr, *p1 = p ? p[1] : nullptr, *p2 = p ? p[2] : nullptr, *p3 = p ? p[3] : nullptr;
  void* args[] = {&src, &sstride, &dst, &dstride, &n, &first_index, &p0, &p1, &p2, &p3};
  if (hipModuleLaunchKernel(k.fn, grid_for(n), 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
    throw hip_failure("expression kernel launch failed: ");
  pstk::note_plan_kind(PST_PLAN_EXPRESSION);
}

void launch_pred(const std::vector<PredAttr>& attrs, const std::string& expr, uint64_t n, uint64_t first_index, uint8_t* mask_dev, const double* const p[4], hipStream_t stream) {
  const pstjit::Kernel k = compile(pred_source(attrs, expr), "pst_jit_expr_pred", "predicate expression");
  if (n == 0) return;
  PredArgs a{};
  for (size_t i = 0; i < attrs.size(); ++i) { a.base[i] = attrs[i].base; a.stride[i] = attrs[i].stride; }
  const double *p0 = p ? p[0] : nullptr, *p1 = p ? p[1] : nullptr, *p2 = p ? p[2] : nullptr, *p3 = p ? p[3] : nullptr;
  void* args[] = {&a, &n, &first_index, &mask_dev, &p0, &p1, &p2, &p3};
  if (hipModuleLaunchKernel(k.fn, grid_for(n), 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
    throw hip_failure("predicate kernel launch failed: ");
}

static bool is_reserved_name(const std::string& n) {
  static const char* const names[] = {
      "i", "p0", "p1", "p2", "p3",  // the predicate function's own parameters
      "alignas", "alignof", "and", "and_eq", "asm", "auto", "bitand", "bitor", "bool", "break", "case", "catch", "char", "char16_t", "char32_t", "class", "compl", "const",
      "constexpr", "const_cast", "continue", "decltype", "default", "delete", "do", "double", "dynamic_cast", "else", "enum", "explicit", "export", "extern", "false", "float",
      "for", "friend", "goto", "if", "inline", "int", "long", "mutable", "namespace", "new", "noexcept", "not", "not_eq", "nullptr", "operator", "or", "or_eq", "private",
      "protected", "public", "register", "reinterpret_cast", "return", "short", "signed", "sizeof", "static", "static_assert", "static_cast", "struct", "switch", "template",
      "this", "thread_local", "throw", "true", "try", "typedef", "typeid", "typename", "union", "unsigned", "using", "virtual", "void", "volatile", "wchar_t", "while", "xor",
      "xor_eq", "uint8_t", "int8_t", "uint16_t", "int16_t", "uint32_t", "int32_t", "uint64_t", "int64_t", "size_t", "threadIdx", "blockIdx", "blockDim", "gridDim"};
  for (const char* r : names) if (n == r) return true;
  return n.compare(0, 2, "__") == 0 || n.compare(0, 4, "pst_") == 0 || n.compare(0, 3, "Pst") == 0;
}
// `name` followed by '(' somewhere in the text (and not a member access `.name`)
static bool names_a_call(const std::string& expr, const std::string& name) {
  for (size_t p = expr.find(name); p != std::string::npos; p = expr.find(name, p + 1)) {
    const bool starts = p == 0 || !(std::isalnum((unsigned char)expr[p - 1]) || expr[p - 1] == '_' || expr[p - 1] == '.');
    size_t q = p + name.size();
    if (!starts || (q < expr.size() && (std::isalnum((unsigned char)expr[q]) || expr[q] == '_'))) continue;
    while (q < expr.size() && std::isspace((unsigned char)expr[q])) ++q;
    if (q < expr.size() && expr[q] == '(') return true;
  }
  return false;
}
void launch_pred_count(const std::vector<PredAttr>& attrs, const std::string& expr, uint64_t n, uint64_t first_index, uint32_t* counts_dev, const double* const p[4], hipStream_t stream) {
  const pstjit::Kernel k = compile(pred_count_source(attrs, expr), "pst_jit_pred_count", "predicate expression");
  if (n == 0) return;
  PredArgs a{};
  for (size_t i = 0; i < attrs.size(); ++i) { a.base[i] = attrs[i].base; a.stride[i] = attrs[i].stride; }
  const double *p0 = p ? p[0] : nullptr, *p1 = p ? p[1] : nullptr, *p2 = p ? p[2] : nullptr, *p3 = p ? p[3] : nullptr;
  void* args[] = {&a, &n, &first_index, &counts_dev, &p0, &p1, &p2, &p3};
  const uint64_t n_tiles = (n + 2047) / 2048;
  if (hipModuleLaunchKernel(k.fn, (unsigned)((n_tiles + 3) / 4), 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
    throw hip_failure("predicate count kernel launch failed: ");
}

// the attributes of `layout` an expression names (C identifiers only; scalars and Vec3)
std::vector<PredAttr> referenced_attributes(const Layout& layout, const std::string& expr) {
  std::vector<PredAttr> out;
  const std::vector<std::string> ids = identifiers(expr);
  for (const Member& m : layout.members) {
    if (!is_identifier(m.def.name)) continue;
    bool used = false;
    for (const std::string& id : ids) used = used || id == m.def.name;
    if (!used) continue;
    // An attribute becomes a PARAMETER of the predicate function under its own name: names the wrapper or the language already use would end as
    // a duplicate-parameter or keyword error somewhere inside generated code (round-5 advisor finding) -- said here instead.
    if (is_reserved_name(m.def.name))
      throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "predicate expression: the attribute name `" + m.def.name + "` collides with a name the expression language reserves (i, p0 .. p3, "
                                                 "C++ keywords and type names): such an attribute cannot be named in a predicate");
    if (names_a_call(expr, m.def.name))
      throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "predicate expression: `" + m.def.name + "(` -- the name is an attribute of the layout (a value) and is used as a function");
    if (!(m.def.datatype.is_scalar() || m.def.datatype.is_vec3()))
      throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "predicate expression: attribute " + m.def.name + " of datatype " + m.def.datatype.display() + " cannot be named (scalars and Vec3 only)");
    PredAttr a;
    a.name = m.def.name;
    a.ct = (uint32_t)m.def.datatype.comp_type();
    a.ncomp = m.def.datatype.num_components();
    out.push_back(a);
    if (out.size() > kMaxPredAttrs) throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "predicate expression names more than 16 attributes");
  }
  return out;
}

MapSpec map_spec(const DataType& src, const DataType& dst, bool pre, const char* expr) {
  check_text(expr);
  if (!((src.is_scalar() && dst.is_scalar()) || (src.is_vec3() && dst.is_vec3())))
    throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "transformation expression: scalar and Vec3 attributes only (" + src.display() + " -> " + dst.display() + ")");
  MapSpec s;
  s.src_ct = (uint32_t)src.comp_type();
  s.dst_ct = (uint32_t)dst.comp_type();
  s.ncomp = src.num_components();
  s.pre = pre;
  s.expr = expr;
  return s;
}

}  // namespace pstexpr

namespace pst {
uint64_t transform_records_with_expression(const pst_buffer& b, int slot, const std::string& expr, const double* const p[4], hipStream_t stream);  // converter.cpp
// converter.cpp: a mapping whose transformation is an expression (one strided launch per such mapping)
void launch_expression_mapping(const DataType& src_dt, const DataType& dst_dt, bool apply_to_source, const std::string& expr, uint64_t src, uint64_t sstride, uint64_t dst,
                               uint64_t dstride, uint64_t n, uint64_t first_index, hipStream_t stream) {
  const pstexpr::MapSpec s = pstexpr::map_spec(src_dt, dst_dt, apply_to_source, expr.c_str());
  pstexpr::launch_map(s, src, sstride, dst, dstride, n, first_index, nullptr, stream);
}
void validate_expression_mapping(const DataType& src_dt, const DataType& dst_dt, bool apply_to_source, const char* expr) {
  (void)pstexpr::map_source(pstexpr::map_spec(src_dt, dst_dt, apply_to_source, expr));  // shape errors now; syntax errors at the first conversion
}
}  // namespace pst

namespace {
struct TempDev {
  uint8_t* p = nullptr;
  explicit TempDev(size_t bytes) { p = bytes ? dev_alloc(bytes, PST_MEM_DEVICE) : nullptr; }
  ~TempDev() { dev_free(p, PST_MEM_DEVICE); }
};
void copy_out(const std::string& text, char* buf, size_t cap, size_t* needed) {
  if (needed) *needed = text.size() + 1;
  if (buf && cap) {
    const size_t m = std::min(cap - 1, text.size());
    std::memcpy(buf, text.data(), m);
    buf[m] = 0;
  }
}
}  // namespace

extern "C" {

// transform_attribute(attribute, |index, value| expr), point_buffer.rs:391-404, in place on the current stream
int pst_transform_attribute_expr(pst_buffer* b, const char* name, const pst_datatype* dt, const char* expr, const double* const* device_params, size_t n_params) {
  PST_API_BEGIN
  not_null(b, "buffer");
  AttributeDef def{not_null(name, "name"), DataType::from_c(dt)};
  const int slot = b->layout.index_of(def);
  if (slot < 0) throw Error(PST_ERR_MISSING_ATTRIBUTE, "Attribute not found in PointLayout of buffer");
  const pstexpr::MapSpec s = pstexpr::map_spec(def.datatype, def.datatype, false, not_null(expr, "expr"));
  const double* p[4];
  pstexpr::fill_params(device_params, n_params, p);
  ensure_device();
  hipStream_t st = current_stream();
  pstk::reset_plan_kinds();
  const Member& m = b->layout.members[(size_t)slot];
  const uint64_t base = b->columnar ? col_addr(*b, (size_t)slot, 0) : aos_addr(*b, 0) + m.offset;
  const uint64_t stride = b->columnar ? m.size : b->layout.size;
  // interleaved records without padding: the whole tiles through the plan-specialised kernel with the expression inside (one pass over the
  // records instead of a strided read-modify-write of one attribute); the ragged rest -- or everything -- through the expression's own kernel
  const uint64_t done = transform_records_with_expression(*b, slot, s.expr, p, st);
  if (done < b->len) pstexpr::launch_map(s, base + done * stride, stride, base + done * stride, stride, b->len - done, done, p, st);
  stream_sync(st);
  PST_API_END
}

int pst_buffer_filter(const pst_buffer* src, const uint8_t* mask, uint32_t mask_memkind, uint32_t out_storage, pst_buffer** out);

// HashMapBuffer::filter(|index| expr) -> a new buffer of `out_storage` (point_buffer.rs:1064-1075): the predicate is evaluated on the device
// into a byte mask, which the compaction kernels then take like a caller-provided device mask
int pst_buffer_filter_expr(const pst_buffer* src, const char* expr, const double* const* device_params, size_t n_params, uint32_t out_storage, pst_buffer** out) {
  PST_API_BEGIN
  not_null(src, "src");
  not_null(out, "out");
  pstexpr::check_text(not_null(expr, "expr"));
  if (!src->columnar) throw Error(PST_ERR_INVALID_ARGUMENT, "filter is defined on HashMapBuffer (point_buffer.rs:1064)");
  std::vector<pstexpr::PredAttr> attrs = pstexpr::referenced_attributes(src->layout, expr);
  for (pstexpr::PredAttr& a : attrs) {
    const Member* m = src->layout.find_by_name(a.name);
    const size_t slot = (size_t)(m - src->layout.members.data());
    a.base = col_addr(*src, slot, 0);
    a.stride = m->size;
  }
  const double* p[4];
  pstexpr::fill_params(device_params, n_params, p);
  ensure_device();
  hipStream_t st = current_stream();
  if (out_storage > PST_STORAGE_COLUMNAR) throw Error(PST_ERR_INVALID_ARGUMENT, "invalid storage kind");
  // Round 6: the predicate INSIDE the compaction (the reference evaluates the closure in filter's own loop, point_buffer.rs:1064-1136).  Where the
  // layout takes the streaming compaction kernel (filter_stream.hpp: points of at most 64 / 96 bytes), the count pass evaluates the predicate on the
  // columns it names and the scatter pass evaluates it again on the values it holds in registers: no byte mask is written or read, one launch
  // less.  Everything else -- wider points, PST_JIT constraints, PST_EXPR_FUSE=0 (the A/B switch) -- keeps the mask.
  static const bool fuse_env = [] { const char* v = std::getenv("PST_EXPR_FUSE"); return !(v && *v == '0'); }();
  const size_t na = src->layout.members.size();
  std::vector<uint32_t> sizes(na);
  size_t covered_bytes = 0;
  for (size_t a = 0; a < na; ++a) { sizes[a] = (uint32_t)src->layout.members[a].size; covered_bytes += sizes[a]; }
  const bool dst_aos = out_storage != PST_STORAGE_COLUMNAR;
  const uint32_t dst_stride = (uint32_t)src->layout.size;
  if (fuse_env && src->len > 0 && pstk::filter_predicate_streams(sizes.data(), (int)na, dst_aos, dst_stride, covered_bytes == src->layout.size)) {
    const uint64_t n = src->len;
    const uint32_t tile = pstk::filter_tile(dst_aos, dst_stride);
    pstk::FilterPredicate fp;
    fp.function_text = pstexpr::pred_function_text(attrs, expr);
    for (const pstexpr::PredAttr& a : attrs) {
      const Member* m = src->layout.find_by_name(a.name);
      fp.attrs.push_back({(int)(m - src->layout.members.data()), pstexpr::ct_name(a.ct), a.ncomp});
    }
    for (int q = 0; q < 4; ++q) fp.p[q] = p[q];
    // 1. matches per tile, straight from the columns; scan
    uint8_t* scratch = workspace().partials(pstk::filter_workspace_bytes(n));
    pstexpr::launch_pred_count(attrs, expr, n, 0, pstk::filter_counts(scratch, n, tile), p, st);
    const unsigned long long* total_dev = nullptr;
    Workspace& ws = workspace();
    unsigned long long* const host_word = (unsigned long long*)(ws.pinned + 768);
    pstk::launch_filter_scan(n, tile, scratch, &total_dev, st, results_to_host() ? host_word : nullptr);
    if (!results_to_host()) PST_HIP_CHECK(hipMemcpyAsync(host_word, total_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    stream_sync(st);
    const size_t matches = (size_t)*(const unsigned long long*)(ws.pinned + 768);
    // 2. the target, exactly as large as the count says (filter(): count, allocate, filter_into -- :1071-1075)
    auto b = std::make_unique<pst_buffer>();
    b->layout = src->layout;
    b->columnar = !dst_aos;
    if (b->columnar) b->columns.assign(na, nullptr);
    resize_buffer(*b, matches, false);
    if (dst_aos && matches && covered_bytes != src->layout.size) PST_HIP_CHECK(hipMemsetAsync(b->data, 0, matches * src->layout.size, st));  // record padding: VectorBuffer::resize zero-fills (:831-835)
    if (matches) {
      std::vector<uint64_t> src_addr(na), dst_addr(na);
      std::vector<uint32_t> src_stride(na), dst_off(na);
      for (size_t a = 0; a < na; ++a) {
        const Member& m = src->layout.members[a];
        src_addr[a] = col_addr(*src, a, 0);
        src_stride[a] = (uint32_t)m.size;
        dst_addr[a] = b->columnar ? col_addr(*b, a, 0) : 0;
        dst_off[a] = (uint32_t)m.offset;
      }
      // the ragged last tile (less than 2048 points) goes through the gather kernel, which reads a mask: its bytes only
      const uint64_t n_full = n / tile * tile;
      TempDev tail(n - n_full);
      if (n_full < n) {
        std::vector<pstexpr::PredAttr> shifted = attrs;
        for (pstexpr::PredAttr& a : shifted) a.base += n_full * a.stride;
        pstexpr::launch_pred(shifted, expr, n - n_full, n_full, tail.p, p, st);
      }
      std::string err;
      const uint8_t* tail_mask = n_full < n ? tail.p - n_full : nullptr;  // (rebased: mask[i] is point i's byte)
      bool ok = pstk::launch_filter_scatter(tail_mask, n, tile, scratch, matches, src_addr.data(), src_stride.data(), dst_addr.data(), dst_off.data(), sizes.data(), (int)na, dst_aos,
                                            dst_aos ? aos_addr(*b, 0) : 0, dst_stride, covered_bytes == src->layout.size, st, &fp, &err);
      if (!ok) {
        if (err.find("hipRTC compilation failed") != std::string::npos)
          throw Error(PST_ERR_UNSUPPORTED_TRANSFORM, "predicate expression: the compaction kernel with the predicate in it does not compile:\n" + err);
        // no streaming kernel after all (target alignment): the counts stand, the mask is written now and the gather kernels take it
        TempDev mask(n);
        pstexpr::launch_pred(attrs, expr, n, 0, mask.p, p, st);
        ok = pstk::launch_filter_scatter(mask.p, n, tile, scratch, matches, src_addr.data(), src_stride.data(), dst_addr.data(), dst_off.data(), sizes.data(), (int)na, dst_aos,
                                         dst_aos ? aos_addr(*b, 0) : 0, dst_stride, covered_bytes == src->layout.size, st);
        stream_sync(st);
        if (!ok) throw hip_failure("filter launch failed: ");
      }
      stream_sync(st);  // (the tail mask is freed on return)
    }
    *out = b.release();
    return PST_OK;
  }
  TempDev mask(src->len);
  pstexpr::launch_pred(attrs, expr, src->len, 0, mask.p, p, st);
  static uint8_t empty_mask = 0;
  const int rc = pst_buffer_filter(src, src->len ? mask.p : &empty_mask, PST_MEM_DEVICE, out_storage, out);
  stream_sync(st);  // the mask is freed on return
  if (rc != PST_OK) return rc;
  PST_API_END
}

// The translation units the two kinds of expression become (tests compile them without a device through pst_jit_compile_source; also what a
// maintainer reads when a compile error points into generated code).  kind 0: transformation (source / target datatype, apply_to_source),
// kind 1: predicate over `layout` (datatypes ignored).  Returns the text's size in *needed (terminator included).
int pst_expr_source(int kind, const pst_layout* layout, const pst_datatype* src_dt, const pst_datatype* dst_dt, int apply_to_source, const char* expr, char* buf,
                    size_t cap, size_t* needed) {
  PST_API_BEGIN
  pstexpr::check_text(not_null(expr, "expr"));
  std::string text;
  if (kind == 0) {
    text = pstexpr::map_source(pstexpr::map_spec(DataType::from_c(not_null(src_dt, "src_dt")), DataType::from_c(not_null(dst_dt, "dst_dt")), apply_to_source != 0, expr));
  } else if (kind == 1) {
    text = pstexpr::pred_source(pstexpr::referenced_attributes(not_null(layout, "layout")->l, expr), expr);
  } else if (kind == 2) {  // the per-tile count pass of a compaction with the predicate fused (round 6)
    text = pstexpr::pred_count_source(pstexpr::referenced_attributes(not_null(layout, "layout")->l, expr), expr);
  } else if (kind == 3 || kind == 4) {  // the streaming compaction kernel with the predicate inside, into columns (3) / records (4); empty: no such kernel for this layout
    const Layout& l = not_null(layout, "layout")->l;
    const std::vector<pstexpr::PredAttr> attrs = pstexpr::referenced_attributes(l, expr);
    pstk::FilterPredicate fp;
    fp.function_text = pstexpr::pred_function_text(attrs, expr);
    for (const pstexpr::PredAttr& a : attrs) fp.attrs.push_back({(int)(l.find_by_name(a.name) - l.members.data()), pstexpr::ct_name(a.ct), a.ncomp});
    std::vector<uint32_t> sizes;
    size_t cov = 0;
    for (const Member& m : l.members) { sizes.push_back((uint32_t)m.size); cov += m.size; }
    text = pstk::filter_stream_source(sizes.data(), (int)sizes.size(), kind == 4, (uint32_t)l.size, cov == l.size, &fp);
  } else {
    throw Error(PST_ERR_INVALID_ARGUMENT, "pst_expr_source: kind must be 0 (transformation), 1 (predicate -> byte mask), 2 (predicate -> matches per tile), 3 / 4 (compaction with the predicate inside)");
  }
  copy_out(text, buf, cap, needed);
  PST_API_END
}

}  // extern "C"
