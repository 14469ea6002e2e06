/*
 * pasture_amd — C ABI of the MI355X-native implementation of pasture's per-point attribute-transform hot path.
 *
 * The reference (igd-geo/pasture, Rust) has no FFI on this path; its seam is the Rust trait surface
 * (BorrowedBuffer / InterleavedBuffer / ColumnarBuffer / PointLayout) plus the entry points listed below.  Every
 * function here names the reference interface it replaces (paths relative to the reference checkout).  A Rust
 * shim that re-implements those traits on top of this ABI is sketched in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns a pst_status; 0 = ok.  The reference signals precondition violations with
 *    panic!/assert!/expect; nothing unwinds across this ABI — the shim re-raises codes 2..14 as panic!.
 *    pst_last_error() returns the panic message of the last failing call on the calling thread.
 *  - plain pointers and sizes only; no torch / C++ types.  Device pointers are HIP device addresses.
 *  - all device work is enqueued on the calling thread's current stream (pst_set_stream; default = the null
 *    stream).  Functions that return results to host memory synchronise that stream before returning; the
 *    *_async variants do not.
 *  - there is NO CPU fallback: compute entry points fail with PST_ERR_NO_DEVICE / PST_ERR_HIP when no gfx950
 *    device is usable.  Host-only logic (layouts, converter mapping construction, argument checks) works
 *    without a device.
 */
#ifndef PASTURE_AMD_H
#define PASTURE_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pst_status {
  PST_OK = 0,
  PST_ERR_INVALID_ARGUMENT = 1,
  PST_ERR_LAYOUT_MISMATCH = 2,         /* buffer_conversion.rs:302-303 assert_eq! on layouts */
  PST_ERR_RANGE = 3,                   /* buffer_conversion.rs:304-306 range asserts, slice bounds */
  PST_ERR_MISSING_ATTRIBUTE = 4,       /* .expect("... not found in ... PointLayout") */
  PST_ERR_INVALID_CONVERSION = 5,      /* attribute_conversion.rs:267-269 "Invalid conversion X -> Y" */
  PST_ERR_TRANSFORM_TYPE_MISMATCH = 6, /* buffer_conversion.rs:209-213 */
  PST_ERR_UNSUPPORTED_TRANSFORM = 7,   /* closure outside the closed descriptor set: use the CPU path */
  PST_ERR_DUPLICATE_ATTRIBUTE = 8,     /* point_layout.rs:783-788 */
  PST_ERR_INVALID_LAYOUT = 9,          /* point_layout.rs:725-745, Layout::from_size_align failures */
  PST_ERR_BOUNDS_INVALID = 10,         /* math/bounds.rs:21-26 AABB::from_min_max panic (e.g. all-NaN input) */
  PST_ERR_TOO_FEW_POINTS = 11,         /* normal_estimation.rs:86-88 */
  PST_ERR_K_TOO_SMALL = 12,            /* normal_estimation.rs:89-91 */
  PST_ERR_NOT_ENOUGH_NEIGHBOURS = 13,  /* normal_estimation.rs:471 unwrap on Err */
  PST_ERR_UNSUPPORTED_ATTRIBUTE = 14,  /* voxel_grid.rs:475 "Waveform data currently not supported!", :683 non-standard attribute */
  PST_ERR_HIP = 20,
  PST_ERR_NO_DEVICE = 21,
  PST_ERR_OUT_OF_MEMORY = 22,
  PST_ERR_UNSUPPORTED = 23
} pst_status;

/* PointAttributeDataType, point_layout.rs:23-68; kind = declaration order :25-50 */
enum {
  PST_U8 = 0, PST_I8, PST_U16, PST_I16, PST_U32, PST_I32, PST_U64, PST_I64, PST_F32, PST_F64,
  PST_VEC3U8, PST_VEC3U16, PST_VEC3F32, PST_VEC3I32, PST_VEC3F64, PST_VEC4U8, PST_BYTEARRAY, PST_CUSTOM
};
typedef struct pst_datatype {
  uint32_t kind;
  uint32_t reserved;
  uint64_t size_param;  /* ByteArray(length) / Custom.size */
  uint64_t align_param; /* Custom.min_alignment */
  uint8_t uuid[16];     /* Custom.name */
} pst_datatype;

/* PointAttributeMember, point_layout.rs:353-431 */
typedef struct pst_member {
  const char* name; /* borrowed from the layout; valid until the layout is destroyed or mutated */
  pst_datatype datatype;
  uint64_t offset;
  uint64_t size;
} pst_member;

/* Closed set of attribute transformations standing in for the reference's `Fn(T) -> T` closures
 * (buffer_conversion.rs:14-31, :194-234).  Anything else must stay on the CPU path. */
enum { PST_XF_NONE = 0, PST_XF_AFFINE = 1, PST_XF_BITFIELD = 2 };
typedef struct pst_transform {
  uint32_t kind;
  uint32_t shift;        /* BITFIELD: (v >> shift) & mask on U8/U16/U32/U64 — raw_readers.rs:61-164 */
  pst_datatype datatype; /* the closure's T; checked like buffer_conversion.rs:209-213 */
  double scale[3];       /* AFFINE on F64/Vec3f64: (p*scale)+offset, two roundings, never fused — raw_readers.rs:42-48; */
  double offset[3];      /*        on F32/Vec3f32: ((p as f64*scale)+offset) as f32 — raw_readers.rs:49-55            */
  uint64_t mask;
} pst_transform;

typedef struct pst_mapping_info { /* AttributeMapping, buffer_conversion.rs:41-55 (introspection) */
  const char* source_name;
  const char* target_name;
  pst_datatype source_datatype;
  pst_datatype target_datatype;
  uint64_t source_offset;
  uint64_t target_offset;
  int32_t has_converter;
  uint32_t transform_kind;
  int32_t apply_to_source;
  int32_t reserved;
} pst_mapping_info;

typedef struct pst_layout pst_layout;       /* PointLayout,           point_layout.rs:648-997 */
typedef struct pst_buffer pst_buffer;       /* VectorBuffer :659 / HashMapBuffer :1031 / ExternalMemoryBuffer :1479 (point_buffer.rs) */
typedef struct pst_converter pst_converter;
typedef struct pst_point_converter pst_point_converter;
typedef struct pst_comm pst_comm;           /* RCCL communicator(s) of the sharded path */
typedef struct pst_comm_id { uint8_t bytes[128]; } pst_comm_id;  /* rendezvous token (RCCL unique id), shared out of band */ /* BufferLayoutConverter, buffer_conversion.rs:98-663 */

const char* pst_last_error(void);

/* ---- device / stream ------------------------------------------------------------------------------- */
int pst_device_count(int* out);
int pst_set_device(int device);
int pst_set_stream(void* hip_stream); /* hipStream_t; thread-local; NULL = null stream */
int pst_get_stream(void** out_hip_stream); /* the calling thread's current stream (so that a helper can restore it) */
int pst_stream_synchronize(void);
/* Frees the device scratch the calling thread's pst_compute_normals* calls keep between calls, per device (about 55 bytes per point of the
 * largest recent cloud, never more than PST_SCRATCH_MAX_BYTES = 16 GiB by default; the reference's temporaries die with every call: normal_estimation.rs:79-130),
 * and hands the unused blocks of the current device's stream-ordered pool -- the memory of destroyed / shrunk buffers, which the library keeps for its next
 * allocation -- back to the driver, where other allocators of the process (hipMalloc) can get at them.  Synchronises the device.  Never needed for correctness:
 * an allocation of the library that fails for lack of memory does the same and tries once more before it reports PST_ERR_OUT_OF_MEMORY. */
int pst_release_scratch(void);
/* The PST_KNN_* / PST_SCRATCH_MAX_BYTES tuning switches are read from the environment ONCE (first use) -- never on the call path.  This reads them
 * again: for test and A/B harnesses that change them inside one process.  Not to be called while another thread is inside the library. */
int pst_reload_tuning(void);

/* ---- PointLayout ------------------------------------------------------------------------------------ */
int pst_layout_create(pst_layout** out);                           /* PointLayout::default()            :1011-1023 */
int pst_layout_destroy(pst_layout* l);
int pst_layout_clone(const pst_layout* l, pst_layout** out);
/* add_attribute(attr, FieldAlignment::{Default | Packed(max_alignment)})                               :778-822  */
int pst_layout_add_attribute(pst_layout* l, const char* name, const pst_datatype* dt, uint32_t packed, uint64_t max_alignment);
int pst_layout_from_members(const pst_member* members, size_t n, uint64_t type_alignment, pst_layout** out); /* :719-759 */
int pst_layout_num_attributes(const pst_layout* l, size_t* out);
int pst_layout_get_member(const pst_layout* l, size_t index, pst_member* out);                           /* at() :898-900 */
int pst_layout_size_of_point_entry(const pst_layout* l, uint64_t* out);                                  /* :928-931 */
int pst_layout_alignment(const pst_layout* l, uint64_t* out);
int pst_layout_equals(const pst_layout* a, const pst_layout* b, int* out);                                /* derive(PartialEq) :646 */

/* ---- buffers ------------------------------------------------------------------------------------------ */
enum { PST_STORAGE_INTERLEAVED = 0, PST_STORAGE_COLUMNAR = 1 };
enum { PST_MEM_DEVICE = 0, PST_MEM_PINNED_HOST = 1 };
/* MakeBufferFromLayout::new_from_layout (:497-500): empty buffer, library-owned memory of `memkind`.
 * Interleaved: one allocation, stride = size_of_point_entry.  Columnar: one 256-B aligned column per attribute. */
int pst_buffer_create(const pst_layout* l, uint32_t storage, uint32_t memkind, pst_buffer** out);
/* ExternalMemoryBuffer<T: AsRef<[u8]>> (:1479-1708): interleaved view over caller-owned device-accessible memory;
 * len = nbytes / size_of_point_entry; nbytes must be a multiple of the point size (:1488-1497).  Any base address (records behind a file header).
 * "Device-accessible" is checked (both ends of the range, hipPointerGetAttributes): device memory, managed memory, or host memory the device maps
 * (hipHostMalloc / hipHostRegister); ordinary host memory -- what the reference's type wraps -- is PST_ERR_INVALID_ARGUMENT here instead of a GPU fault
 * at the first kernel.  PST_EXTERNAL_UNCHECKED=1 skips the check.  The same holds for every column of pst_buffer_wrap_external_columns. */
int pst_buffer_wrap_external(const pst_layout* l, void* device_ptr, size_t nbytes, pst_buffer** out);
/* Columnar view over caller-owned columns (one device pointer per layout attribute, layout order), `len` points. */
int pst_buffer_wrap_external_columns(const pst_layout* l, void* const* column_ptrs, size_t len, pst_buffer** out);
/* SliceBuffer::slice / SliceBufferMut::slice_mut, pasture-core/src/containers/slice.rs:16-43 (BufferSlice :76-330): a non-owning view of
 * points [first, first + count) of `parent` with the parent's layout and storage kind.  Every entry point that takes a buffer takes a
 * slice: calculate_bounds(&buf.slice(a..b)), the chunked minmax_attribute of pasture-tools/src/bin/info.rs:66-78, transform_attribute on
 * slice_mut, compute_normals, conversions from / into it.  A range outside the parent is PST_ERR_RANGE (the slice's index assertions
 * :52-75); a slice cannot be resized (PST_ERR_UNSUPPORTED: it is not an OwningBuffer).  It borrows the parent's memory: destroy it
 * before the parent is resized or destroyed (what the borrow checker enforces in Rust).  The library checks what the borrow checker would:
 * every owning buffer carries a storage epoch that its resize and destroy bump, and any later use of a slice cut before that (or of a slice
 * of such a slice) is PST_ERR_INVALID_ARGUMENT -- never a read of freed device memory.  (pst_buffer_destroy of the stale slice itself
 * still succeeds.  Slices of caller-owned external memory are not tracked: that memory's lifetime is the caller's.) */
int pst_buffer_slice(const pst_buffer* parent, size_t first, size_t count, pst_buffer** out);
int pst_buffer_destroy(pst_buffer* b);
int pst_buffer_len(const pst_buffer* b, size_t* out);                 /* BorrowedBuffer::len :29 */
int pst_buffer_resize(pst_buffer* b, size_t count);                   /* OwningBuffer::resize :263 — new points zero-filled */
/* BorrowedMutBuffer::swap (point_buffer.rs:229; VectorBuffer :770-783, HashMapBuffer :1276-1292, ExternalMemoryBuffer :1591-1612): exchanges two
 * points in place, on the current stream (three small device copies per record / per column through the library's scratch).  Either index out of
 * bounds -> PST_ERR_RANGE (the reference's assert!); equal indices return at once.  Not a bulk path: a per-point device round trip. */
int pst_buffer_swap(pst_buffer* b, size_t from_index, size_t to_index);
int pst_buffer_is_columnar(const pst_buffer* b, int* out);            /* as_columnar / as_interleaved probes :143-151 */
int pst_buffer_layout(const pst_buffer* b, pst_layout** out_clone);   /* point_layout() :33 (returns a clone) */
int pst_buffer_points_ptr(const pst_buffer* b, void** out);           /* get_point_range_ref(0..len).as_ptr() :524-526 */
int pst_buffer_column_ptr(const pst_buffer* b, const char* name, const pst_datatype* dt, void** out); /* get_attribute_range_ref :593-599 */
/* host <-> buffer transfers (synchronous) */
int pst_buffer_write_points(pst_buffer* b, size_t first, size_t count, const void* host_src);  /* set_point_range :90 */
int pst_buffer_read_points(const pst_buffer* b, size_t first, size_t count, void* host_dst);   /* get_point_range :45 */
int pst_buffer_write_attribute(pst_buffer* b, const char* name, const pst_datatype* dt, size_t first, size_t count, const void* host_src); /* set_attribute_range :110 */
int pst_buffer_read_attribute(const pst_buffer* b, const char* name, const pst_datatype* dt, size_t first, size_t count, void* host_dst);  /* get_attribute_range :71 */
/* view_attribute_with_conversion::<T>(attribute).into_iter().collect(), point_buffer.rs:322-330 / buffer_views.rs:533-650: the attribute `name`
 * of points [first, first + count) converted from its stored datatype to `target_dt` with the Rust-`as` table (convert_unit when they are
 * equal).  Not in the layout -> PST_ERR_MISSING_ATTRIBUTE (:549-552); no conversion between the datatypes -> PST_ERR_INVALID_CONVERSION
 * ("Conversion between attribute types is impossible", :553-561).  _device: the dense array of `target_dt` values lands in device memory,
 * stream-ordered. */
int pst_buffer_read_attribute_converted(const pst_buffer* b, const char* name, const pst_datatype* target_dt, size_t first, size_t count, void* host_dst);
int pst_buffer_read_attribute_converted_device(const pst_buffer* b, const char* name, const pst_datatype* target_dt, size_t first, size_t count, void* device_dst);
/* deterministic synthetic points generated on the device (DESIGN.md "Synthetic inputs") */
int pst_buffer_synth_fill(pst_buffer* b, uint64_t seed, uint64_t first_index);

/* OwningBufferExt::append (point_buffer.rs:419-489): appends every point of `other`; the layouts must be equal
 * (PST_ERR_LAYOUT_MISMATCH = the assert_eq! of :420).  Capacity grows geometrically like Vec. */
int pst_buffer_append(pst_buffer* self, const pst_buffer* other);
/* HashMapBuffer::filter_into (point_buffer.rs:1082-1136): order-preserving compaction of the points whose predicate holds into
 * dst[0, matches).  The predicate `Fn(usize) -> bool` is passed as a byte mask of src->len entries (mask[i] != 0 keeps
 * point i; benches/buffer_filter_bench.rs:62-64); mask_memkind = PST_MEM_DEVICE for a device pointer, anything else = host.
 * num_matches_hint < 0 = None.  Panics of the reference: layouts differ -> PST_ERR_LAYOUT_MISMATCH; dst shorter than
 * num_matches -> PST_ERR_RANGE; more matches than the hint -> PST_ERR_RANGE (slice index out of range).
 * *out_matches (optional) = number of mask hits. */
int pst_buffer_filter_into(const pst_buffer* src, pst_buffer* dst, const uint8_t* mask, uint32_t mask_memkind, int64_t num_matches_hint,
                           size_t* out_matches);
/* stream-ordered variant for callers that know the count (`Some(num_matches)`, as the reference's bench passes it,
 * benches/buffer_filter_bench.rs:62-74): count, scan and copies are enqueued on the current stream and the call returns; no host
 * synchronisation, no panic mapping.  The mask is a device pointer.  At most num_matches points are written; the number of mask hits
 * is copied to *device_count_out (device-accessible memory, optional) in stream order -- a value above num_matches is the
 * reference's slice panic (:1103-1108) and the caller's to check.  dst->len < num_matches -> PST_ERR_RANGE (checked on the host). */
int pst_buffer_filter_into_async(const pst_buffer* src, pst_buffer* dst, const uint8_t* device_mask, size_t num_matches, uint64_t* device_count_out);
/* HashMapBuffer::filter::<B, _> (point_buffer.rs:1064-1076): new buffer of out_storage holding exactly the matching points */
int pst_buffer_filter(const pst_buffer* src, const uint8_t* mask, uint32_t mask_memkind, uint32_t out_storage, pst_buffer** out);

/* ---- device expressions: user-written closures as source text (round 5) -----------------------------------------------------------------
 * The reference takes ANY closure: Fn(T) -> T transformations (buffer_conversion.rs:13-36, 194-234), transform_attribute's Fn(usize, T) -> T
 * (point_buffer.rs:391-404) and filter's Fn(usize) -> bool (point_buffer.rs:1064-1136).  A closure cannot cross a C ABI into a kernel; its
 * source text can.  An expression is C++ EXPRESSION syntax (no statements, no braces, no string literals) over these names, compiled at run
 * time with hipRTC (cached per text; -ffp-contract=off: `v * s + o` keeps its two roundings) and run as a device function:
 *   transformations:  v = this component of the value (of T's component type; the value itself for scalars),  x y z = the components of a
 *                     Vec3 value,  c = the component being computed (0 1 2),  i = the point's index (the closure's usize),  p0 .. p3 = device
 *                     arrays of double given with the call (what a closure would capture).  One expression serves every component; a Vec3
 *                     attribute may give three, `x-expr ; y-expr ; z-expr`.  The result is converted to T's component type with Rust `as`.
 *   predicates:       every attribute of the buffer's layout whose name is a C identifier -- scalars by value, Vec3 as .x .y .z --, i, p0 .. p3:
 *                     "Classification == 2 && Position3D.z < 120.0".
 * Results are those of the same arithmetic on the host, bit for bit, EXCEPT which NaN an operation returns: sign and payload of a NaN result
 * are unspecified in Rust and differ between gfx950 and x86 (`y - z` with a NaN z: the GPU evaluates y + (-z) and returns the NaN with its sign
 * flipped); a NaN is a NaN on both.
 * Scalar and Vec3 attributes only.  A text that does not compile is PST_ERR_UNSUPPORTED_TRANSFORM with the compiler's log in pst_last_error;
 * PST_JIT=0 (no run-time compiler) makes every expression PST_ERR_UNSUPPORTED_TRANSFORM.  The closed descriptors (pst_transform) remain the
 * fast path for the in-tree callers' closures; an expression mapping is its own strided launch. */
/* BufferLayoutConverter::set_custom_mapping_with_transformation (buffer_conversion.rs:194-234) with the closure as an expression; T = the
 * source attribute's datatype when apply_to_source (the conversion follows), the target's otherwise (the conversion precedes) :209-213 */
int pst_converter_set_custom_mapping_with_expression(pst_converter* c, const char* from_name, const pst_datatype* from_dt, const char* to_name,
                                                     const pst_datatype* to_dt, const char* expr, int apply_to_source);
/* BorrowedMutBufferExt::transform_attribute(attribute, |index, value| expr), point_buffer.rs:391-404, in place; device_params = up to four
 * device arrays of double the expression names p0 .. p3 (NULL / 0 for none) */
int pst_transform_attribute_expr(pst_buffer* b, const char* name, const pst_datatype* dt, const char* expr, const double* const* device_params, size_t n_params);
/* HashMapBuffer::filter(|index| expr) -> new buffer of out_storage holding exactly the matching points (point_buffer.rs:1064-1076) */
int pst_buffer_filter_expr(const pst_buffer* src, const char* expr, const double* const* device_params, size_t n_params, uint32_t out_storage, pst_buffer** out);
/* the translation unit an expression becomes (kind 0: transformation between src_dt / dst_dt; kind 1: predicate over `layout`): what a compile
 * error's line numbers refer to, and what the CPU tests hand to pst_jit_compile_source.  *needed = its size, terminator included. */
int pst_expr_source(int kind, const pst_layout* layout, const pst_datatype* src_dt, const pst_datatype* dst_dt, int apply_to_source, const char* expr, char* buf,
                    size_t cap, size_t* needed);

/* RawPointConverter::{from_to, convert}, pasture-core/src/layout/conversion/attribute_conversion.rs:62-109 — the point-major variant:
 * one `as` converter per attribute present in BOTH layouts (matched by name, in the order of `from`) whose datatypes differ.
 * Attributes with equal datatypes get no converter and are SKIPPED, not copied (:73-90); an impossible pair is the panic of :267-269
 * (PST_ERR_INVALID_CONVERSION).  `convert` runs the converters on `count` interleaved points (the reference converts one point slice
 * per call); bytes of the target points that no converter writes keep their values.  Both buffers must be interleaved and carry the
 * layouts given to create (the reference's `unsafe` contract, checked here: PST_ERR_LAYOUT_MISMATCH). */
int pst_point_converter_create(const pst_layout* from, const pst_layout* to, pst_point_converter** out);
int pst_point_converter_destroy(pst_point_converter* c);
int pst_point_converter_num_converters(const pst_point_converter* c, size_t* out);
int pst_point_converter_convert(const pst_point_converter* c, const pst_buffer* src, size_t src_first, pst_buffer* dst, size_t dst_first, size_t count);
/* ---- BufferLayoutConverter ------------------------------------------------------------------------- */
int pst_converter_create(const pst_layout* from, const pst_layout* to, int with_default, pst_converter** out); /* for_layouts :112 / for_layouts_with_default :126 */
int pst_converter_destroy(pst_converter* c);
int pst_converter_set_custom_mapping(pst_converter* c, const char* from_name, const pst_datatype* from_dt, const char* to_name,
                                     const pst_datatype* to_dt);                                       /* :156-183 */
int pst_converter_set_custom_mapping_with_transformation(pst_converter* c, const char* from_name, const pst_datatype* from_dt,
                                                         const char* to_name, const pst_datatype* to_dt, const pst_transform* xf,
                                                         int apply_to_source);                         /* :194-234 */
int pst_converter_num_mappings(const pst_converter* c, size_t* out);
int pst_converter_get_mapping(const pst_converter* c, size_t index, pst_mapping_info* out);
/* convert_into_range :292-359 (convert_into :268 = full ranges). Enqueues on the current stream, then synchronises. */
int pst_converter_convert_into_range(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1);
int pst_converter_convert_into_range_async(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1);
/* convert :242-259 — allocates the target (new_from_layout + resize) and converts into it */
int pst_converter_convert(const pst_converter* c, pst_buffer* src, uint32_t out_storage, pst_buffer** out);
/* convert_into_range followed by calculate_bounds(target range) in ONE pass over HBM when the mapping set allows
 * (columnar Vec3f64 POSITION_3D target); identical results to the two separate calls. */
int pst_converter_convert_into_range_with_bounds(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst,
                                                 size_t t0, size_t t1, double out_min[3], double out_max[3], int* has_value);
/* stream-ordered variant: the result record {min[3], max[3]} (6 doubles; seeds +/-f64::MAX if the range is empty)
 * is written to `device_out6` (device-accessible memory); no host synchronisation, no panic mapping. */
int pst_converter_convert_into_range_with_bounds_async(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst,
                                                       size_t t0, size_t t1, double* device_out6);

/* ---- plan specialisation ------------------------------------------------------------------------------
 * The reference's converter is layout-generic (buffer_conversion.rs:112-234) and walks its Vec<AttributeMapping> at run time.  Here a
 * conversion takes one of these kernel families; pst_last_plan_kinds reports (one bit per family, 1u << PST_PLAN_*) which ones the calling
 * thread's last pst_converter_convert* call launched.  PST_PLAN_JIT: the mapping list compiled into the kernel as a constant -- hipRTC at run
 * time, cached per plan in memory and on disk (PST_JIT = 0 | async (default: compiled on a background thread once a call of at least
 * PST_JIT_MIN_POINTS points has shown the plan, interpreted until then) | sync; PST_JIT_CACHE_DIR).  Results are identical whichever
 * family runs. */
enum {
  PST_PLAN_NONE = 0,
  PST_PLAN_INTERPRETED = 1, /* generic LDS-tile kernels interpreting the mapping list */
  PST_PLAN_JIT = 2,         /* the plan as a compile-time constant, compiled at run time */
  PST_PLAN_STATIC = 3,      /* the same kernels instantiated in-tree for the layouts of the reference's own benches */
  PST_PLAN_LAS = 4,         /* LAS-format-specialised decoder / transposer (raw_readers.rs:31-167, las_types.rs) */
  PST_PLAN_STREAM = 5,      /* columnar Vec3f64 stream kernel (copy / affine / AABB) */
  PST_PLAN_COLUMN = 6,      /* one wide-vector launch per columnar -> columnar mapping */
  PST_PLAN_COPY = 7,        /* identity between equal packed layouts: one byte copy of the records */
  PST_PLAN_DIRECT = 8,      /* strided fall-back without LDS staging (records too large for a tile) */
  PST_PLAN_EXPRESSION = 9   /* an expression mapping's OWN strided pass (one launch per mapping: columns -> columns, ragged tails, plans without a
                               specialised form); an expression that runs INSIDE the plan-specialised kernel reports PST_PLAN_JIT only */
};
int pst_last_plan_kinds(uint32_t* mask);
/* Compiles (or fetches from the cache) the specialised kernel for conversions between buffers of these storage kinds NOW, so that the first
 * call already takes it.  *plan_kind: the family such a call will use (PST_PLAN_JIT when a specialised kernel is ready). */
int pst_converter_prepare(const pst_converter* c, int src_columnar, int dst_columnar, int with_bounds, uint32_t* plan_kind);
/* Two families can serve LAS-shaped plans (typed LasPointFormatN records -> columns and columns -> records; raw LAS records -> typed records): the
 * format-specialised LAS kernels and the plan-specialised kernel.  Which one is faster differs from box to box by a few per cent, so the
 * converter MEASURES it once per storage pairing on the first SYNCHRONOUS conversion of at least 2^22 points (convert / convert_into_range /
 * ..._with_bounds; both families run on the caller's range -- they write the same bytes --, median of three timed passes each, one host wait; not
 * while the stream is being captured, not when source and target memory overlap; PST_FAMILY_AUTOTUNE=0: never, plan-specialised first) and
 * keeps the winner.  The stream-ordered `_async` entry points never measure (they neither block the host nor repeat the caller's conversion):
 * callers of those run pst_converter_measure_families once, before their loop.
 * dst_columnar: 0 = records from records, 1 = columns (from records), 2 = records from columns.  *choice: -1 not measured yet, 0 = PST_PLAN_LAS,
 * 1 = plan-specialised (PST_PLAN_STATIC / PST_PLAN_JIT), 2 = the plan has no second family; ms2 (optional): milliseconds per pass the
 * measurement saw for {LAS, plan-specialised}.  (No reference counterpart: the reference has one loop, buffer_conversion.rs:292-359.) */
int pst_converter_family_choice(const pst_converter* c, int dst_columnar, int with_bounds, int* choice, float ms2[2]);
int pst_converter_measure_families(const pst_converter* c, pst_buffer* src, size_t s0, size_t s1, pst_buffer* dst, size_t t0, size_t t1, int with_bounds);
/* Introspection of the run-time compiler (tests, tools): the translation unit generated for a converter (empty when the plan takes another
 * family; *needed = bytes incl. the terminator); compilation of a translation unit against the embedded device headers for gfx950 WITHOUT a
 * device (the code object is copied to code_buf when given; *code_bytes = its size); counters. */
int pst_converter_jit_source(const pst_converter* c, int src_columnar, int dst_columnar, int with_bounds, char* buf, size_t cap, size_t* needed);
int pst_jit_compile_source(const char* source, void* code_buf, size_t code_cap, size_t* code_bytes, char* log, size_t log_cap);
typedef struct pst_jit_stats {
  uint64_t compiled, disk_hits, memory_hits, failures, launches;
  double compile_seconds;
} pst_jit_stats;
int pst_jit_get_stats(pst_jit_stats* out);
int pst_jit_set_mode(int mode); /* -1: back to the PST_JIT environment setting; 0 off; 1 async; 2 sync (tests, A/B harnesses) */

/* ---- pasture-algorithms loops --------------------------------------------------------------------- */
/* calculate_bounds, pasture-algorithms/src/bounds.rs:11-85.  has_value = 0 <=> None. */
int pst_calculate_bounds(const pst_buffer* b, double out_min[3], double out_max[3], int* has_value);
int pst_calculate_bounds_async(const pst_buffer* b, double* device_out6);
/* minmax_attribute::<T>, pasture-algorithms/src/minmax.rs:13-51 with T = `dt` (must be the stored datatype) */
int pst_minmax_attribute(const pst_buffer* b, const char* name, const pst_datatype* dt, void* out_min, void* out_max, int* has_value);
/* BorrowedMutBufferExt::transform_attribute, point_buffer.rs:391-404, with a closed-set transformation */
int pst_transform_attribute(pst_buffer* b, const char* name, const pst_datatype* dt, const pst_transform* xf);
/* compute_centroid, pasture-algorithms/src/normal_estimation.rs:198-237: mean of Position3D (Vec3f64) over all points, or over the
 * finite ones when some coordinate is NaN (is_dense :133-140).  Empty buffer -> PST_ERR_TOO_FEW_POINTS ("The point cloud is empty!").
 * A parallel sum: equal to the reference's left-to-right sum within rounding (1e-9 relative), not bit for bit. */
int pst_compute_centroid(const pst_buffer* b, double out_centroid[3]);
/* compute_normals, pasture-algorithms/src/normal_estimation.rs:79-130: per point (normal[3] f64, curvature f64) to
 * host arrays; out_knn (nullable, n*k int64, -1 padded) receives the neighbour indices in ascending distance. */
int pst_compute_normals(const pst_buffer* b, size_t k, double* out_normals, double* out_curvature, int64_t* out_knn);
/* device-resident variant: writes the NORMAL attribute (Vec3f32, point_layout.rs:594-597; f64 -> f32 `as` narrowing)
 * and an F64 "Curvature" attribute of `dst` (columnar or interleaved, same length) without leaving HBM. */
int pst_compute_normals_into(const pst_buffer* b, size_t k, pst_buffer* dst);
/* the same result as pst_compute_normals in caller-owned DEVICE memory (each pointer nullable, at least one given): normals f64 [n][3],
 * curvature f64 [n], neighbour lists uint32 [n][k] in ascending distance (normal_estimation.rs:103-108; 0xFFFFFFFF pads clouds of
 * fewer than k points).  10^8 points, k = 16: 2.4 GB + 0.8 GB + 6.4 GB. */
int pst_compute_normals_device(const pst_buffer* b, size_t k, double* d_normals, double* d_curvature, uint32_t* d_knn);
/* Stream-ordered compute_normals (round 4).  The synchronous call measures the cloud between its kernels (bounds, occupancy, local scale, a
 * probe and a census of the built index) and reads five to eight values back; pst_compute_normals_plan_create runs it ONCE (dst holds the
 * result) and keeps what it decided -- the grid's frame, box and cell edges, the LDS box shape, the capacities of the occupied-box list and
 * of the hand-back list.  pst_compute_normals_into_async then runs keys -> sort -> permutation -> directory -> box search -> exact search of
 * the hand-backs on the current stream with no host round trip and no allocation (hipGraph-capturable), on this cloud or on ANOTHER cloud of
 * the same length: the grid is fixed by the plan, points outside it are clamped into boundary cells and their queries go to the exact
 * search, so the result is exact for any data.  device_status2[0] = 0 when it is complete; bit 0: the number of finite points differs from
 * the plan's, bit 1 / bit 2: more occupied boxes / more hand-backs than the plan provided for, bit 3: the exact search handed queries back
 * (far points: the coarser levels are host-driven), bit 4: degenerate neighbourhoods ([1] = how many: PST_ERR_NOT_ENOUGH_NEIGHBOURS of the
 * synchronous call) -- with any of bits 0-3 set the caller falls back to pst_compute_normals_into.  Clouds whose synchronous call did not
 * take the box search or left queries open (fewer than 2^20 points, far outliers) have no plan: PST_ERR_UNSUPPORTED. */
typedef struct pst_normals_plan pst_normals_plan;
int pst_compute_normals_plan_create(const pst_buffer* b, size_t k, pst_buffer* dst, pst_normals_plan** out);
int pst_normals_plan_destroy(pst_normals_plan* plan);
int pst_compute_normals_into_async(pst_normals_plan* plan, const pst_buffer* b, pst_buffer* dst, uint64_t* device_status2);
/* voxelgrid_filter, pasture-algorithms/src/voxel_grid.rs:109-166: one centroid point per occupied voxel (cells centred on the
 * axis markers min + k*leafsize, find_leaf :21-52), appended to `filtered` in (x, y, z) voxel order.  Reductions per attribute
 * of filtered's layout (set_all_attributes :459-689): average (Position3D, ColorRGB, Normal, Intensity, NIR; sequential f64
 * sums in point order), most common (return fields, classification, scan angle, user data, point source id; ties: smallest
 * value — the reference's HashMap order is random), max-pool from 0.0 (ClassificationFlags, GpsTime, PointID).
 * Panics: no Position3D -> PST_ERR_MISSING_ATTRIBUTE; empty buffer -> PST_ERR_BOUNDS_INVALID (unwrap on None); waveform or
 * non-standard attribute in filtered's layout -> PST_ERR_UNSUPPORTED_ATTRIBUTE; attribute missing in `buffer` -> PST_ERR_MISSING_ATTRIBUTE.
 * filtered's length is final on return; the attribute reductions are enqueued on the current stream (no final synchronisation). */
int pst_voxelgrid_filter(const pst_buffer* buffer, double leafsize_x, double leafsize_y, double leafsize_z, pst_buffer* filtered);

/* Stream-ordered voxelgrid_filter (round 4).  pst_voxelgrid_filter reads three things back between its kernels (the bounds for the axis
 * markers, the voxel count for the output, the count of very large voxels): on a busy host each round trip costs more than the kernels.
 * pst_voxelgrid_plan_create runs ONE synchronous pass over `buffer` and allocates every scratch buffer for capacities derived from it (points
 * = len(buffer); an eighth more axis markers, a quarter more occupied voxels: *max_voxels).  pst_voxelgrid_filter_async then enqueues
 * calculate_bounds -> markers (create_markers_for_axis :55-83 by one lane: the same sequential additions) -> keys -> sort -> run heads ->
 * reductions on the current stream with NO host synchronisation and NO allocation (hipGraph-capturable); one centroid per occupied voxel
 * lands in filtered[dst_first ..) (filtered must already hold dst_first + max_voxels points; the tail behind the count is not written),
 * device_count_and_status[0] = number of voxels, [1] = status: 0 = the result is the reference's; bit 0 bounds invalid (an all-NaN
 * component: the reference panics), bit 1 more axis markers than planned, bit 2 more voxels than planned, bit 3 a leaf size that does not
 * advance the markers -- with any bit set nothing useful was written and the caller falls back to pst_voxelgrid_filter.  The buffer may be
 * ANOTHER cloud of the same length (the plan fixes capacities, not contents).  One call per plan in flight at a time. */
typedef struct pst_voxel_plan pst_voxel_plan;
int pst_voxelgrid_plan_create(const pst_buffer* buffer, double leafsize_x, double leafsize_y, double leafsize_z, pst_voxel_plan** out, size_t* max_voxels);
int pst_voxelgrid_plan_destroy(pst_voxel_plan* plan);
int pst_voxelgrid_filter_async(pst_voxel_plan* plan, const pst_buffer* buffer, pst_buffer* filtered, size_t dst_first, uint64_t* device_count_and_status);

/* ---- LAS record encoder (the writer side of the hot path) ---------------------------- */
/* RawLASWriter::write_points_default_layout, pasture-io/src/las/raw_writers.rs:203-363 (+ write_helpers.rs:10-55):
 * `src` holds points in the DEFAULT typed layout of `point_format` (LasPointFormatN::layout(), las_types.rs; interleaved or
 * columnar), `dst` is an interleaved buffer in the exact-binary record layout (las_layout.rs:70-107); records are written
 * to dst[dst_first .. dst_first + len(src)).  X,Y,Z = (((p - offset) / scale) as i64) checked into i32 — a position outside
 * the i32 range is PST_ERR_RANGE ("Position is out of bounds given the current LAS offset and scale!"), like the expect().
 * Header side effects: bounds_inout = {min xyz, max xyz} of the header, updated with strict compares (:28-48; the writer
 * starts from f64::MAX / f64::MIN); points_by_return[r-1] += number of points whose return number is r, 1 <= r <= max_return
 * (5 for legacy headers, 15 with the large_file block, :220-229). */
int pst_las_encode_points(const pst_buffer* src, uint32_t point_format, const double scale[3], const double offset[3], pst_buffer* dst,
                          size_t dst_first, double bounds_inout[6], uint64_t points_by_return[15], uint32_t max_return);
/* Asynchronous, ranged variant for pipelined writers (device-side encode of chunk i overlapping the D2H copy of chunk i-1):
 * encodes source points [src_first, src_first + count) into dst[dst_first ..) on the current stream.  THIS call's header
 * contribution is left in device memory: device_bounds6 = {min xyz, max xyz} seeded with the identities; device_counts16[0] =
 * number of positions outside the i32 range (> 0 is the panic of write_helpers.rs:15-17 — the caller checks it after
 * synchronising), device_counts16[r] = points with return number r (1..max_return). */
int pst_las_encode_range_async(const pst_buffer* src, size_t src_first, size_t count, uint32_t point_format, const double scale[3],
                               const double offset[3], pst_buffer* dst, size_t dst_first, double* device_bounds6, uint64_t* device_counts16,
                               uint32_t max_return);

/* ---- multi-GPU: points shard by index range, the ONLY exchange is the global AABB ------------------------------
 * The reference has no distributed code: its own index-range processing (convert_into_range buffer_conversion.rs:292; 1 MiB chunks
 * raw_readers.rs:309-349) is what shards without a data-path collective, and the seeds of calculate_bounds (bounds.rs:31-32,
 * +f64::MAX / f64::MIN) are the identities an EMPTY shard contributes.  One process per GPU: rank 0 calls pst_comm_unique_id, ships
 * the 128 bytes to the other ranks (any channel), every rank calls pst_comm_init_rank with its device selected (pst_set_device).
 * pst_comm_init is the single-process form (one handle driving GPUs 0..n-1, ncclCommInitAll).  RCCL is bound at run time; without
 * it these entry points return PST_ERR_UNSUPPORTED. */
int pst_comm_unique_id(pst_comm_id* out_id);
int pst_comm_init_rank(int n_ranks, int rank, const pst_comm_id* id, pst_comm** out);
int pst_comm_init(int n_gpus, pst_comm** out);
int pst_comm_size(const pst_comm* comm, int* out_n_ranks);
int pst_comm_destroy(pst_comm* comm);
/* global AABB, in place, stream-ordered (no host synchronisation): device_rec6 = {min xyz, max xyz} as written by
 * pst_calculate_bounds_async / pst_converter_convert_into_range_with_bounds_async.  ONE ncclAllReduce of 6 x f64 with ncclMin over
 * {min xyz, -max xyz} (48 bytes over xGMI: latency-bound).  All-empty input leaves the seeds, i.e. None (bounds.rs:12-14). */
int pst_bounds_allreduce(pst_comm* comm, double* device_rec6);
/* the same for a pst_comm_init handle: device_recs[d] lives on GPU d, streams[d] (array nullable = default streams) is its stream */
int pst_bounds_allreduce_multi(pst_comm* comm, double* const* device_recs, void* const* streams);
/* The sharded path's per-step exchange as ONE launch (round 6): form = 1 registers a device record address -- from then on every AABB record the
 * library is asked to leave THERE (pst_calculate_bounds_async, pst_converter_convert_into_range_with_bounds_async) is written as
 * {min xyz, -max xyz} by the producing kernel's own last fold, pst_bounds_allreduce[_multi] on it is the ncclAllReduce alone (no negation kernels
 * before and after) and the record STAYS in that form: its reader negates components 3..5.  An empty shard's record is +f64::MAX six times
 * (the identities of bounds.rs:31-32).  form = 0 forgets the address.  Unregistered records keep {min, max} and the three-launch exchange. */
int pst_bounds_record_set_form(double* device_rec6, int form);

#ifdef __cplusplus
}
#endif
#endif /* PASTURE_AMD_H */
