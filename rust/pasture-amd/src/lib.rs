//! pasture's buffer traits and hot-path entry points on top of libpasture_amd.so (MI355X).
//! UNCOMPILED sketch: the build image has no Rust toolchain.  The executable twin of this file is the Python mirror
//! `pasture_amd/*.py`, which drives the same C entry points and is covered by the parity tests.
//!
//! Drop-in surface (same names / argument meaning / panics as pasture-core 0.5):
//!   DeviceVectorBuffer, DeviceHashMapBuffer         ~ VectorBuffer, HashMapBuffer      (containers/point_buffer.rs), storage in HBM
//!   PinnedVectorBuffer, PinnedHashMapBuffer         ~ the same with `&[u8]` views: pinned host memory both pasture's CPU code and the kernels address
//!   DeviceBufferLayoutConverter                     ~ BufferLayoutConverter            (layout/conversion/buffer_conversion.rs)
//!   calculate_bounds, minmax_attribute, compute_centroid, compute_normals, voxelgrid_filter (pasture-algorithms)
//!   SliceDeviceBuffer::slice / slice_mut            ~ SliceBuffer / SliceBufferMut     (containers/slice.rs)
//!   view_attribute_with_conversion                  ~ BorrowedBufferExt::view_attribute_with_conversion (containers/point_buffer.rs:322)
use pasture_amd_sys::*;
use pasture_core::containers::{
    BorrowedBuffer, BorrowedMutBuffer, ColumnarBuffer, ColumnarBufferMut, InterleavedBuffer, InterleavedBufferMut, MakeBufferFromLayout, OwningBuffer,
};
use pasture_core::layout::{PointAttributeDataType, PointAttributeDefinition, PointAttributeMember, PointLayout, PrimitiveType};
use pasture_core::math::AABB;
use pasture_core::nalgebra::{Point3, Vector3};
use std::ffi::{CStr, CString};
use std::ops::Range;
use std::os::raw::c_int;

/// Status codes 2..=14 are the conditions on which pasture itself panics; re-raise them as panics.
fn check(rc: c_int) {
    if rc != PST_OK {
        let msg = unsafe { CStr::from_ptr(pst_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

fn datatype_to_c(dt: PointAttributeDataType) -> pst_datatype {
    use PointAttributeDataType::*;
    let mut d = pst_datatype::default();
    d.kind = match dt {  // declaration order of the enum, point_layout.rs:25-50
        U8 => 0, I8 => 1, U16 => 2, I16 => 3, U32 => 4, I32 => 5, U64 => 6, I64 => 7, F32 => 8, F64 => 9,
        Vec3u8 => 10, Vec3u16 => 11, Vec3f32 => 12, Vec3i32 => 13, Vec3f64 => 14, Vec4u8 => 15,
        ByteArray(n) => { d.size_param = n; 16 }
        Custom { size, min_alignment, name } => { d.size_param = size; d.align_param = min_alignment; d.uuid = *name.as_bytes(); 17 }
    };
    d
}

/// Memory the library has freed stays in its stream-ordered pool (a `convert()` that allocates its target costs microseconds, not 100 ms) and is
/// invisible to every other allocator of the process.  Hand it back -- together with the kNN search's scratch cache -- before another library needs
/// the GPU's memory; an allocation of THIS library that fails does so on its own and tries once more before it reports status 22 (out of memory:
/// `check` turns it into a panic like every other status; a `try_*` surface would return it).  Synchronises the device.
pub fn release_device_memory() { check(unsafe { pst_release_scratch() }); }

struct LayoutHandle(*mut pst_layout);
impl LayoutHandle {
    /// Exact transfer of a `PointLayout` (attribute order, offsets, size) via pst_layout_from_members; the type alignment is the caller's.
    fn new(layout: &PointLayout, type_alignment: u64) -> Self {
        let names: Vec<CString> = layout.attributes().map(|a| CString::new(a.name()).unwrap()).collect();
        let members: Vec<pst_member> = layout.attributes().zip(names.iter())
            .map(|(a, n)| pst_member { name: n.as_ptr(), datatype: datatype_to_c(a.datatype()), offset: a.offset(), size: a.size() })
            .collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { pst_layout_from_members(members.as_ptr(), members.len(), type_alignment, &mut h) });
        LayoutHandle(h)
    }
}
impl Drop for LayoutHandle { fn drop(&mut self) { unsafe { pst_layout_destroy(self.0) }; } }

/// The alignment of the point type, which pasture-core 0.5 keeps private (`PointLayout::memory_layout`).  It is an explicit ARGUMENT of every
/// constructor here (`new_from_layout_aligned`, `for_layouts_aligned`) -- `std::mem::align_of::<P>()` for a `#[derive(PointType)]` struct,
/// 1 for `#[repr(packed)]`, the `max_alignment` that was passed to `PointLayout::add_attribute` otherwise.  Round 3 reconstructed it
/// from the offsets, which cannot tell a `packed(1)` layout whose fields happen to sit at naturally aligned offsets from its unpacked
/// twin (`pst_layout_equals` and `PointLayout::eq` then disagreed).  `natural_alignment` is the default the `MakeBufferFromLayout`
/// impls use -- right for every `repr(C)` layout and for packed layouts whose packing shows in the offsets; pass the true value when
/// in doubt.
pub fn natural_alignment(layout: &PointLayout) -> u64 {
    let max_field = layout.attributes().map(|a| a.datatype().min_alignment()).max().unwrap_or(1).max(1);
    let mut align = 1u64;
    while align * 2 <= max_field
        && layout.size_of_point_entry() % (align * 2) == 0
        && layout.attributes().all(|a| a.offset() % (align * 2) == 0) {
        align *= 2;
    }
    align
}

macro_rules! device_buffer {
    ($name:ident, $storage:expr) => {
        pub struct $name { handle: *mut pst_buffer, layout: PointLayout }
        impl $name {
            pub fn raw(&self) -> *mut pst_buffer { self.handle }
            /// `new_from_layout` with the point type's alignment stated (see `natural_alignment`)
            pub fn new_from_layout_aligned(point_layout: PointLayout, type_alignment: u64) -> Self {
                let l = LayoutHandle::new(&point_layout, type_alignment);
                let mut h = std::ptr::null_mut();
                check(unsafe { pst_buffer_create(l.0, $storage, 0 /* PST_MEM_DEVICE */, &mut h) });
                Self { handle: h, layout: point_layout }
            }
        }
        impl Drop for $name { fn drop(&mut self) { unsafe { pst_buffer_destroy(self.handle) }; } }
        impl<'a> MakeBufferFromLayout<'a> for $name {
            fn new_from_layout(point_layout: PointLayout) -> Self {
                let align = natural_alignment(&point_layout);
                Self::new_from_layout_aligned(point_layout, align)
            }
        }
        impl<'a> BorrowedBuffer<'a> for $name {
            fn len(&self) -> usize { let mut n = 0usize; check(unsafe { pst_buffer_len(self.handle, &mut n) }); n }
            fn point_layout(&self) -> &PointLayout { &self.layout }
            fn get_point(&self, index: usize, data: &mut [u8]) { self.get_point_range(index..index + 1, data) }
            fn get_point_range(&self, range: Range<usize>, data: &mut [u8]) {
                check(unsafe { pst_buffer_read_points(self.handle, range.start, range.len(), data.as_mut_ptr().cast()) })
            }
            fn get_attribute_range(&self, attribute: &PointAttributeDefinition, point_range: Range<usize>, data: &mut [u8]) {
                let name = CString::new(attribute.name()).unwrap();
                let dt = datatype_to_c(attribute.datatype());
                check(unsafe { pst_buffer_read_attribute(self.handle, name.as_ptr(), &dt, point_range.start, point_range.len(), data.as_mut_ptr().cast()) })
            }
            unsafe fn get_attribute_unchecked(&self, member: &PointAttributeMember, index: usize, data: &mut [u8]) {
                self.get_attribute_range(member.attribute_definition(), index..index + 1, data)  // one D2H per call: debugging only
            }
            // as_interleaved()/as_columnar() stay `None` HERE: `&[u8]` views of HBM cannot be handed to the CPU loops.  Bulk work goes
            // through DeviceBufferLayoutConverter and the functions below; PinnedVectorBuffer / PinnedHashMapBuffer (end of this file)
            // are the kinds that answer with real slices.
        }
        impl<'a> BorrowedMutBuffer<'a> for $name {
            unsafe fn set_point(&mut self, index: usize, point_data: &[u8]) { self.set_point_range(index..index + 1, point_data) }
            unsafe fn set_point_range(&mut self, point_range: Range<usize>, point_data: &[u8]) {
                check(pst_buffer_write_points(self.handle, point_range.start, point_range.len(), point_data.as_ptr().cast()))
            }
            unsafe fn set_attribute(&mut self, attribute: &PointAttributeDefinition, index: usize, attribute_data: &[u8]) {
                self.set_attribute_range(attribute, index..index + 1, attribute_data)
            }
            unsafe fn set_attribute_range(&mut self, attribute: &PointAttributeDefinition, point_range: Range<usize>, attribute_data: &[u8]) {
                let name = CString::new(attribute.name()).unwrap();
                let dt = datatype_to_c(attribute.datatype());
                check(pst_buffer_write_attribute(self.handle, name.as_ptr(), &dt, point_range.start, point_range.len(), attribute_data.as_ptr().cast()))
            }
            // point_buffer.rs:229; panics like the reference's assert! when an index is out of bounds (PST_ERR_RANGE -> panic in `check`)
            fn swap(&mut self, from_index: usize, to_index: usize) { check(unsafe { pst_buffer_swap(self.handle, from_index, to_index) }) }
        }
        impl<'a> OwningBuffer<'a> for $name {
            unsafe fn push_points(&mut self, point_bytes: &[u8]) {
                let stride = self.layout.size_of_point_entry() as usize;
                let (old, add) = (self.len(), point_bytes.len() / stride);
                self.resize(old + add);
                self.set_point_range(old..old + add, point_bytes);
            }
            fn resize(&mut self, count: usize) { check(unsafe { pst_buffer_resize(self.handle, count) }) }
            fn clear(&mut self) { self.resize(0) }
        }
    };
}
device_buffer!(DeviceVectorBuffer, PST_STORAGE_INTERLEAVED);
device_buffer!(DeviceHashMapBuffer, PST_STORAGE_COLUMNAR);

pub trait DeviceBuffer { fn handle(&self) -> *mut pst_buffer; }
impl DeviceBuffer for DeviceVectorBuffer { fn handle(&self) -> *mut pst_buffer { self.raw() } }
impl DeviceBuffer for DeviceHashMapBuffer { fn handle(&self) -> *mut pst_buffer { self.raw() } }

/// The closed set standing in for `Fn(T) -> T` (buffer_conversion.rs:14-31).
pub enum DeviceTransform {
    /// `(p * scale) + offset`, two roundings (pasture-io/src/las/raw_readers.rs:42-55)
    Affine { scale: [f64; 3], offset: [f64; 3] },
    /// `(v >> shift) & mask` (raw_readers.rs:61-164)
    BitField { shift: u32, mask: u64 },
}

pub struct DeviceBufferLayoutConverter { handle: *mut pst_converter }
impl Drop for DeviceBufferLayoutConverter { fn drop(&mut self) { unsafe { pst_converter_destroy(self.handle) }; } }
impl DeviceBufferLayoutConverter {
    fn create(from: &PointLayout, to: &PointLayout, with_default: bool) -> Self {
        Self::for_layouts_aligned(from, natural_alignment(from), to, natural_alignment(to), with_default)
    }
    /// `for_layouts` / `for_layouts_with_default` with both point types' alignments stated (see `natural_alignment`)
    pub fn for_layouts_aligned(from: &PointLayout, from_alignment: u64, to: &PointLayout, to_alignment: u64, with_default: bool) -> Self {
        let (f, t) = (LayoutHandle::new(from, from_alignment), LayoutHandle::new(to, to_alignment));
        let mut h = std::ptr::null_mut();
        check(unsafe { pst_converter_create(f.0, t.0, with_default as c_int, &mut h) });
        Self { handle: h }
    }
    pub fn for_layouts(from: &PointLayout, to: &PointLayout) -> Self { Self::create(from, to, false) }
    pub fn for_layouts_with_default(from: &PointLayout, to: &PointLayout) -> Self { Self::create(from, to, true) }
    pub fn set_custom_mapping(&mut self, from: &PointAttributeDefinition, to: &PointAttributeDefinition) {
        let (fname, tname) = (CString::new(from.name()).unwrap(), CString::new(to.name()).unwrap());
        check(unsafe { pst_converter_set_custom_mapping(self.handle, fname.as_ptr(), &datatype_to_c(from.datatype()), tname.as_ptr(), &datatype_to_c(to.datatype())) })
    }
    pub fn set_custom_mapping_with_transformation<T: PrimitiveType>(&mut self, from: &PointAttributeDefinition, to: &PointAttributeDefinition,
                                                                    transform: DeviceTransform, apply_to_source_attribute: bool) {
        let mut xf = pst_transform { kind: 0, shift: 0, datatype: datatype_to_c(T::data_type()), scale: [1.0; 3], offset: [0.0; 3], mask: u64::MAX };
        match transform {
            DeviceTransform::Affine { scale, offset } => { xf.kind = PST_XF_AFFINE; xf.scale = scale; xf.offset = offset; }
            DeviceTransform::BitField { shift, mask } => { xf.kind = PST_XF_BITFIELD; xf.shift = shift; xf.mask = mask; }
        }
        let (fname, tname) = (CString::new(from.name()).unwrap(), CString::new(to.name()).unwrap());
        check(unsafe { pst_converter_set_custom_mapping_with_transformation(self.handle, fname.as_ptr(), &datatype_to_c(from.datatype()), tname.as_ptr(),
                                                                           &datatype_to_c(to.datatype()), &xf, apply_to_source_attribute as c_int) })
    }
    /// The same with the closure as a DEVICE EXPRESSION (include/pasture_amd.h, "device expressions"): what `F: Fn(T) -> T` is in
    /// buffer_conversion.rs:194-234, given as C++ expression text over `v`, `x y z`, `c`, `i` and compiled at run time.  A text that does not
    /// compile panics at the first conversion with the compiler's log (PST_ERR_UNSUPPORTED_TRANSFORM).
    pub fn set_custom_mapping_with_expression(&mut self, from: &PointAttributeDefinition, to: &PointAttributeDefinition, expression: &str,
                                              apply_to_source_attribute: bool) {
        let (fname, tname, text) = (CString::new(from.name()).unwrap(), CString::new(to.name()).unwrap(), CString::new(expression).unwrap());
        check(unsafe { pst_converter_set_custom_mapping_with_expression(self.handle, fname.as_ptr(), &datatype_to_c(from.datatype()), tname.as_ptr(),
                                                                       &datatype_to_c(to.datatype()), text.as_ptr(), apply_to_source_attribute as c_int) })
    }
    /// Compiles (or fetches) the plan-specialised kernel for conversions between these storage kinds now instead of in the background
    /// (`pst_converter_prepare`); returns the PST_PLAN_* family such a conversion takes.
    pub fn prepare(&self, source_columnar: bool, target_columnar: bool, with_bounds: bool) -> u32 {
        let mut kind = 0u32;
        check(unsafe { pst_converter_prepare(self.handle, source_columnar as c_int, target_columnar as c_int, with_bounds as c_int, &mut kind) });
        kind
    }
    /// Which of the two kernel families that can serve a LAS-shaped plan this converter measured to be the faster one on this device (its first
    /// SYNCHRONOUS conversion of at least 2^22 points measures, or `measure_families`): `None` = not measured yet or one family only,
    /// `Some((plan_specialised, [ms LAS, ms plan-specialised]))`.  Records produced from columns have their own slot.
    pub fn family_choice(&self, source_columnar: bool, target_columnar: bool, with_bounds: bool) -> Option<(bool, [f32; 2])> {
        let (mut choice, mut ms) = (0 as c_int, [0f32; 2]);
        let pairing: c_int = if target_columnar { 1 } else if source_columnar { 2 } else { 0 };
        check(unsafe { pst_converter_family_choice(self.handle, pairing, with_bounds as c_int, &mut choice, ms.as_mut_ptr()) });
        if choice == 0 || choice == 1 { Some((choice == 1, ms)) } else { None }
    }
    /// The measurement behind `family_choice`, run now (callers of stream-ordered conversions: once before their loop).
    pub fn measure_families(&self, source: &impl DeviceBuffer, target: &mut impl DeviceBuffer, n: usize, with_bounds: bool) {
        check(unsafe { pst_converter_measure_families(self.handle, source.handle(), 0, n, target.handle(), 0, n, with_bounds as c_int) })
    }
    pub fn convert_into(&self, source: &impl DeviceBuffer, target: &mut impl DeviceBuffer, n: usize) { self.convert_into_range(source, 0..n, target, 0..n) }
    pub fn convert_into_range(&self, source: &impl DeviceBuffer, source_range: Range<usize>, target: &mut impl DeviceBuffer, target_range: Range<usize>) {
        check(unsafe { pst_converter_convert_into_range(self.handle, source.handle(), source_range.start, source_range.end, target.handle(),
                                                       target_range.start, target_range.end) })
    }
}

/// pasture-algorithms/src/bounds.rs:11
pub fn calculate_bounds(buffer: &impl DeviceBuffer) -> Option<AABB<f64>> {
    let (mut mn, mut mx, mut has) = ([0f64; 3], [0f64; 3], 0 as c_int);
    check(unsafe { pst_calculate_bounds(buffer.handle(), mn.as_mut_ptr(), mx.as_mut_ptr(), &mut has) });
    if has == 0 { None } else { Some(AABB::from_min_max_unchecked(Point3::new(mn[0], mn[1], mn[2]), Point3::new(mx[0], mx[1], mx[2]))) }
}

/// pasture-algorithms/src/normal_estimation.rs:198 -- panics on an empty cloud like the reference
pub fn compute_centroid(buffer: &impl DeviceBuffer) -> Vector3<f64> {
    let mut c = [0f64; 3];
    check(unsafe { pst_compute_centroid(buffer.handle(), c.as_mut_ptr()) });
    Vector3::new(c[0], c[1], c[2])
}

/// pasture-algorithms/src/minmax.rs:13 for the stored datatype `T`
pub fn minmax_attribute<T: PrimitiveType + Default + Copy>(buffer: &impl DeviceBuffer, attribute: &PointAttributeDefinition) -> Option<(T, T)> {
    let name = CString::new(attribute.name()).unwrap();
    let (mut mn, mut mx, mut has) = (T::default(), T::default(), 0 as c_int);
    check(unsafe { pst_minmax_attribute(buffer.handle(), name.as_ptr(), &datatype_to_c(T::data_type()), (&mut mn as *mut T).cast(), (&mut mx as *mut T).cast(), &mut has) });
    if has == 0 { None } else { Some((mn, mx)) }
}

/// SliceBuffer::slice / SliceBufferMut::slice_mut (containers/slice.rs:16-43): a view of `range` that every function of this file takes like
/// a buffer -- `calculate_bounds(&buf.slice(a..b))`, the chunked `minmax_attribute` of pasture-tools/src/bin/info.rs:66-78, conversions from
/// and into ranges.  The lifetime ties it to the parent, as BufferSlice<'a, T> does.
pub struct DeviceBufferSlice<'p> { handle: *mut pst_buffer, _parent: std::marker::PhantomData<&'p ()> }
impl<'p> Drop for DeviceBufferSlice<'p> { fn drop(&mut self) { unsafe { pst_buffer_destroy(self.handle) }; } }
impl<'p> DeviceBuffer for DeviceBufferSlice<'p> { fn handle(&self) -> *mut pst_buffer { self.handle } }
pub trait SliceDeviceBuffer: DeviceBuffer {
    fn slice<'p>(&'p self, range: Range<usize>) -> DeviceBufferSlice<'p> {
        let mut h = std::ptr::null_mut();
        check(unsafe { pst_buffer_slice(self.handle(), range.start, range.len(), &mut h) });   // out of bounds: PST_ERR_RANGE -> panic
        DeviceBufferSlice { handle: h, _parent: std::marker::PhantomData }
    }
    fn slice_mut<'p>(&'p mut self, range: Range<usize>) -> DeviceBufferSlice<'p> { self.slice(range) }
}
impl<B: DeviceBuffer> SliceDeviceBuffer for B {}

/// BorrowedBufferExt::view_attribute_with_conversion::<T>(attribute).into_iter().collect() (point_buffer.rs:322-330, buffer_views.rs:533-650)
pub fn view_attribute_with_conversion<T: PrimitiveType + Default + Clone>(buffer: &impl DeviceBuffer, attribute: &PointAttributeDefinition) -> Result<Vec<T>, String> {
    assert_eq!(T::data_type(), attribute.datatype());   // buffer_views.rs:548
    let mut n = 0usize;
    check(unsafe { pst_buffer_len(buffer.handle(), &mut n) });
    let mut out = vec![T::default(); n];
    let name = CString::new(attribute.name()).unwrap();
    let rc = unsafe { pst_buffer_read_attribute_converted(buffer.handle(), name.as_ptr(), &datatype_to_c(T::data_type()), 0, n, out.as_mut_ptr().cast()) };
    if rc == 5 /* PST_ERR_INVALID_CONVERSION */ { return Err("Conversion between attribute types is impossible".into()); }   // the reference's Err(..) :553-561
    check(rc);
    Ok(out)
}

/// pasture-algorithms/src/normal_estimation.rs:79
pub fn compute_normals(buffer: &impl DeviceBuffer, k_nn: usize) -> Vec<(Vector3<f64>, f64)> {
    // the output length comes from the buffer itself: a caller-supplied count smaller than it would let the C side write past the Vecs
    let mut n_points = 0usize;
    check(unsafe { pst_buffer_len(buffer.handle(), &mut n_points) });
    let (mut normals, mut curvature) = (vec![0f64; 3 * n_points], vec![0f64; n_points]);
    check(unsafe { pst_compute_normals(buffer.handle(), k_nn, normals.as_mut_ptr(), curvature.as_mut_ptr(), std::ptr::null_mut()) });
    (0..n_points).map(|i| (Vector3::new(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]), curvature[i])).collect()
}

/// pasture-algorithms/src/voxel_grid.rs:109
pub fn voxelgrid_filter(buffer: &impl DeviceBuffer, leafsize_x: f64, leafsize_y: f64, leafsize_z: f64, filtered_buffer: &mut impl DeviceBuffer) {
    check(unsafe { pst_voxelgrid_filter(buffer.handle(), leafsize_x, leafsize_y, leafsize_z, filtered_buffer.handle()) })
}

impl DeviceHashMapBuffer {
    /// HashMapBuffer::filter_into (point_buffer.rs:1082-1136); the closure is evaluated into a byte mask once.
    pub fn filter_into<F: Fn(usize) -> bool>(&self, buffer: &mut impl DeviceBuffer, predicate: F, num_matches_hint: Option<usize>) -> usize {
        let mut len = 0usize;  // the mask must cover every point of `self`: its length is read from the buffer, never taken from the caller
        check(unsafe { pst_buffer_len(self.raw(), &mut len) });
        let mask: Vec<u8> = (0..len).map(|i| predicate(i) as u8).collect();
        let mut matches = 0usize;
        check(unsafe { pst_buffer_filter_into(self.raw(), buffer.handle(), mask.as_ptr(), /* host memory */ 1,
                                              num_matches_hint.map(|n| n as i64).unwrap_or(-1), &mut matches) });
        matches
    }

    /// The stream-ordered form for `Some(num_matches)` and a mask that already lives on the device: count, scan and copies are enqueued and the
    /// call returns.  `device_count_out` (device-accessible, may be null) receives the number of mask hits in stream order; more hits than
    /// `num_matches` is the reference's slice panic (point_buffer.rs:1103-1108) and the caller's to check after its own synchronisation.
    ///
    /// # Safety
    /// `device_mask` must point to `self.len()` bytes of device memory that stay valid until the stream has run the call.
    pub unsafe fn filter_into_async(&self, buffer: &mut impl DeviceBuffer, device_mask: *const u8, num_matches: usize, device_count_out: *mut u64) {
        check(pst_buffer_filter_into_async(self.raw(), buffer.handle(), device_mask, num_matches, device_count_out))
    }
}

/// BorrowedMutBufferExt::transform_attribute (point_buffer.rs:391-404) with `|index, value| expression` as a device expression; `device_params`
/// (at most four device arrays of f64) are what the closure would capture: `p0[3 * i + c]`.
///
/// # Safety
/// Every pointer of `device_params` must be device memory valid for the indices the expression uses.
pub unsafe fn transform_attribute_expr(buffer: &mut impl DeviceBuffer, attribute: &PointAttributeDefinition, expression: &str, device_params: &[*const f64]) {
    let (name, text) = (CString::new(attribute.name()).unwrap(), CString::new(expression).unwrap());
    check(pst_transform_attribute_expr(buffer.handle(), name.as_ptr(), &datatype_to_c(attribute.datatype()), text.as_ptr(),
                                       if device_params.is_empty() { std::ptr::null() } else { device_params.as_ptr() }, device_params.len()))
}

impl DeviceHashMapBuffer {
    /// HashMapBuffer::filter (point_buffer.rs:1064-1076) with the predicate as a device expression over the layout's attribute names:
    /// `buffer.filter_expr::<DeviceVectorBuffer>("Classification == 2 && Position3D.z < 120.0")`.  Returns the raw handle of the new buffer's
    /// storage kind `out_storage` (PST_STORAGE_*); wrap it with the matching Device*Buffer.
    pub fn filter_expr_raw(&self, expression: &str, out_storage: u32) -> *mut pst_buffer {
        let text = CString::new(expression).unwrap();
        let mut out: *mut pst_buffer = std::ptr::null_mut();
        check(unsafe { pst_buffer_filter_expr(self.raw(), text.as_ptr(), std::ptr::null(), 0, out_storage, &mut out) });
        out
    }
}

/// OwningBufferExt::append (point_buffer.rs:419-489)
pub fn append(this: &mut impl DeviceBuffer, other: &impl DeviceBuffer) { check(unsafe { pst_buffer_append(this.handle(), other.handle()) }) }

/// RawLASWriter::write_points_default_layout (pasture-io/src/las/raw_writers.rs:203-363): typed points -> raw records + header
/// side effects.  `bounds` = [min xyz, max xyz] of the header, `points_by_return[r - 1]` the count of return number r.
pub fn las_encode_points(points: &impl DeviceBuffer, point_format: u8, scale: [f64; 3], offset: [f64; 3], records: &mut DeviceVectorBuffer,
                         first_record: usize, bounds: &mut [f64; 6], points_by_return: &mut [u64; 15], large_file: bool) {
    check(unsafe { pst_las_encode_points(points.handle(), point_format as u32, scale.as_ptr(), offset.as_ptr(), records.raw(), first_record,
                                         bounds.as_mut_ptr(), points_by_return.as_mut_ptr(), if large_file { 15 } else { 5 }) })
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Pinned-host buffers: ONE allocation that pasture's unmodified CPU code and the kernels both address.
//
// `pst_buffer_create(.., PST_MEM_PINNED_HOST)` allocates the storage with hipHostMalloc: host memory that the GPU reads and writes in place
// over the host link (zero-copy).  Its pointers (`pst_buffer_points_ptr` / `pst_buffer_column_ptr`) are ordinary host addresses, so these
// types CAN answer `as_interleaved()` / `as_columnar()` with real `&[u8]` slices (containers/point_buffer.rs:504-654) -- pasture's own
// dispatch (`buffer_conversion.rs:326-356`, `.expect("Source buffer must either be an interleaved or columnar buffer")`), its views,
// iterators, readers and writers work on them unchanged -- while DeviceBufferLayoutConverter / calculate_bounds / compute_normals take the
// same handle.  The price is the link: a kernel streams such a buffer at ~55 GB/s instead of ~6 TB/s.  Use them where the CPU side
// touches the points anyway (a reader that fills a chunk, a writer that drains one: pasture-io's 1 MiB chunk loops) and keep
// DeviceVectorBuffer / DeviceHashMapBuffer for everything that stays on the GPU between kernels.
//
// Synchronisation: the kernels are stream-ordered; a `&[u8]` view must not be read while a launch that writes the buffer is in flight.
// The synchronous entry points (everything in this file) return after the stream has drained; after an `*_async` call the caller runs
// `pst_stream_synchronize()` before touching the slices.
macro_rules! pinned_common {
    ($name:ident, $storage:expr) => {
        pub struct $name { handle: *mut pst_buffer, layout: PointLayout }
        impl $name {
            pub fn raw(&self) -> *mut pst_buffer { self.handle }
            fn len_(&self) -> usize { let mut n = 0usize; check(unsafe { pst_buffer_len(self.handle, &mut n) }); n }
        }
        impl Drop for $name { fn drop(&mut self) { unsafe { pst_buffer_destroy(self.handle) }; } }
        impl DeviceBuffer for $name { fn handle(&self) -> *mut pst_buffer { self.raw() } }
        impl<'a> MakeBufferFromLayout<'a> for $name {
            fn new_from_layout(point_layout: PointLayout) -> Self {
                let l = LayoutHandle::new(&point_layout, natural_alignment(&point_layout));
                let mut h = std::ptr::null_mut();
                check(unsafe { pst_buffer_create(l.0, $storage, PST_MEM_PINNED_HOST, &mut h) });
                Self { handle: h, layout: point_layout }
            }
        }
        impl<'a> OwningBuffer<'a> for $name {
            unsafe fn push_points(&mut self, point_bytes: &[u8]) {
                let stride = self.layout.size_of_point_entry() as usize;
                let (old, add) = (self.len_(), point_bytes.len() / stride);
                self.resize(old + add);
                self.set_point_range(old..old + add, point_bytes);
            }
            /// zero-fills new points like `Vec::resize(_, 0)` (point_buffer.rs:777-781, 1308-1320); may move the storage: slices taken
            /// before are invalidated by the borrow checker (`&mut self`)
            fn resize(&mut self, count: usize) { check(unsafe { pst_buffer_resize(self.handle, count) }) }
            fn clear(&mut self) { self.resize(0) }
        }
    };
}

pinned_common!(PinnedVectorBuffer, PST_STORAGE_INTERLEAVED);
impl PinnedVectorBuffer {
    fn bytes(&self) -> &[u8] {
        let mut p: *mut std::os::raw::c_void = std::ptr::null_mut();
        check(unsafe { pst_buffer_points_ptr(self.handle, &mut p) });
        let n = self.len_() * self.layout.size_of_point_entry() as usize;
        if n == 0 { &[] } else { unsafe { std::slice::from_raw_parts(p as *const u8, n) } }
    }
    fn bytes_mut(&mut self) -> &mut [u8] {
        let mut p: *mut std::os::raw::c_void = std::ptr::null_mut();
        check(unsafe { pst_buffer_points_ptr(self.handle, &mut p) });
        let n = self.len_() * self.layout.size_of_point_entry() as usize;
        if n == 0 { &mut [] } else { unsafe { std::slice::from_raw_parts_mut(p as *mut u8, n) } }
    }
}
impl<'a> BorrowedBuffer<'a> for PinnedVectorBuffer {
    fn len(&self) -> usize { self.len_() }
    fn point_layout(&self) -> &PointLayout { &self.layout }
    // the same copies VectorBuffer makes (point_buffer.rs:706-739), from the shared allocation
    fn get_point(&self, index: usize, data: &mut [u8]) { data.copy_from_slice(self.get_point_ref(index)) }
    fn get_point_range(&self, range: Range<usize>, data: &mut [u8]) { data.copy_from_slice(self.get_point_range_ref(range)) }
    unsafe fn get_attribute_unchecked(&self, member: &PointAttributeMember, index: usize, data: &mut [u8]) {
        let stride = self.layout.size_of_point_entry() as usize;
        let start = index * stride + member.offset() as usize;
        data.copy_from_slice(&self.bytes()[start..start + member.size() as usize])
    }
    fn as_interleaved(&self) -> Option<&dyn InterleavedBuffer<'a>> { Some(self) }   // pasture's dispatch takes the interleaved branch
}
impl<'a> InterleavedBuffer<'a> for PinnedVectorBuffer {
    fn get_point_ref<'b>(&'b self, index: usize) -> &'b [u8] where 'a: 'b { self.get_point_range_ref(index..index + 1) }
    fn get_point_range_ref<'b>(&'b self, range: Range<usize>) -> &'b [u8] where 'a: 'b {
        let stride = self.layout.size_of_point_entry() as usize;
        &self.bytes()[range.start * stride..range.end * stride]   // point_buffer.rs:856-866
    }
}
impl<'a> BorrowedMutBuffer<'a> for PinnedVectorBuffer {
    unsafe fn set_point(&mut self, index: usize, point_data: &[u8]) { self.get_point_mut(index).copy_from_slice(point_data) }
    unsafe fn set_point_range(&mut self, point_range: Range<usize>, point_data: &[u8]) { self.get_point_range_mut(point_range).copy_from_slice(point_data) }
    unsafe fn set_attribute(&mut self, attribute: &PointAttributeDefinition, index: usize, attribute_data: &[u8]) {
        self.set_attribute_range(attribute, index..index + 1, attribute_data)
    }
    unsafe fn set_attribute_range(&mut self, attribute: &PointAttributeDefinition, point_range: Range<usize>, attribute_data: &[u8]) {
        let m = self.layout.get_attribute(attribute).expect("Attribute not found in PointLayout of buffer").clone();
        let (stride, off, size) = (self.layout.size_of_point_entry() as usize, m.offset() as usize, m.size() as usize);
        let first = point_range.start;
        let bytes = self.bytes_mut();
        for (i, chunk) in attribute_data.chunks_exact(size).enumerate() {
            let start = (first + i) * stride + off;
            bytes[start..start + size].copy_from_slice(chunk);
        }
    }
    fn swap(&mut self, from_index: usize, to_index: usize) {
        let stride = self.layout.size_of_point_entry() as usize;
        if from_index == to_index { return; }
        let (lo, hi) = (from_index.min(to_index), from_index.max(to_index));
        let (a, b) = self.bytes_mut().split_at_mut(hi * stride);
        a[lo * stride..(lo + 1) * stride].swap_with_slice(&mut b[..stride]);
    }
    fn as_interleaved_mut(&mut self) -> Option<&mut dyn InterleavedBufferMut<'a>> { Some(self) }
}
impl<'a> InterleavedBufferMut<'a> for PinnedVectorBuffer {
    fn get_point_mut<'b>(&'b mut self, index: usize) -> &'b mut [u8] where 'a: 'b { self.get_point_range_mut(index..index + 1) }
    fn get_point_range_mut<'b>(&'b mut self, range: Range<usize>) -> &'b mut [u8] where 'a: 'b {
        let stride = self.layout.size_of_point_entry() as usize;
        &mut self.bytes_mut()[range.start * stride..range.end * stride]
    }
}

pinned_common!(PinnedHashMapBuffer, PST_STORAGE_COLUMNAR);
impl PinnedHashMapBuffer {
    fn column(&self, attribute: &PointAttributeDefinition) -> (*mut u8, usize) {
        let name = CString::new(attribute.name()).unwrap();
        let dt = datatype_to_c(attribute.datatype());
        let mut p: *mut std::os::raw::c_void = std::ptr::null_mut();
        // PST_ERR_MISSING_ATTRIBUTE -> the panic of HashMapBuffer::get_attribute_ref (point_buffer.rs:1374-1387)
        check(unsafe { pst_buffer_column_ptr(self.handle, name.as_ptr(), &dt, &mut p) });
        (p as *mut u8, attribute.size() as usize)
    }
}
impl<'a> BorrowedBuffer<'a> for PinnedHashMapBuffer {
    fn len(&self) -> usize { self.len_() }
    fn point_layout(&self) -> &PointLayout { &self.layout }
    fn get_point(&self, index: usize, data: &mut [u8]) {   // gathers the attributes like HashMapBuffer::get_point (:1131-1147)
        for a in self.layout.attributes() {
            let (off, size) = (a.offset() as usize, a.size() as usize);
            data[off..off + size].copy_from_slice(self.get_attribute_ref(a.attribute_definition(), index));
        }
    }
    fn get_point_range(&self, range: Range<usize>, data: &mut [u8]) {
        let stride = self.layout.size_of_point_entry() as usize;
        for (k, i) in range.enumerate() { self.get_point(i, &mut data[k * stride..(k + 1) * stride]) }
    }
    unsafe fn get_attribute_unchecked(&self, member: &PointAttributeMember, index: usize, data: &mut [u8]) {
        data.copy_from_slice(self.get_attribute_ref(member.attribute_definition(), index))
    }
    fn as_columnar(&self) -> Option<&dyn ColumnarBuffer<'a>> { Some(self) }   // pasture's dispatch takes the columnar branch
}
impl<'a> ColumnarBuffer<'a> for PinnedHashMapBuffer {
    fn get_attribute_ref<'b>(&'b self, attribute: &PointAttributeDefinition, index: usize) -> &'b [u8] where 'a: 'b {
        self.get_attribute_range_ref(attribute, index..index + 1)
    }
    fn get_attribute_range_ref<'b>(&'b self, attribute: &PointAttributeDefinition, range: Range<usize>) -> &'b [u8] where 'a: 'b {
        let (p, size) = self.column(attribute);
        assert!(range.end <= self.len_());
        if range.is_empty() { &[] } else { unsafe { std::slice::from_raw_parts(p.add(range.start * size), range.len() * size) } }   // :1389-1402
    }
}
impl<'a> BorrowedMutBuffer<'a> for PinnedHashMapBuffer {
    unsafe fn set_point(&mut self, index: usize, point_data: &[u8]) {
        let members: Vec<PointAttributeMember> = self.layout.attributes().cloned().collect();
        for a in &members {
            let (off, size) = (a.offset() as usize, a.size() as usize);
            self.get_attribute_mut(a.attribute_definition(), index).copy_from_slice(&point_data[off..off + size]);
        }
    }
    unsafe fn set_point_range(&mut self, point_range: Range<usize>, point_data: &[u8]) {
        let stride = self.layout.size_of_point_entry() as usize;
        for (k, i) in point_range.enumerate() { self.set_point(i, &point_data[k * stride..(k + 1) * stride]) }
    }
    unsafe fn set_attribute(&mut self, attribute: &PointAttributeDefinition, index: usize, attribute_data: &[u8]) {
        self.get_attribute_mut(attribute, index).copy_from_slice(attribute_data)
    }
    unsafe fn set_attribute_range(&mut self, attribute: &PointAttributeDefinition, point_range: Range<usize>, attribute_data: &[u8]) {
        self.get_attribute_range_mut(attribute, point_range).copy_from_slice(attribute_data)
    }
    fn swap(&mut self, from_index: usize, to_index: usize) {
        let len = self.len_();
        assert!(from_index < len && to_index < len, "swap index out of bounds");   // HashMapBuffer::swap panics likewise (slice indexing); without it
                                                                                    // the raw-pointer swap below would write outside the allocation
        let members: Vec<PointAttributeMember> = self.layout.attributes().cloned().collect();
        for a in &members {
            let (p, size) = self.column(a.attribute_definition());
            if from_index != to_index { unsafe { std::ptr::swap_nonoverlapping(p.add(from_index * size), p.add(to_index * size), size) } }
        }
    }
    fn as_columnar_mut(&mut self) -> Option<&mut dyn ColumnarBufferMut<'a>> { Some(self) }
}
impl<'a> ColumnarBufferMut<'a> for PinnedHashMapBuffer {
    fn get_attribute_mut<'b>(&'b mut self, attribute: &PointAttributeDefinition, index: usize) -> &'b mut [u8] where 'a: 'b {
        self.get_attribute_range_mut(attribute, index..index + 1)
    }
    fn get_attribute_range_mut<'b>(&'b mut self, attribute: &PointAttributeDefinition, range: Range<usize>) -> &'b mut [u8] where 'a: 'b {
        let (p, size) = self.column(attribute);
        assert!(range.end <= self.len_());
        if range.is_empty() { &mut [] } else { unsafe { std::slice::from_raw_parts_mut(p.add(range.start * size), range.len() * size) } }   // :1425-1438
    }
}
