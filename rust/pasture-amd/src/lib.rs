//! pasture's buffer traits and hot-path entry points on top of libpasture_amd.so (MI355X).
//! UNCOMPILED sketch: the build image has no Rust toolchain.  The executable twin of this file is the Python mirror
//! `pasture_amd/*.py`, which drives the same C entry points and is covered by the parity tests.
//!
//! Drop-in surface (same names / argument meaning / panics as pasture-core 0.5):
//!   DeviceVectorBuffer, DeviceHashMapBuffer         ~ VectorBuffer, HashMapBuffer      (containers/point_buffer.rs)
//!   DeviceBufferLayoutConverter                     ~ BufferLayoutConverter            (layout/conversion/buffer_conversion.rs)
//!   calculate_bounds, minmax_attribute, compute_normals, transform_attribute           (pasture-algorithms)
use pasture_amd_sys::*;
use pasture_core::containers::{BorrowedBuffer, BorrowedMutBuffer, MakeBufferFromLayout, OwningBuffer};
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

struct LayoutHandle(*mut pst_layout);
impl LayoutHandle {
    /// Exact transfer of a `PointLayout` (attribute order, offsets, size, alignment) via pst_layout_from_members.
    fn new(layout: &PointLayout) -> Self {
        let names: Vec<CString> = layout.attributes().map(|a| CString::new(a.name()).unwrap()).collect();
        let members: Vec<pst_member> = layout.attributes().zip(names.iter())
            .map(|(a, n)| pst_member { name: n.as_ptr(), datatype: datatype_to_c(a.datatype()), offset: a.offset(), size: a.size() })
            .collect();
        let mut h = std::ptr::null_mut();
        // size_of_point_entry is a multiple of the type alignment; the alignment itself is recovered as the largest power of
        // two <= max field alignment that divides the size (PointLayout does not expose it directly in 0.5).
        let align = layout_alignment(layout);
        check(unsafe { pst_layout_from_members(members.as_ptr(), members.len(), align, &mut h) });
        LayoutHandle(h)
    }
}
impl Drop for LayoutHandle { fn drop(&mut self) { unsafe { pst_layout_destroy(self.0) }; } }
/// pasture-core 0.5 keeps `memory_layout` private, so the alignment is reconstructed: the largest power of two that is at most the
/// largest `min_alignment` of the attribute datatypes and divides the point size and every attribute offset.  This is the value
/// `add_attribute` / the derive macro produce for `repr(C)` and `repr(packed(n))` layouts whose packing is visible in the offsets; a
/// `packed(1)` layout whose fields happen to sit at naturally aligned offsets is indistinguishable from its unpacked twin here (the two
/// then compare equal on the device side although `PointLayout::eq` separates them) -- an upstream accessor would remove the guess.
fn layout_alignment(layout: &PointLayout) -> u64 {
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
        impl $name { pub fn raw(&self) -> *mut pst_buffer { self.handle } }
        impl Drop for $name { fn drop(&mut self) { unsafe { pst_buffer_destroy(self.handle) }; } }
        impl<'a> MakeBufferFromLayout<'a> for $name {
            fn new_from_layout(point_layout: PointLayout) -> Self {
                let l = LayoutHandle::new(&point_layout);
                let mut h = std::ptr::null_mut();
                check(unsafe { pst_buffer_create(l.0, $storage, 0 /* PST_MEM_DEVICE */, &mut h) });
                Self { handle: h, layout: point_layout }
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
            // as_interleaved()/as_columnar() stay `None`: `&[u8]` views of HBM cannot be handed to the CPU loops.  Bulk work
            // goes through DeviceBufferLayoutConverter and the functions below.
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
            fn swap(&mut self, _from_index: usize, _to_index: usize) { unimplemented!("per-point swap is not a bulk path") }
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
        let (f, t) = (LayoutHandle::new(from), LayoutHandle::new(to));
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
