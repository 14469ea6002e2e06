// LAS point layouts shared by the encoder plumbing (las_api.cpp) and the decoder's plan matcher (converter.cpp).
// Reference: pasture-io/src/las/las_layout.rs:64-125 (exact-binary record layout), las_types.rs (LasPointFormatN::layout()).
#pragma once
#include "core.hpp"

namespace pst {
namespace laslayout {

struct Fmt { bool ext, gps, color, nir, wave; };
inline Fmt fmt_of(uint32_t n) {
  return Fmt{n >= 6, n == 1 || n == 3 || n == 4 || n == 5 || n >= 6, n == 2 || n == 3 || n == 5 || n == 7 || n == 8 || n == 10, n == 8 || n == 10,
             n == 4 || n == 5 || n == 9 || n == 10};
}
inline AttributeDef def(const char* name, uint32_t kind) {
  AttributeDef d{name, DataType{}};
  d.datatype.kind = kind;
  return d;
}
// LasPointFormatN::layout(): #[repr(C, packed)] structs of las_types.rs, field order = attribute order
inline Layout typed_layout(uint32_t format) {
  const Fmt f = fmt_of(format);
  Layout l;
  auto add = [&](const char* n, uint32_t k) { l.add_attribute(def(n, k), true, 1); };
  add("Position3D", PST_VEC3F64); add("Intensity", PST_U16); add("ReturnNumber", PST_U8); add("NumberOfReturns", PST_U8);
  if (f.ext) { add("ClassificationFlags", PST_U8); add("ScannerChannel", PST_U8); }
  add("ScanDirectionFlag", PST_U8); add("EdgeOfFlightLine", PST_U8); add("Classification", PST_U8);
  if (f.ext) { add("UserData", PST_U8); add("ScanAngle", PST_I16); } else { add("ScanAngleRank", PST_I8); add("UserData", PST_U8); }
  add("PointSourceID", PST_U16);
  if (f.gps) add("GpsTime", PST_F64);
  if (f.color) add("ColorRGB", PST_VEC3U16);
  if (f.nir) add("NIR", PST_U16);
  if (f.wave) {
    add("WavePacketDescriptorIndex", PST_U8); add("WaveformDataOffset", PST_U64); add("WaveformPacketSize", PST_U32);
    add("ReturnPointWaveformLocation", PST_F32); add("WaveformParameters", PST_VEC3F32);
  }
  return l;
}
// point_layout_from_las_point_format(format, exact_binary_representation = true), las_layout.rs:70-107
inline Layout raw_layout(uint32_t format) {
  const Fmt f = fmt_of(format);
  Layout l;
  auto add = [&](const char* n, uint32_t k) { l.add_attribute(def(n, k), true, 1); };
  add("LASLocalPosition", PST_VEC3I32); add("Intensity", PST_U16);
  if (f.ext) add("LASExtendedFlags", PST_U16); else add("LASBasicFlags", PST_U8);
  add("Classification", PST_U8);
  if (f.ext) { add("UserData", PST_U8); add("ScanAngle", PST_I16); } else { add("ScanAngleRank", PST_I8); add("UserData", PST_U8); }
  add("PointSourceID", PST_U16);
  if (f.gps) add("GpsTime", PST_F64);
  if (f.color) add("ColorRGB", PST_VEC3U16);
  if (f.nir) add("NIR", PST_U16);
  if (f.wave) {
    add("WavePacketDescriptorIndex", PST_U8); add("WaveformDataOffset", PST_U64); add("WaveformPacketSize", PST_U32);
    add("ReturnPointWaveformLocation", PST_F32); add("WaveformParameters", PST_VEC3F32);
  }
  return l;
}


}  // namespace laslayout
}  // namespace pst
