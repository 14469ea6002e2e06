"""Experiment: generic tile kernels over record strides (bank-conflict behaviour of the lane -> record mapping).
usage: exp_strides.py <h2v|v2v|v2h>   (env PST_TILE_QUAD=0/1 selects the lane mapping)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 50_000_000
mode = sys.argv[1] if len(sys.argv) > 1 else "h2v"
LAYOUTS = {
    16: [A.GPS_TIME, A.POINT_ID],
    24: [A.POSITION_3D],
    26: [A.POSITION_3D, A.INTENSITY],
    27: [A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION],
    28: [A.POSITION_3D, A.INTENSITY, A.POINT_SOURCE_ID],
    32: [A.POSITION_3D, A.GPS_TIME],
    33: [A.POSITION_3D, A.GPS_TIME, A.CLASSIFICATION],
    36: [A.POSITION_3D, A.NORMAL],
    40: [A.POSITION_3D, A.GPS_TIME, A.COLOR_RGB, A.INTENSITY],
    41: [A.POSITION_3D, A.GPS_TIME, A.COLOR_RGB, A.INTENSITY, A.CLASSIFICATION],
    48: [A.POSITION_3D, A.GPS_TIME, A.POINT_ID, A.WAVEFORM_DATA_OFFSET],
    56: [A.POSITION_3D, A.GPS_TIME, A.POINT_ID, A.WAVEFORM_DATA_OFFSET, A.WAVEFORM_PACKET_SIZE, A.RETURN_POINT_WAVEFORM_LOCATION],
    64: [A.POSITION_3D, A.GPS_TIME, A.COLOR_RGB, A.INTENSITY, A.NORMAL, A.WAVEFORM_PARAMETERS],
}
for stride, attrs in LAYOUTS.items():
    lay = PointLayout.from_attributes_packed(attrs, 1)
    assert lay.size_of_point_entry() == stride, (stride, lay.size_of_point_entry())
    src = (pa.HashMapBuffer if mode[0] == "h" else pa.VectorBuffer).new_from_layout(lay); src.resize(n); src.synth_fill(42, 0)
    dst = (pa.HashMapBuffer if mode[2] == "h" else pa.VectorBuffer).new_from_layout(lay); dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(lay, lay)
    r = range(0, n)
    for _ in range(3): conv.convert_into_range_async(src, r, dst, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): conv.convert_into_range_async(src, r, dst, r)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(mode, "stride", stride, f"{ms:.3f} ms", f"{2 * stride * n / ms / 1e9:.2f} TB/s", flush=True)
    del src, dst
