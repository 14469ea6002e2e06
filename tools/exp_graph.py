"""Experiment: hipGraph capture of a launch-bound chunk loop (64 x 1 MiB raw LAS-0 decode calls) through torch.cuda.CUDAGraph."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api()
raw = las.point_layout_from_las_point_format(las.Format(0), True); typed = las.point_layout_from_las_point_format(las.Format(0), False)
chunk, k = 52_428, 256
N = chunk * k
src = pa.VectorBuffer.new_from_layout(raw); src.resize(N); src.synth_fill(1, 0)
dst = pa.HashMapBuffer.new_from_layout(typed); dst.resize(N)
ref = pa.HashMapBuffer.new_from_layout(typed); ref.resize(N)
conv = las.get_default_las_converter(raw, typed, (0.001,) * 3, (0.0,) * 3)
s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
conv.convert_into(src, ref)
def loop(target):
    for c in range(k):
        conv.convert_into_range_async(src, range(c * chunk, (c + 1) * chunk), target, range(c * chunk, (c + 1) * chunk))
loop(dst); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(dst); torch.cuda.synchronize(); eager = time.perf_counter() - t0
g = torch.cuda.CUDAGraph()
cs = torch.cuda.Stream()
with torch.cuda.stream(cs):
    api.set_stream(ctypes.c_void_p(cs.cuda_stream))
    with torch.cuda.graph(g, stream=cs):
        loop(dst)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); graph = time.perf_counter() - t0
ok = all(torch.equal(torch.as_tensor(dst.view_attribute(a.attribute_definition())), torch.as_tensor(ref.view_attribute(a.attribute_definition()))) for a in typed.attributes())
print(f"{k} calls of {chunk} points: eager {eager * 1e3:.2f} ms ({N / eager / 1e9:.1f} Gpts/s), hipGraph replay {graph * 1e3:.2f} ms ({N / graph / 1e9:.1f} Gpts/s), identical: {ok}")
