"""Experiment (single rank under torchrun): host enqueue time per step of the bench loop with the asynchronous AABB all-reduce."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import pasture_amd as pa
from pasture_amd.distributed import PipelinedBoundsReduce
from pasture_amd.layout import attributes as A, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
layout = pa.PointLayout.from_attributes([A.POSITION_3D])
src = pa.HashMapBuffer.new_from_layout(layout); src.resize(n); src.synth_fill(42, 0)
dst = pa.HashMapBuffer.new_from_layout(layout); dst.resize(n)
conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, (0.001,) * 3, (1.0, 2.0, 3.0)), False)
ring = PipelinedBoundsReduce(lambda: torch.empty(6, dtype=torch.float64, device="cuda"), depth=int(os.environ.get("DEPTH", "4")))
for mode in ("kernel only", "kernel + async all-reduce"):
    for _ in range(5):
        conv.convert_into_with_bounds_async(src, dst, ring.current().data_ptr())
        if mode != "kernel only": ring.submit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        conv.convert_into_with_bounds_async(src, dst, ring.current().data_ptr())
        if mode != "kernel only": ring.submit()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:28s}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step", flush=True)
ring.finish()
dist.destroy_process_group()
