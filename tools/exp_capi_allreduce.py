"""Per-call cost of pst_bounds_allreduce at world size 1 (RCCL, one GPU): host enqueue time and device time, on torch's current stream and on a
second, high-priority stream (what BoundsExchange does)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.distributed import Communicator

api = pa.product_api()
api.set_device(0)
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=-1)
comm = Communicator.from_unique_id(1, 0, Communicator.unique_id())
rec = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], dtype=torch.float64, device="cuda")
for name, st in (("main stream", main), ("side stream", side)):
    api.set_stream(ctypes.c_void_p(st.cuda_stream))
    for _ in range(5):
        comm.allreduce_bounds(rec.data_ptr())
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        comm.allreduce_bounds(rec.data_ptr())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e6 * (t1 - t0) / n:.1f} us per call, with sync {1e6 * (t2 - t0) / n:.1f} us per call", flush=True)
api.set_stream(ctypes.c_void_p(main.cuda_stream))
# the pieces of BoundsExchange.submit
n = 200
t0 = time.perf_counter()
for _ in range(n):
    e = torch.cuda.Event(); e.record(main); side.wait_event(e)
    api.set_stream(ctypes.c_void_p(side.cuda_stream)); api.set_stream(ctypes.c_void_p(main.cuda_stream))
    d = torch.cuda.Event(); d.record(side); main.wait_event(d)
torch.cuda.synchronize()
print(f"events + stream switches only: {1e6 * (time.perf_counter() - t0) / n:.1f} us per step")
comm.destroy()
