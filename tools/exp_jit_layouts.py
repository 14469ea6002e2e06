"""Random packed layouts through the generic converter, interpreted against plan-specialised (hipRTC) kernels.

For every seed: a random source layout (scalars, Vec3s, opaque byte strings; packed(1)), a target layout holding the same attributes in another
order with about a third of the datatypes changed (Rust `as`), and the three pairings with an interleaved side (records -> columns, columns ->
records, records -> records).  Per pairing: the interpreted tile kernel (jit off), the specialised kernel (compiled synchronously), the two
outputs compared byte for byte on the device, and the time of each as a fraction of the 8 TB/s peak over the bytes the pairing moves (record
sizes on interleaved sides, attribute sizes on columnar sides).

usage: exp_jit_layouts.py [--seeds 20] [--points 100000000] [--out profiles/r04_random_layouts.jsonl] [--first-seed 0]
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pasture_amd as pa
from pasture_amd import conversion as cv
from pasture_amd.layout import PointAttributeDataType as T
from pasture_amd.layout import PointAttributeDefinition, PointLayout

PEAK = 8.0e12
SC = [T.U8, T.I8, T.U16, T.I16, T.U32, T.I32, T.U64, T.I64, T.F32, T.F64]
V3 = [T.Vec3u8, T.Vec3u16, T.Vec3f32, T.Vec3i32, T.Vec3f64]
OPAQUE = [T.Vec4u8, T.ByteArray(5), T.ByteArray(16)]


def random_layouts(seed):
    rng = np.random.default_rng(seed)
    n_attr = int(rng.integers(2, 13))
    src = []
    for i in range(n_attr):
        fam = rng.integers(0, 10)
        dt = SC[rng.integers(0, 10)] if fam < 6 else (V3[rng.integers(0, 5)] if fam < 9 else OPAQUE[rng.integers(0, 3)])
        src.append(PointAttributeDefinition(f"a{i}", dt))
    if not any(a.datatype() == T.Vec3f64 for a in src) and rng.random() < 0.7:  # most point clouds carry POSITION_3D
        src[int(rng.integers(0, n_attr))] = PointAttributeDefinition(src[0].name() if False else f"a{n_attr}", T.Vec3f64)

    def other(dt):
        if dt in SC:
            return SC[rng.integers(0, 10)]
        if dt in V3:
            return V3[rng.integers(0, 5)]
        return dt
    order = rng.permutation(len(src))
    tgt = [PointAttributeDefinition(src[i].name(), other(src[i].datatype()) if rng.random() < 0.33 else src[i].datatype()) for i in order]
    return PointLayout.from_attributes_packed(src, 1), PointLayout.from_attributes_packed(tgt, 1)


def moved_bytes(sl, tl, src_cols, dst_cols):
    r = sum(a.size() for a in sl.attributes()) if src_cols else sl.size_of_point_entry()
    w = sum(a.size() for a in tl.attributes()) if dst_cols else tl.size_of_point_entry()
    return r + w


def dev_bytes(ptr, nbytes):
    class _Mem:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Mem(), device="cuda")


def raw_bytes(buf):
    """All bytes of a buffer as device tensors (for equality checks)."""
    n = buf.len()
    if isinstance(buf, pa.VectorBuffer):
        return [dev_bytes(buf.points_ptr(), n * buf.point_layout().size_of_point_entry())]
    return [dev_bytes(buf.column_ptr(a.attribute_definition()), n * a.size()) for a in buf.point_layout().attributes()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--skip-interp", action="store_true", help="time only the specialised kernels (no comparison run)")
    ap.add_argument("--pairings", default="VH,HV,VV")
    args = ap.parse_args()
    api = pa.product_api()
    s = torch.cuda.current_stream()
    api.set_stream(ctypes.c_void_p(s.cuda_stream))
    n = args.points
    r = range(0, n)
    out = open(args.out, "a") if args.out else None
    worst = {}
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        sl, tl = random_layouts(seed)
        for pairing in args.pairings.split(","):
            ST = pa.VectorBuffer if pairing[0] == "V" else pa.HashMapBuffer
            DT = pa.VectorBuffer if pairing[1] == "V" else pa.HashMapBuffer
            src = ST.new_from_layout(sl)
            src.resize(n)
            src.synth_fill(1000 + seed, 0)
            conv = pa.BufferLayoutConverter.for_layouts(sl, tl)
            res = {}
            outs = {}
            for mode in (("sync",) if args.skip_interp else ("off", "sync")):
                cv.jit_set_mode(mode)
                dst = DT.new_from_layout(tl)
                dst.resize(n)
                t0 = time.time()
                kind = conv.prepare(ST, DT) if mode == "sync" else 0
                prep = time.time() - t0
                for _ in range(2):
                    conv.convert_into_range_async(src, r, dst, r)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(args.steps):
                    conv.convert_into_range_async(src, r, dst, r)
                e1.record(s)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.steps
                res[mode] = (ms, cv.last_plan_kinds(), prep)
                outs[mode] = dst
            if args.skip_interp:
                res["off"], outs["off"] = (float("inf"), [], 0.0), outs["sync"]
            same = all(torch.equal(a, b) for a, b in zip(raw_bytes(outs["off"]), raw_bytes(outs["sync"])))
            b = moved_bytes(sl, tl, pairing[0] == "H", pairing[1] == "H")
            line = {"seed": seed, "pairing": pairing, "n_attrs": len(sl.attributes()), "src_record": sl.size_of_point_entry(),
                    "dst_record": tl.size_of_point_entry(), "bytes_per_point": b, "points": n,
                    "interpreted_ms": round(res["off"][0], 4), "interpreted_frac": round(b * n / (res["off"][0] * 1e-3) / PEAK, 4),
                    "interpreted_plan": res["off"][1], "jit_ms": round(res["sync"][0], 4),
                    "jit_frac": round(b * n / (res["sync"][0] * 1e-3) / PEAK, 4), "jit_plan": res["sync"][1],
                    "compile_s": round(res["sync"][2], 2), "identical": bool(same)}
            print(json.dumps(line), flush=True)
            if out:
                out.write(json.dumps(line) + "\n")
                out.flush()
            k = pairing
            worst[k] = min(worst.get(k, 9), line["jit_frac"])
            del src, outs, dst
    print("worst jit_frac per pairing:", worst, "jit stats:", cv.jit_stats())


if __name__ == "__main__":
    main()
