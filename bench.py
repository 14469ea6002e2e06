#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on N GPUs of one node (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic points already resident in HBM:

  default workload `convert_affine_bounds` (BASELINE.json configs[1], the configuration the metric is quoted on):
      10^8 points per GPU, columnar (HashMapBuffer) POSITION_3D Vec3f64
      -> BufferLayoutConverter with the LAS affine transformation (p*scale)+offset -> columnar POSITION_3D
      + calculate_bounds of the result, fused into ONE pass over HBM (24 B read + 24 B written per point).
      N > 1: points shard by index range; the only exchange is one all-reduce (RCCL) of the 6-double AABB record per step.
      Two sharding modes: weak scaling (default: `--points` per GPU, global cloud of N x points) and BASELINE.json
      configs[3] (`--global-points 1000000000`: ONE fixed cloud sharded N ways, rank r owns
      [r*ceil(G/N), min(G, (r+1)*ceil(G/N))), "scaling": "strong").
      `python bench.py --gpus N` without a launcher starts the N ranks itself (re-exec under torch.distributed.run) and
      refuses to run when the box has fewer than N devices; under the driver's torchrun it uses the environment's ranks.
  other workloads (`--workload`): bounds (AABB only, 24 B/pt), las0_to_columns (configs[2]: 35 B interleaved LAS-0 ->
      10 columns, 70 B/pt), rawlas_to_columns (20 B raw records -> 10 columns with i32->f64 affine + bit fields, 55 B/pt).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
The CPU baseline is the oracle (oracle/: faithful-shape single-thread restatement of the reference's Rust loops — the
reference itself cannot be built offline) timed on the GPU box's host, rank 0, N = 1 only, on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling
SCALE, OFFSET = (0.001, 0.001, 0.001), (500000.0, 5400000.0, 100.0)
SEED = 42

WORKLOADS = {
    # name: (algorithmic bytes per point, description)
    "convert_affine_bounds": (48, "SoA POSITION_3D f64 affine layout-conversion + AABB, fused (24 R + 24 W)"),
    "bounds": (24, "SoA POSITION_3D f64 AABB (24 R)"),
    "las0_to_columns": (70, "configs[2]: AoS LAS format-0 (35 B, 10 attrs, packed) -> 10 SoA columns (35 R + 35 W)"),
    "las0_to_columns_bounds": (70, "AoS LAS format-0 -> 10 SoA columns + AABB of the result, fused (35 R + 35 W)"),
    "rawlas_to_columns": (55, "raw LAS-0 records (20 B) -> 10 SoA columns, i32->f64 affine + bit fields (20 R + 35 W)"),
    "rawlas_to_columns_bounds": (55, "raw LAS-0 records -> 10 SoA columns + AABB of the result, fused (20 R + 35 W)"),
    "rawlas_to_records": (55, "raw LAS-0 records (20 B) -> interleaved typed LAS-0 records (VectorBuffer of LasPointFormat0, 35 B): 20 R + 35 W"),
    "columns_to_las0": (70, "10 SoA columns -> AoS LAS format-0 (35 R + 35 W)"),
    "columns_to_custom41": (82, "CustomPointTypeBig (test_utils.rs:19-31: 41 B, 5 attrs, packed) HashMapBuffer -> VectorBuffer (41 R + 41 W); "
                                "generic converter: config.plan says which kernel family ran (--plan interpreted | specialised)"),
    "las1_records_to_custom27": (70, "typed LAS-1 records (43 B) -> packed records {Position3D, Intensity, Classification} (27 B), VectorBuffer -> "
                                     "VectorBuffer: 43 R + 27 W; generic converter, config.plan says which kernel family ran"),
    "randomlayout_records_to_columns": (None, "random packed layout (--layout-seed; tools/exp_jit_layouts.py random_layouts: 2-12 attributes of every "
                                              "datatype, target = the same attributes permuted with a third of the datatypes changed) VectorBuffer -> HashMapBuffer"),
    "randomlayout_columns_to_records": (None, "the same random layouts HashMapBuffer -> VectorBuffer"),
    "randomlayout_records_to_records": (None, "the same random layouts VectorBuffer -> VectorBuffer"),
    "benchlayout_records_to_columns": (60, "layout_conversion_bench.rs: PointTypeSource (35 B packed) VectorBuffer -> PointTypeTarget (25 B) HashMapBuffer, "
                                           "three `as` casts (Vec3f64->Vec3f32, u8->u32, u16->u8): 35 R + 25 W"),
    "benchlayout_columns_to_records": (60, "the same conversion HashMapBuffer -> VectorBuffer"),
    "benchlayout_records_to_records": (60, "the same conversion VectorBuffer -> VectorBuffer"),
    "las0_encode": (55, "LAS writer: 10 SoA columns (typed LAS-0) -> raw LAS-0 records + header AABB + per-return counts, fused (35 R + 20 W)"),
    "filter_big_columnar": (63.5, "HashMapBuffer::filter_into, CustomPointTypeBig (41 B, 5 attrs) columnar -> columnar, random mask density 0.5 "
                                  "resident in HBM (2 mask reads + 41 R + 20.5 W per input point)"),
    "filter_big_interleaved": (63.5, "buffer_filter_bench: HashMapBuffer::filter_into, CustomPointTypeBig columnar -> VectorBuffer, density 0.5"),
    "filter_las0_columnar": (54.5, "HashMapBuffer::filter_into, typed LAS-0 points (35 B, 10 attrs) columnar -> columnar, density 0.5 (2 mask reads + 35 R + 17.5 W); "
                                   "config.plan says which kernel family ran (--plan interpreted = the gather kernels)"),
    "filter_las0_interleaved": (54.5, "the same into a VectorBuffer of LasPointFormat0"),
    "filter_las3_columnar": (75.5, "typed LAS-3 points (49 B, 12 attrs) columnar -> columnar, density 0.5"),
    "filter_las3_interleaved": (75.5, "the same into a VectorBuffer of LasPointFormat3"),
    "voxelgrid_xyz": (24, "voxelgrid_filter, columnar POSITION_3D, leaf 2.5 (about 15 points per voxel): keys + radix sort + run-length + "
                          "per-voxel sequential centroid sums (sort-bound; 24 B/pt is only the unavoidable read)"),
    "voxelgrid_xyz_async": (24, "the same through pst_voxelgrid_filter_async (round 4): planned once, then bounds + markers + keys + sort + run heads + "
                               "reduction stream-ordered, no host round trip, no allocation (24 R lower bound)"),
    "narrow_f64_f32": (36, "SoA POSITION_3D Vec3f64 -> Vec3f32 `as` narrowing (24 R + 12 W)"),
    "normals_knn16_async": (44, "configs[4] through pst_compute_normals_into_async (round 4): planned once, then keys + sort + permutation + directory + box "
                               "search + exact search of the hand-backs stream-ordered, no host round trip, no measurement passes"),
    "normals_knn16": (44, "configs[4]: kNN(k=16) normal estimation, NORMAL Vec3f32 + curvature f64 written to columns "
                          "(lower-bound traffic 24 R + 12 W + 8 W; the search itself is latency/compute-bound)"),
    "normals_knn16_sheet": (44, "the same on a LiDAR-like sheet (a noisy 2-D manifold z = f(x, y) in a 3-D box, 23 % of the box occupied) with 0.001 % "
                                "stray points far above and below it: measured scale, trimmed box, box search, exact search of the strays"),
}
# typed LAS points of the other formats (sizes of LasPointFormatN::layout(), las_types.rs): compaction at density 0.5 = 2 mask reads + S R + S/2 W
for _f, _s in ((1, 43), (2, 41), (4, 72), (5, 78), (6, 46), (7, 52), (8, 54), (9, 75)):
    for _t in ("columnar", "interleaved"):
        WORKLOADS[f"filter_las{_f}_{_t}"] = (2 + 1.5 * _s, f"HashMapBuffer::filter_into, typed LAS-{_f} points ({_s} B) columnar -> {_t}, density 0.5"
                                             + (" (waveform packet attributes: no in-tree kernel, compiled at run time before the timed region)" if _f in (4, 5, 9) else ""))


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--points", type=int, default=100_000_000, help="points per GPU (weak scaling)")
    p.add_argument("--global-points", type=int, default=0,
                   help="BASELINE.json configs[3]: ONE cloud of this many points sharded by index range over the ranks (strong scaling); overrides --points")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only with --launch-dry-run (CPU test of the launcher)")
    p.add_argument("--launch-dry-run", action="store_true",
                   help="start the ranks, count them with one all-reduce, print the census and exit: no GPU work (CPU test of the N>1 launcher)")
    p.add_argument("--collective", default="capi", choices=["capi", "torch"],
                   help="N > 1: the AABB exchange per step. capi (default) = the boundary's own pst_bounds_allreduce (pst_comm_init_rank bootstrapped from "
                        "the launcher's rendezvous); torch = torch.distributed.all_reduce")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="N > 1 on a ONE-GPU box: every rank uses GPU 0, the ranks talk over gloo (torch's collectives; the C ABI's RCCL collective needs a GPU "
                        "per rank).  Exercises the sharding, the per-step exchange path, the in-run self-check and the configs[3] leg with real kernels; "
                        "the ranks share the GPU, so the timings mean nothing and the line says so")
    p.add_argument("--no-configs3", action="store_true", help="N > 1: skip the configs[3] leg (ONE 10^9-point cloud sharded over the ranks) appended to the weak-scaling line")
    p.add_argument("--configs3-points", type=int, default=1_000_000_000)
    p.add_argument("--no-north-star", action="store_true", help="skip the 10^9-point single-GPU leg (north_star size) appended at N=1")
    p.add_argument("--north-star-points", type=int, default=1_000_000_000)
    p.add_argument("--workload", default="convert_affine_bounds", choices=sorted(WORKLOADS))
    p.add_argument("--plan", default="auto", choices=["auto", "interpreted", "specialised"],
                   help="generic conversion workloads: auto / specialised = the plan-specialised kernel (in-tree instantiation or hipRTC, compiled before the "
                        "timed region with pst_converter_prepare); interpreted = the generic tile kernels interpreting the mapping list (PST_JIT=0)")
    p.add_argument("--layout-seed", type=int, default=0, help="randomlayout_* workloads: which random layout pair")
    p.add_argument("--no-extra-legs", action="store_true",
                   help="skip the configs[2] (LAS-0 records -> 10 columns) and configs[4] (kNN(16) normals) legs appended at N=1 to the default workload's line")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-traffic-run", action="store_true", help="do not re-run the workload under rocprofv3 for roofline.traffic (the committed profiles/hbm_traffic.json figure is quoted instead)")
    p.add_argument("--cpu-sample-points", type=int, default=100_000_000, help="CPU baseline sample (default: the whole 10^8-point workload, ~10 s of CPU work)")
    return p.parse_args()


def _native_oracle():
    """The oracle built for THIS host (`-O3 -march=native`, on the GPU box): oracle/_native/liboracle_native.so,
    compiled here when g++ is present (about 8 s, cached by mtime).  Falls back to the travelling -march=x86-64-v2 build.
    Returns (path, flags-description) or (None, None)."""
    import shutil
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    generic = os.path.join(odir, "liboracle.so")
    srcs = [os.path.join(odir, f) for f in ("oracle_capi.cpp", "oracle_capi.h", "pasture_oracle.hpp")]
    gxx = shutil.which("g++")
    if gxx and all(os.path.exists(f) for f in srcs):
        out = os.path.join(odir, "_native", "liboracle_native.so")
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            stale = not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in srcs)
            if stale:
                subprocess.check_call([gxx, "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out + ".tmp", srcs[0]],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
                os.replace(out + ".tmp", out)
            return out, "g++ -O3 -march=native, built on this host"
        except Exception:
            pass
    if os.path.exists(generic):
        return generic, "g++ -O3 -march=x86-64-v2 (no g++ on this host for a native build)" if not gxx else "g++ -O3 -march=x86-64-v2 (native build failed)"
    return None, None


def _torch_bytes(ptr, nbytes):
    """A uint8 torch view of `nbytes` of device memory at `ptr` (no copy)."""
    import torch

    class _Mem:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Mem(), device="cuda")


def _family_report(conv, dst_type, with_bounds, src_type=None):
    """What the converter measured on this device for plans two kernel families can serve (pst_converter_family_choice): the first call of
    >= 2^22 points through a synchronous entry point (or measure_families) times the LAS-format kernels against the plan-specialised one, median of three
    passes each, and keeps the faster -- the choice config.plan then shows."""
    try:
        choice, ms = conv.family_choice(dst_type, with_bounds, src_type)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    names = {-1: "not measured", 0: "las", 1: "plan-specialised", 2: "one family only"}
    rep = {"choice": names.get(choice, str(choice))}
    if choice in (0, 1):
        rep["ms_per_pass"] = {"las": round(ms[0], 4), "plan-specialised": round(ms[1], 4)}
    return rep


def measure_traffic_in_run(workload, n, kernel_substr, timeout_s=150):
    """roofline.traffic measured IN this run (round-5 review, weak #9): the same workload re-executed under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes: the two counters do not fit one; no other trace domain with --pmc), 3 steps each, in child processes; per launch of the
    dominant kernel FETCH_SIZE x 2 (lane-contiguous streaming reads: the counter reports half the bytes, profiles/pmc_calibration.json) + WRITE_SIZE, KiB x 1024
    -- the correction of /opt/skills/guides/MI355X_MICROARCH.md's HBM section as tools/rocprof_summary.py applies it.  Returns (bytes, source) or None."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pst_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "b", "--", sys.executable, os.path.abspath(__file__), "--workload", workload, "--points", str(n),
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-north-star", "--no-extra-legs", "--no-traffic-run"]
            subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = list(cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ? group by kernel_name",
                                    (counter, f"%{kernel_substr}%")))
            if not rows:
                return None
            vals[counter] = max(rows, key=lambda r: r[2])[1]
        except Exception:  # noqa: BLE001  (no profiler, no counters, a time-out: the committed figure stands in, and the line says which it is)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = 2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    return round(total), ("measured IN THIS RUN: child processes of this bench.py under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, 3 steps each); "
                          f"per launch of {kernel_substr}: FETCH_SIZE x 2 (streaming reads: counter reports half, profiles/pmc_calibration.json) + WRITE_SIZE, KiB x 1024")


def measure_valu_in_run(workload, n, kernel_substr="knn_tile2_kernel", timeout_s=240):
    """bound_valu measured IN this run: the kNN workload once more in a child process under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU` (its own pass, no
    other trace domain); vector wave-instructions per launch of the box kernel and that kernel's name.  None when the profiler or the counter is not to be had."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="pst_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "-d", d, "-o", "b", "--", sys.executable, os.path.abspath(__file__), "--workload", workload, "--points", str(n),
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-north-star", "--no-extra-legs", "--no-traffic-run"]
        subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        cur = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]).cursor()
        rows = list(cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name='SQ_INSTS_VALU' and kernel_name like ? group by kernel_name",
                                (f"%{kernel_substr}%",)))
        if not rows:
            return None
        name, val, _cnt = max(rows, key=lambda r: r[1])
        return {"kernel": name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], "valu_wave_instructions_per_launch": round(val)}
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def leg_configs2(pa, las, cv, torch, stream, n, seed):
    """BASELINE.json configs[2]: n typed LAS-0 points (35 B, 10 attributes, packed) VectorBuffer -> HashMapBuffer of 10 columns, 70 B/point,
    HIP events around each of 10 steps.  Returns (report, sample): sample = the first 10^5 points of every column as bytes, for the oracle check."""
    src_layout = las.point_layout_from_las_point_format(las.Format(0), False)
    src = pa.VectorBuffer.new_from_layout(src_layout)
    src.resize(n)
    src.synth_fill(seed, 0)
    dst = pa.HashMapBuffer.new_from_layout(src_layout)
    dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(src_layout, src_layout)
    plan = cv.PLAN_NAMES[conv.prepare(type(src), type(dst), False)]
    conv.measure_families(src, dst, False)  # (the stream-ordered calls below never measure: once, before the loop)
    steps = 10
    for _ in range(2):
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in ev:
        e0.record(stream)
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
        e1.record(stream)
    torch.cuda.synchronize()
    kinds = cv.last_plan_kinds()
    family = _family_report(conv, type(dst), False, type(src))
    ms_all = [a.elapsed_time(b) for a, b in ev]
    ms = sum(ms_all) / steps
    gbs = 70 * n / (ms * 1e-3) / 1e9
    m = min(n, 100_000)
    sample = {a.name(): dst.get_attribute_range(a.attribute_definition(), range(0, m)).tobytes() for a in src_layout.attributes()}
    report = {"points": n, "steps": steps, "ms_per_step": round(ms, 4), "ms_per_step_min": round(min(ms_all), 4), "value": round(n / (ms * 1e-3) / 1e6, 2), "unit": "Mpoints/s",
              "algorithmic_bytes_per_point": 70, "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "plan": kinds, "plan_prepared": plan, "family_measured": family,
              "note": "BASELINE.json configs[2]: typed LAS-0 records (35 B, 10 attributes) VectorBuffer -> 10 columns HashMapBuffer, 35 R + 35 W per point; HIP events "
                      "around each of 10 steps after the timed region of the headline"}
    return report, {"points": m, "seed": seed, "columns": sample}


def _sheet_cloud(torch, n, seed):
    """The LiDAR-shaped input of the kNN legs: a noisy 2-D manifold z = f(x, y) in a 3-D box (23 % of the box occupied) with 0.001 % stray points far
    above and below it -- compute_normals is run on scans, not on uniform boxes (normal_estimation.rs:79-130)."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    xy = torch.rand(n, 2, device="cuda", dtype=torch.float64, generator=g) * 1000.0
    z = 10.0 * torch.sin(xy[:, 0] / 50.0) * torch.cos(xy[:, 1] / 80.0) + 50.0 + 0.02 * torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    sheet = torch.cat([xy, z[:, None]], dim=1).contiguous()
    n_stray = max(1, n // 100000)
    sheet[torch.randint(0, n, (n_stray,), device="cuda", generator=g), 2] = (torch.rand(n_stray, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 6000.0
    return sheet


def leg_configs4(pa, torch, stream, n, seed, n_queries=48, sheet=False, pmc=True):
    """BASELINE.json configs[4]: kNN(k = 16) normal estimation over n uniform points (sheet: over the LiDAR-shaped sheet), NORMAL (Vec3f32) + Curvature (F64) written to columns.
    Timed: the synchronous call (wall clock around call + synchronize) and the planned stream-ordered form (HIP events).  Then, outside the timing:
    the raw f64 results + neighbour lists once more (pst_compute_normals_device), `n_queries` sampled neighbour lists against a brute force over all
    n points on the device, and the columns against the f64 results narrowed with `as`.  Returns (report, sample for the oracle's plane fit)."""
    from pasture_amd.algorithms import NormalsPlan, compute_normals_device, compute_normals_into
    from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, attributes as A
    k = 16
    layout = pa.PointLayout.from_attributes([A.POSITION_3D])
    cloud = None
    if sheet:
        cloud = _sheet_cloud(torch, n, seed)
        src = pa.ExternalColumnsBuffer([cloud], layout, n)
    else:
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(seed, 0)
    curv_def = PointAttributeDefinition("Curvature", T.F64)
    dst = pa.HashMapBuffer.new_from_layout(pa.PointLayout.from_attributes([A.NORMAL, curv_def]))
    dst.resize(n)
    compute_normals_into(src, k, dst)  # warm-up: scratch allocation, measurement passes
    torch.cuda.synchronize()
    sync_ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        compute_normals_into(src, k, dst)
        torch.cuda.synchronize()
        sync_ms.append((time.perf_counter() - t0) * 1e3)
    plan_ms, plan_status = None, None
    try:
        nplan = NormalsPlan(src, k, dst)
        nst = torch.zeros(2, dtype=torch.int64, device="cuda")
        nplan.compute_into_async(src, dst, nst.data_ptr())
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for e0, e1 in ev:
            e0.record(stream)
            nplan.compute_into_async(src, dst, nst.data_ptr())
            e1.record(stream)
        torch.cuda.synchronize()
        plan_ms = [a.elapsed_time(b) for a, b in ev]
        plan_status = nst.tolist()
        nplan.destroy()
    except Exception as e:  # noqa: BLE001  (no plan for this cloud: the line says so)
        plan_status = f"{type(e).__name__}: {e}"[:300]
    # -- checks, outside the timing
    normals = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    curv = torch.empty(n, dtype=torch.float64, device="cuda")
    knn = torch.empty((n, k), dtype=torch.int32, device="cuda")  # uint32 bit patterns; n < 2^31
    compute_normals_device(src, k, normals.data_ptr(), curv.data_ptr(), knn.data_ptr())
    pts = cloud if sheet else _torch_bytes(src.column_ptr(A.POSITION_3D), n * 24).view(torch.float64).view(n, 3)
    n32 = _torch_bytes(dst.column_ptr(A.NORMAL), n * 12).view(torch.float32).view(n, 3)
    c64 = _torch_bytes(dst.column_ptr(curv_def), n * 8).view(torch.float64)
    columns_equal = bool(torch.equal(n32, normals.to(torch.float32))) and bool(torch.equal(c64, curv))
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    queries = torch.randint(0, n, (n_queries,), generator=g).tolist()
    lists_exact, fits = 0, []
    for q in queries:
        d = ((pts - pts[q]) ** 2).sum(dim=1)
        dist, _want = torch.topk(d, k, largest=False, sorted=True)
        got = knn[q].long()
        dg = ((pts[got] - pts[q]) ** 2).sum(dim=1)
        lists_exact += int(bool((dg == dist).all()) and int(got[0]) == q)
        fits.append({"query": q, "neighbours": pts[got].cpu().numpy(), "normal": normals[q].cpu().numpy(), "curvature": float(curv[q])})
        del d
    med = statistics.median(sync_ms)
    report = {"points": n, "k": k, "ms_per_call": round(med, 3), "ms_per_call_all": [round(x, 3) for x in sync_ms], "value": round(n / (med * 1e-3) / 1e6, 2), "unit": "Mpoints/s",
              "ms_per_call_planned": round(sum(plan_ms) / len(plan_ms), 3) if plan_ms else None, "planned_status": plan_status,
              "algorithmic_bytes_per_point": 44, "frac_of_hbm_lower_bound": round(44 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              "checks": {"columns_equal_f64_results_narrowed": columns_equal, "neighbour_lists_checked": n_queries, "neighbour_lists_exact": lists_exact},
              "note": ("the reference's caller shape of configs[4]: a LiDAR-like sheet (noisy 2-D manifold in a 3-D box, 0.001 % far strays)" if sheet else "BASELINE.json configs[4]: uniform cloud") +
                      ", k = 16, NORMAL (Vec3f32) + Curvature (F64) columns; ms_per_call = median wall time of 3 synchronous "
                      "pst_compute_normals_into calls (index build, sort, searches, fits; host round trips included), ms_per_call_planned = HIP events around the "
                      "stream-ordered replay; the call is bound by its vector instructions, not by HBM (bound_valu)"}
    try:
        wl = "normals_knn16_sheet" if sheet else "normals_knn16"
        live = None if (sheet or not pmc) else measure_valu_in_run(wl, n)  # (the uniform leg: ~10 s of a child process; the sheet quotes its committed pass)
        if live is not None:
            issue_rate = 256 * 4 * 2.4e9 / 4.0
            bound_ms = live["valu_wave_instructions_per_launch"] / issue_rate * 1e3
            report["bound_valu"] = {**live, "bound_ms": round(bound_ms, 3), "frac_of_call": round(bound_ms / med, 4),
                                    "source": "measured IN THIS RUN: a child process of this bench.py under rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU (2 calls); issue rate = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles"}
        v = None if live is not None else json.load(open(os.path.join(ROOT, "profiles", "knn_valu.json"))).get(wl)
        if v and v.get("points") == n:
            issue_rate = 256 * 4 * 2.4e9 / 4.0
            bound_ms = v["valu_wave_instructions_per_launch"] / issue_rate * 1e3
            report["bound_valu"] = {"kernel": v["kernel"], "valu_wave_instructions_per_launch": v["valu_wave_instructions_per_launch"], "bound_ms": round(bound_ms, 3),
                                    "frac_of_call": round(bound_ms / med, 4), "kernel_ms_in_that_profile": v.get("kernel_ms"),
                                    "source": f"profiles/knn_valu.json (round {v.get('round')}, tree {v.get('commit', '?')}: rocprofv3 --pmc SQ_INSTS_VALU + --kernel-trace passes); NOT measured in this run"}
    except Exception:
        pass
    return report, {"k": k, "fits": fits}


def leg_dropin(pa, cv, torch, stream, n, seed, fused_bounds):
    """What an UNMODIFIED pasture caller issues for the headline workload: BufferLayoutConverter::convert_into (the affine mapping on POSITION_3D), then
    calculate_bounds on the result (bounds.rs:11) -- two calls, 24 R + 24 W + 24 R = 72 B per point, against the fused entry point's 48.  HIP events around
    each pair of 10 steps; the AABB must be the fused run's, digit for digit."""
    from pasture_amd.conversion import Transform
    from pasture_amd.distributed import bounds_from_record
    from pasture_amd.layout import PointAttributeDataType as T, attributes as A
    layout = pa.PointLayout.from_attributes([A.POSITION_3D])
    src = pa.HashMapBuffer.new_from_layout(layout)
    src.resize(n)
    src.synth_fill(seed, 0)
    dst = pa.HashMapBuffer.new_from_layout(layout)
    dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
    conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, Transform.affine(T.Vec3f64, SCALE, OFFSET), False)
    rec = torch.empty(6, dtype=torch.float64, device="cuda")
    steps = 10

    def pair():
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
        pa.calculate_bounds_async(dst, rec.data_ptr())
    for _ in range(2):
        pair()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1, e2 in ev:
        e0.record(stream)
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
        e1.record(stream)
        pa.calculate_bounds_async(dst, rec.data_ptr())
        e2.record(stream)
    torch.cuda.synchronize()
    conv_ms = [a.elapsed_time(b) for a, b, _ in ev]
    bnd_ms = [b.elapsed_time(c) for _, b, c in ev]
    ms = (sum(conv_ms) + sum(bnd_ms)) / steps
    got = bounds_from_record(rec.cpu())
    # the synchronous drop-in calls themselves (host round trip and the record's read-back included): wall clock
    t0 = time.perf_counter()
    for _ in range(5):
        conv.convert_into(src, dst)
        sync_b = pa.calculate_bounds(dst)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    gbs = 72 * n / (ms * 1e-3) / 1e9
    return {"points": n, "steps": steps, "ms_per_step": round(ms, 4), "ms_convert_into": round(sum(conv_ms) / steps, 4), "ms_calculate_bounds": round(sum(bnd_ms) / steps, 4),
            "ms_per_step_synchronous_calls": round(wall, 4), "value": round(n / (ms * 1e-3) / 1e6, 2), "unit": "Mpoints/s", "algorithmic_bytes_per_point": 72,
            "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "plan": cv.last_plan_kinds(),
            "verified": bool(got == fused_bounds and (sync_b.min(), sync_b.max()) == tuple(fused_bounds)) if fused_bounds else None,
            "note": "the reference's caller shape of the headline: convert_into then calculate_bounds (bounds.rs:11), unfused -- 72 B per point; HIP events around each "
                    "call of 10 steps (stream-ordered forms), and the wall time of the synchronous pair; verified = both AABBs equal the fused run's (config.bounds)"}


def leg_chunked(pa, las, cv, torch, stream, seed):
    """pasture-io's LAS reader (raw_readers.rs:309-349): raw point records arrive in 1 MiB chunks and every chunk goes through convert_into_range into the
    caller's buffer.  Raw LAS-0 records (20 B) -> typed LAS-0 columns (35 B), 55 B per point, chunk = 1 MiB / 20 B = 52 428 points, 256 chunks: the synchronous
    call per chunk (what a drop-in issues: launch + wait) and the stream-ordered loop (one wait at the end); then single calls over chunk sizes from 2^4 to
    2^22 points -- the oracle times the same sizes on the host (cpu_baseline leg) and the line reports where the GPU call overtakes it."""
    raw = las.point_layout_from_las_point_format(las.Format(0), True)
    typed = las.point_layout_from_las_point_format(las.Format(0), False)
    chunk = (1 << 20) // 20
    n_chunks = 256
    total = chunk * n_chunks
    src = pa.VectorBuffer.new_from_layout(raw)
    src.resize(total)
    src.synth_fill(seed, 0)
    dst = pa.HashMapBuffer.new_from_layout(typed)
    dst.resize(total)
    conv = las.get_default_las_converter(raw, typed, SCALE, OFFSET)
    conv.prepare(type(src), type(dst), False)
    for c in range(4):
        conv.convert_into_range(src, range(c * chunk, (c + 1) * chunk), dst, range(c * chunk, (c + 1) * chunk))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(n_chunks):
        conv.convert_into_range(src, range(c * chunk, (c + 1) * chunk), dst, range(c * chunk, (c + 1) * chunk))
    sync_us = (time.perf_counter() - t0) / n_chunks * 1e6
    kinds = cv.last_plan_kinds()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(n_chunks):
        conv.convert_into_range_async(src, range(c * chunk, (c + 1) * chunk), dst, range(c * chunk, (c + 1) * chunk))
    torch.cuda.synchronize()
    async_us = (time.perf_counter() - t0) / n_chunks * 1e6
    sizes = [1 << b for b in range(4, 23, 2)] + [chunk]
    per_size = {}
    for m in sorted(sizes):
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            conv.convert_into_range(src, range(0, m), dst, range(0, m))
            ts.append((time.perf_counter() - t0) * 1e6)
        per_size[m] = round(statistics.median(ts), 2)
    m = min(chunk, 100_000)
    sample = {a.name(): dst.get_attribute_range(a.attribute_definition(), range(0, m)).tobytes() for a in typed.attributes()}
    rep = {"chunk_points": chunk, "chunk_bytes": chunk * 20, "chunks": n_chunks, "us_per_chunk_synchronous": round(sync_us, 2), "us_per_chunk_stream_ordered": round(async_us, 2),
           "value": round(chunk / sync_us, 2), "value_stream_ordered": round(chunk / async_us, 2), "unit": "Mpoints/s", "algorithmic_bytes_per_point": 55,
           "achieved_GBps_stream_ordered": round(55 * chunk / async_us / 1e3, 1), "plan": kinds, "us_per_call_by_points": {str(k): v for k, v in sorted(per_size.items())},
           "note": "the reference's production caller (raw_readers.rs:309-349): raw LAS-0 records -> typed LAS-0 columns in 1 MiB chunks through convert_into_range; wall clock "
                   "per chunk of 256 chunks, synchronous (launch + wait per chunk) and stream-ordered (one wait at the end); us_per_call_by_points = median of 9 synchronous calls"}
    return rep, {"points": m, "seed": seed, "columns": sample, "sizes": sorted(sizes)}


def oracle_spot_checks(lib_path, checks):
    """The oracle (test infrastructure) as the checker of the extra legs' samples -- called from the cpu_baseline leg only."""
    import numpy as np
    import pasture_amd as pa
    from pasture_amd import las
    from pasture_amd._capi import CApi
    from pasture_amd.algorithms import compute_normals
    from pasture_amd.layout import attributes as A
    orc = CApi(ctypes.CDLL(lib_path), "orc", product=False)
    out = {}
    c2 = checks.get("configs2")
    if c2:
        layout = las.point_layout_from_las_point_format(las.Format(0), False, api=orc)
        src = pa.VectorBuffer.new_from_layout(layout)
        src.resize(c2["points"])
        src.synth_fill(c2["seed"], 0)
        conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
        cols = conv.convert(src, pa.HashMapBuffer)
        bad = [a.name() for a in layout.attributes()
               if cols.get_attribute_range(a.attribute_definition(), range(0, c2["points"])).tobytes() != c2["columns"][a.name()]]
        out["configs2"] = {"verified": not bad, "points_compared": c2["points"], "columns_compared": len(c2["columns"]), "columns_differing": bad,
                           "how": "the first points of the GPU's 10 columns, byte for byte against the oracle's conversion of the same synthetic records"}
    ck = checks.get("chunked")
    if ck:
        raw = las.point_layout_from_las_point_format(las.Format(0), True, api=orc)
        typed = las.point_layout_from_las_point_format(las.Format(0), False, api=orc)
        biggest = max(ck["sizes"])
        src = pa.VectorBuffer.new_from_layout(raw)
        src.resize(biggest)
        src.synth_fill(ck["seed"], 0)
        dst = pa.HashMapBuffer.new_from_layout(typed)
        dst.resize(biggest)
        conv = las.get_default_las_converter(raw, typed, SCALE, OFFSET)
        per_size = {}
        for m in ck["sizes"]:
            ts = []
            for _ in range(3 if m > (1 << 20) else 7):
                t0 = time.perf_counter()
                conv.convert_into_range(src, range(0, m), dst, range(0, m))
                ts.append((time.perf_counter() - t0) * 1e6)
            per_size[m] = round(statistics.median(ts), 2)
        bad = [a.name() for a in typed.attributes()
               if dst.get_attribute_range(a.attribute_definition(), range(0, ck["points"])).tobytes() != ck["columns"][a.name()]]
        out["chunked"] = {"verified": not bad, "points_compared": ck["points"], "columns_differing": bad, "oracle_us_per_call_by_points": {str(k): v for k, v in sorted(per_size.items())},
                          "how": "the first chunk of the GPU's 10 columns byte for byte against the oracle's convert_into_range of the same raw records; the oracle's own "
                                 "call times (one pinned core, through the same ctypes binding) for the crossover"}
    for key in ("configs4", "configs4_sheet"):
        _spot_check_fits(out, key, checks.get(key), pa, orc, np, A, compute_normals)
    return out


def _spot_check_fits(out, key, c4, pa, orc, np, A, compute_normals):
    if c4:
        k, worst_n, worst_c, failed = c4["k"], 0.0, 0.0, 0
        for f in c4["fits"]:
            ob = pa.HashMapBuffer.new_from_layout(pa.PointLayout.from_attributes([A.POSITION_3D], api=orc))
            ob.resize(k)
            ob.set_attribute_range(A.POSITION_3D, range(0, k), f["neighbours"])
            on, oc = compute_normals(ob, k)  # the 16 neighbours in ascending distance as a 16-point cloud: the fit of point 0 is the full computation's
            nerr = float(np.linalg.norm(f["normal"] - on[0]) / max(np.linalg.norm(on[0]), 1e-300))
            dev = f["neighbours"] - f["neighbours"].mean(axis=0)
            floor = max(1e-12, 1e-13 * float(np.abs(dev.T @ dev).max()))  # the window of tests/test_gpu_parity.py::_compare_normals
            cdiff = abs(f["curvature"] - float(oc[0]))
            worst_n, worst_c = max(worst_n, nerr), max(worst_c, cdiff)
            failed += int(nerr > 1e-9 or cdiff > 1e-9 * abs(float(oc[0])) + floor)
        out[key] = {"verified": failed == 0, "fits_compared": len(c4["fits"]), "fits_outside_window": failed, "worst_rel_normal_diff": worst_n,
                    "worst_abs_curvature_diff": worst_c,
                    "how": "the oracle's plane fit of the GPU's own neighbour lists (16-point clouds, k = 16) against the GPU's f64 normals (1e-9 relative) "
                           "and curvatures (1e-9 relative + the documented floor)"}


def cpu_baseline(workload, sample_points, checks=None):
    """Oracle timed on one pinned host core.  Only the checker lives under oracle/; it is never the thing shipped."""
    lib_path, flags = _native_oracle()
    if lib_path is None:
        return None
    lib = ctypes.CDLL(lib_path)
    aff = None
    try:
        aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(aff)[0]})
    except Exception:
        pass
    try:
        reps = 5
        secs = (ctypes.c_double * reps)()
        bounds = (ctypes.c_double * 6)()
        sc, of = (ctypes.c_double * 3)(*SCALE), (ctypes.c_double * 3)(*OFFSET)
        lib.orc_bench_config2.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.orc_bench_config1.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        t0 = time.time()
        rc = lib.orc_bench_config2(sample_points, reps, SEED, sc, of, secs, bounds)
        if rc != 0:
            return None
        med = statistics.median(list(secs))
        out = {
            "value": round(sample_points / med / 1e6, 3), "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "sample": f"{sample_points} of the same synthetic points, same workload (columnar POSITION_3D affine convert + calculate_bounds), "
                      f"median of {reps} runs, single thread pinned to one core; oracle = faithful-shape C++ restatement of the Rust "
                      f"reference ({flags}), the reference itself is not buildable offline",
            "effective_GBps": round(48 * sample_points / med / 1e9, 3),
            "bounds": list(bounds),
        }
        # the same code at a tenth of the sample to show linearity
        lin_n = max(1, sample_points // 10)
        secs_l = (ctypes.c_double * reps)()
        if lib.orc_bench_config2(lin_n, reps, SEED, sc, of, secs_l, bounds) == 0:
            out["linearity"] = {"points": lin_n, "value": round(lin_n / statistics.median(list(secs_l)) / 1e6, 3), "unit": "Mpoints/s"}
        # BASELINE.json configs[0] exactly: 10^6 XYZ f64 points VectorBuffer -> HashMapBuffer + calculate_bounds, median of 10
        reps1 = 10
        secs1 = (ctypes.c_double * reps1)()
        if lib.orc_bench_config1(1_000_000, reps1, SEED, secs1, bounds) == 0:
            out["config0_1e6_aos_to_soa_plus_bounds_Mpoints_per_s"] = round(1.0 / statistics.median(list(secs1)), 3)
        out["wall_s"] = round(time.time() - t0, 2)
        if checks:
            t1 = time.time()
            try:
                out["spot_checks"] = oracle_spot_checks(lib_path, checks)
            except Exception as e:  # noqa: BLE001
                out["spot_checks"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            out["spot_checks_wall_s"] = round(time.time() - t1, 2)
        try:
            with open("/proc/cpuinfo") as f:
                models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
            out["host_cpu"] = models[0] if models else "unknown"
            out["host_logical_cores"] = os.cpu_count()
        except Exception:
            pass
        return out
    finally:
        if aff is not None:
            try:
                os.sched_setaffinity(0, aff)
            except Exception:
                pass


def _free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks of this file under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and hand back their exit code.  Refuses when the box has fewer than N devices."""
    import subprocess
    if args.backend == "nccl" and not args.rehearse_on_one_gpu:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}; refusing to run fewer ranks "
                             f"than asked (no silent 1-GPU run)\n")
            sys.exit(2)
    # --standalone: the launcher's own agent picks AND HOLDS the rendezvous port.  (Choosing a free port here and passing --master-port left a
    # window of seconds -- python start-up, import torch -- in which any outgoing connection of the box could take that port as its source port:
    # the one unexplained multi-rank failure in five full suites of round 4.)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus == 1:
            args.gpus = world  # launched under torchrun without --gpus: the launcher's world size is the truth
        else:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
            sys.exit(2)
    if args.launch_dry_run:
        # CPU test of the launcher: every rank joins ONE all-reduce; the sum is the number of workers that really started
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(args.backend if args.backend == "gloo" or torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        if dev == "cuda":
            torch.cuda.set_device(local_rank)
        census = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(census)
        from pasture_amd.distributed import shard_range
        g_pts = args.global_points or args.points * world
        shard = shard_range(g_pts, rank, world) if args.global_points else range(rank * args.points, (rank + 1) * args.points)
        mine = torch.tensor([shard.start, shard.stop, os.getpid()], dtype=torch.int64, device=dev)
        allr = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(allr, mine)
        if rank == 0:
            print(json.dumps({"launch_dry_run": True, "n_gpus": int(census.item()), "world_size": world, "backend": dist.get_backend(),
                              "scaling": "strong" if args.global_points else "weak", "global_points": g_pts,
                              "shards": [[int(t[0]), int(t[1])] for t in allr], "pids": [int(t[2]) for t in allr]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU: pasture_amd has no CPU fallback"
    if args.rehearse_on_one_gpu:
        local_rank = 0
        args.collective = "torch"
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} wants device {local_rank} but this node has {torch.cuda.device_count()} GPUs\n")
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    # torchrun with one rank (or PASTURE_FORCE_DIST=1) still exercises the RCCL path: init, all-reduce, barrier
    distributed = world > 1 or (os.environ.get("PASTURE_FORCE_DIST") == "1" and "RANK" in os.environ)
    n_ranks_seen = 1
    ctl = "cpu" if args.rehearse_on_one_gpu else "cuda"  # where the run's own control tensors live (gloo gathers on the host)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the 48-byte AABB collectives must not queue behind the 97k-workgroup conversion kernels: high-priority RCCL stream
        # (measured with one rank: step 0.806 -> 0.774 ms, kernel-only 0.767 ms; tools/exp_dist_overhead.py)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        census = torch.ones(1, dtype=torch.int64, device=ctl)
        dist.all_reduce(census)  # n_gpus in the JSON line = the ranks RCCL really saw
        n_ranks_seen = int(census.item())
        if n_ranks_seen != world:
            sys.stderr.write(f"bench.py: RCCL saw {n_ranks_seen} ranks, expected {world}\n")
            sys.exit(2)

    import pasture_amd as pa
    from pasture_amd import las
    from pasture_amd.distributed import allreduce_bounds_record, bounds_from_record
    from pasture_amd.layout import PointAttributeDataType as T, attributes as A

    api = pa.product_api()
    api.set_device(local_rank)
    stream = torch.cuda.current_stream()
    api.set_stream(ctypes.c_void_p(stream.cuda_stream))  # kernels run on torch's current stream => torch events see them
    from pasture_amd import conversion as cv
    if args.plan == "interpreted":
        cv.jit_set_mode("off")  # neither the in-tree instantiations nor the run-time compiler: every generic plan is interpreted
    conv = None  # generic-conversion workloads leave their converter here: prepared before the timed region, its kernel family reported

    if args.global_points:
        # BASELINE.json configs[3]: ONE cloud sharded by index range; the ranks' shards differ by at most one tile
        from pasture_amd.distributed import shard_range
        shard = shard_range(args.global_points, rank, world)
        n, first_index = len(shard), shard.start
        global_points, scaling = args.global_points, "strong"
    else:
        n = args.points
        first_index = rank * n  # weak scaling: every rank owns its own index range of one global synthetic cloud
        global_points, scaling = n * world, "weak"
    bytes_per_point, desc = WORKLOADS[args.workload]
    # ring of AABB records: step i writes ring[i % 4]; with N > 1 its all-reduce runs asynchronously behind the next steps
    from pasture_amd.distributed import BoundsExchange, CapiTransport, PipelinedBoundsReduce

    def make_rec():
        return torch.empty(6, dtype=torch.float64, device="cuda")

    transport, collective_note = None, None
    if distributed and args.collective == "capi":
        # the product's own collective: pst_comm_init_rank over the launcher's rendezvous, pst_bounds_allreduce per step.  If ANY rank cannot
        # set it up (RCCL not loadable through dlopen, say) every rank falls back to torch.distributed together -- and the line says so.
        err = None
        try:
            transport = CapiTransport(None, api)
            if transport.size() != world:
                err = f"pst_comm_size = {transport.size()}, expected {world}"
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        bad = torch.tensor([1 if err else 0], dtype=torch.int64, device=ctl)
        dist.all_reduce(bad)
        if int(bad.item()):
            if transport is not None:
                try:
                    transport.close()
                except Exception:  # noqa: BLE001
                    pass
            transport = None
            collective_note = f"--collective capi could not be set up on {int(bad.item())} of {world} ranks ({err or 'another rank failed'}): torch.distributed.all_reduce instead"
            if rank == 0:
                sys.stderr.write("bench.py: " + collective_note + "\n")
    if transport is not None:
        ring = BoundsExchange(make_rec, transport, depth=4)
    else:
        ring = PipelinedBoundsReduce(make_rec, depth=4)
    rec = ring.recs[0]

    def rec_ptr():
        return ring.current().data_ptr()

    after = None  # a workload may set a check to run after the timed region
    if args.workload in ("convert_affine_bounds", "bounds"):
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        if args.workload == "convert_affine_bounds":
            dst = pa.HashMapBuffer.new_from_layout(layout)
            dst.resize(n)
            conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
            conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, SCALE, OFFSET), False)

            def step():
                conv.convert_into_with_bounds_async(src, dst, rec_ptr())
        else:
            def step():
                pa.calculate_bounds_async(src, rec_ptr())
    elif args.workload == "narrow_f64_f32":
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        layout32 = pa.PointLayout.from_attributes([A.POSITION_3D.with_custom_datatype(T.Vec3f32)])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.HashMapBuffer.new_from_layout(layout32)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(layout, layout32)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload == "normals_knn16":
        from pasture_amd.layout import PointAttributeDefinition
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.HashMapBuffer.new_from_layout(pa.PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)]))
        dst.resize(n)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            pa.compute_normals_into(src, 16, dst)
    elif args.workload == "normals_knn16_async":
        from pasture_amd.algorithms import NormalsPlan
        from pasture_amd.layout import PointAttributeDefinition
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.HashMapBuffer.new_from_layout(pa.PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)]))
        dst.resize(n)
        pa.calculate_bounds_async(src, rec.data_ptr())
        nplan = NormalsPlan(src, 16, dst)
        nst = torch.zeros(2, dtype=torch.int64, device="cuda")

        def step():
            nplan.compute_into_async(src, dst, nst.data_ptr())

        def after():
            assert nst.tolist() == [0, 0], nst.tolist()
    elif args.workload == "normals_knn16_sheet":
        from pasture_amd.layout import PointAttributeDefinition
        g = torch.Generator(device="cuda")
        g.manual_seed(SEED + first_index)
        xy = torch.rand(n, 2, device="cuda", dtype=torch.float64, generator=g) * 1000.0
        z = 10.0 * torch.sin(xy[:, 0] / 50.0) * torch.cos(xy[:, 1] / 80.0) + 50.0 + 0.02 * torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
        sheet = torch.cat([xy, z[:, None]], dim=1).contiguous()
        n_stray = max(1, n // 100000)
        sheet[torch.randint(0, n, (n_stray,), device="cuda", generator=g), 2] = (torch.rand(n_stray, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 6000.0
        del xy, z
        src = pa.ExternalColumnsBuffer([sheet], pa.PointLayout.from_attributes([A.POSITION_3D]), n)
        dst = pa.HashMapBuffer.new_from_layout(pa.PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)]))
        dst.resize(n)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            pa.compute_normals_into(src, 16, dst)
    elif args.workload == "las0_encode":
        layout = las.point_layout_from_las_point_format(las.Format(0), False)
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.VectorBuffer.new_from_layout(las.point_layout_from_las_point_format(las.Format(0), True))
        dst.resize(n)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            las.encode_points(src, 0, (0.001, 0.001, 0.001), (0.0, 0.0, 0.0), dst)
    elif args.workload.startswith("filter_"):
        if args.workload.startswith("filter_big"):
            big = pa.PointLayout.from_attributes_packed([A.GPS_TIME, A.COLOR_RGB, A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY.with_custom_datatype(T.I16)], 1)
        else:
            big = las.point_layout_from_las_point_format(las.Format(int(args.workload[len("filter_las")])), False)
        src = pa.HashMapBuffer.new_from_layout(big)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        g = torch.Generator(device="cuda")
        g.manual_seed(SEED + rank)
        mask = (torch.rand(n, device="cuda", generator=g) < 0.5).to(torch.uint8)
        k = int(mask.sum().item())
        dst = (pa.VectorBuffer if args.workload.endswith("interleaved") else pa.HashMapBuffer).new_from_layout(big)
        dst.resize(k)
        pa.calculate_bounds_async(src, rec.data_ptr())

        # `Some(num_matches)` as the reference's bench passes it (buffer_filter_bench.rs:62-74): the stream-ordered form -- count, scan and the
        # copies of every step are on the stream, nothing waits on the host between steps; the hit count of the last step is checked below
        hits = torch.zeros(1, dtype=torch.int64, device="cuda")

        def step():
            src.filter_into_async(dst, mask.data_ptr(), k, hits.data_ptr())

        if args.plan != "interpreted":  # a layout without an in-tree streaming kernel: compiled now, before the timed region
            cv.jit_set_mode("sync")
            step()
            cv.jit_set_mode("env")

        def after():
            assert int(hits.item()) == k, (int(hits.item()), k)
    elif args.workload == "voxelgrid_xyz":
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            out = pa.HashMapBuffer.new_from_layout(layout)
            pa.voxelgrid_filter(src, 2.5, 2.5, 2.5, out)
    elif args.workload == "voxelgrid_xyz_async":
        from pasture_amd.algorithms import VoxelGridPlan
        layout = pa.PointLayout.from_attributes([A.POSITION_3D])
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        pa.calculate_bounds_async(src, rec.data_ptr())
        vplan = VoxelGridPlan(src, 2.5, 2.5, 2.5)
        vout = pa.HashMapBuffer.new_from_layout(layout)
        vout.resize(vplan.max_voxels)
        vcs = torch.zeros(2, dtype=torch.int64, device="cuda")

        def step():
            vplan.filter_async(src, vout, 0, vcs.data_ptr())

        def after():
            cnt, st = (int(x) for x in vcs.tolist())
            assert st == 0 and 0 < cnt <= vplan.max_voxels, (cnt, st)
    elif args.workload == "rawlas_to_records":
        src_layout = las.point_layout_from_las_point_format(las.Format(0), True)
        dst_layout = las.point_layout_from_las_point_format(las.Format(0), False)
        src = pa.VectorBuffer.new_from_layout(src_layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.VectorBuffer.new_from_layout(dst_layout)
        dst.resize(n)
        conv = las.get_default_las_converter(src_layout, dst_layout, SCALE, OFFSET)
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
        pa.calculate_bounds_async(dst, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload == "columns_to_custom41":
        big = pa.PointLayout.from_attributes_packed([A.GPS_TIME, A.COLOR_RGB, A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY.with_custom_datatype(T.I16)], 1)
        src = pa.HashMapBuffer.new_from_layout(big)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.VectorBuffer.new_from_layout(big)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(big, big)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload.startswith("benchlayout_"):
        src_layout = pa.PointLayout.from_attributes_packed([A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY, A.GPS_TIME], 1)
        dst_layout = pa.PointLayout.from_attributes_packed([A.GPS_TIME, A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.CLASSIFICATION.with_custom_datatype(T.U32),
                                                            A.INTENSITY.with_custom_datatype(T.U8)], 1)
        src_kind = pa.VectorBuffer if args.workload.split("_")[1] == "records" else pa.HashMapBuffer
        dst_kind = pa.VectorBuffer if args.workload.endswith("_records") else pa.HashMapBuffer
        src = src_kind.new_from_layout(src_layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = dst_kind.new_from_layout(dst_layout)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(src_layout, dst_layout)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload == "las1_records_to_custom27":
        src_layout = las.point_layout_from_las_point_format(las.Format(1), False)
        dst_layout = pa.PointLayout.from_attributes_packed([A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION], 1)
        src = pa.VectorBuffer.new_from_layout(src_layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.VectorBuffer.new_from_layout(dst_layout)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(src_layout, dst_layout)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload.startswith("randomlayout_"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from exp_jit_layouts import moved_bytes, random_layouts
        src_layout, dst_layout = random_layouts(args.layout_seed)
        src_kind = pa.VectorBuffer if args.workload.split("_")[1] == "records" else pa.HashMapBuffer
        dst_kind = pa.VectorBuffer if args.workload.endswith("_records") else pa.HashMapBuffer
        bytes_per_point = moved_bytes(src_layout, dst_layout, src_kind is pa.HashMapBuffer, dst_kind is pa.HashMapBuffer)
        desc += (f"; seed {args.layout_seed}: {len(src_layout.attributes())} attributes, records of {src_layout.size_of_point_entry()} -> "
                 f"{dst_layout.size_of_point_entry()} B, {bytes_per_point} B moved per point")
        src = src_kind.new_from_layout(src_layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = dst_kind.new_from_layout(dst_layout)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(src_layout, dst_layout)

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    elif args.workload == "columns_to_las0":
        layout = las.point_layout_from_las_point_format(las.Format(0), False)
        src = pa.HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.VectorBuffer.new_from_layout(layout)
        dst.resize(n)
        conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
        pa.calculate_bounds_async(src, rec.data_ptr())

        def step():
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    else:
        raw = args.workload.startswith("rawlas")
        with_bounds = args.workload.endswith("_bounds")
        src_layout = las.point_layout_from_las_point_format(las.Format(0), raw)
        dst_layout = las.point_layout_from_las_point_format(las.Format(0), False)
        src = pa.VectorBuffer.new_from_layout(src_layout)
        src.resize(n)
        src.synth_fill(SEED, first_index)
        dst = pa.HashMapBuffer.new_from_layout(dst_layout)
        dst.resize(n)
        conv = las.get_default_las_converter(src_layout, dst_layout, SCALE, OFFSET) if raw else \
            pa.BufferLayoutConverter.for_layouts(src_layout, dst_layout)
        if with_bounds:
            def step():
                conv.convert_into_with_bounds_async(src, dst, rec_ptr())
        else:
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
            pa.calculate_bounds_async(dst, rec.data_ptr())  # reported once in config.bounds; not part of the timed step

            def step():
                conv.convert_into_range_async(src, range(0, n), dst, range(0, n))

    has_reduction = args.workload in ("convert_affine_bounds", "bounds") or args.workload.endswith("_bounds")
    prepared_plan = None
    if conv is not None and args.plan != "interpreted":
        # the run-time compiler normally works on a background thread while the first calls are interpreted; a benchmark wants the steady state
        prepared_plan = cv.PLAN_NAMES[conv.prepare(type(src), type(dst), has_reduction)]
        _s, _d = locals().get("src"), locals().get("dst")
        if _s is not None and _d is not None and _s is not _d and hasattr(conv, "measure_families") and _s.len() == _d.len():
            conv.measure_families(_s, _d, has_reduction)  # (the `_async` conversions of the timed steps never measure: once, before the loop)

    def full_step():
        step()
        if has_reduction:
            if distributed:
                ring.submit()  # ONE all-reduce of 6 doubles (RCCL over xGMI), asynchronous: overlaps the next step
            else:
                ring.i += 1

    for _ in range(args.warmup):
        full_step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()

    # HIP-event pairs around a step's launches (roofline.achieved) on every tenth step -- at least five pairs per run.  An event is a marker packet between
    # the launches of consecutive steps: a pair around EVERY step cost 0.006 ms of wall time per 0.72 ms step (three runs each on one box, profiles/
    # r06_experiments.txt E14) -- instrumentation inside the region `value` is taken from.  PST_BENCH_EVENT_EVERY=1 is the old behaviour.
    every = max(1, int(os.environ.get("PST_BENCH_EVENT_EVERY", str(min(10, max(1, args.steps // 5))))))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    timed_steps = [i for i in range(args.steps) if i % every == 0]
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % every == 0:
            ev[i][0].record(stream)
        step()
        if i % every == 0:
            ev[i][1].record(stream)
        if has_reduction:
            if distributed:
                ring.submit(timed=(i == args.steps - 1))  # the last step's exchange is the one nothing hides: its exposed time is reported
            else:
                ring.i += 1
    # The closing bracket: this rank's K steps AND its exchanges are complete (exchange stream dry, device idle) -> its clock stops; then the barrier.
    # The MAX over ranks below is the moment the barrier releases minus the barrier's own latency (a collective + two host round trips: 2-3 ms once
    # per run, which is not work of the K steps -- at N = 1 there is none; with five short steps it was most of the measured time, tests/test_distributed_gloo.py).
    if distributed and has_reduction:
        ring.wait()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    final = ring.finish() if (distributed and has_reduction) else None  # (decoding the last record: outside the timed region)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [ev[i][0].elapsed_time(ev[i][1]) for i in timed_steps]
    kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
    # which kernel families the last step's conversion / compaction call launched, as the library reports it (pst_last_plan_kinds)
    plan_kinds = cv.last_plan_kinds() if (conv is not None or args.workload.startswith("filter_")) else None
    family_rep = None
    _src, _dst = locals().get("src"), locals().get("dst")
    if conv is not None and _dst is not None and _src is not None and hasattr(conv, "family_choice"):
        family_rep = _family_report(conv, type(_dst), has_reduction, type(_src))
    if after is not None:
        after()  # (a workload's own check of what its stream-ordered steps left behind; outside the timed region)
    per_rank = None
    if distributed:
        # per-rank kernel time and the exposed part of the last exchange, gathered so that rank 0's line shows every rank
        mine = torch.tensor([kernel_ms_avg, (ring.exposed_us() or -1.0) if has_reduction else -1.0], dtype=torch.float64, device=ctl)
        rows = [torch.zeros(2, dtype=torch.float64, device=ctl) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = {"kernel_ms_avg": [round(float(r[0]), 4) for r in rows],
                    "last_exchange_exposed_us": [round(float(r[1]), 1) if float(r[1]) >= 0 else None for r in rows]}
        kernel_ms_avg = max(float(r[0]) for r in rows)

    # Self-check of the sharded run (N > 1 has never been seen on hardware by the builder: the first run must not be silently wrong): the
    # all-reduced AABB of the LAST step must equal affine(union of the ranks' local source bounds) bit for bit on every rank -- the
    # local bounds are computed now, outside the timed region, and gathered through torch.distributed, not through the exchange under test.
    self_check = None
    if distributed and has_reduction and final is not None and args.workload in ("convert_affine_bounds", "bounds"):
        from pasture_amd.distributed import F64_MAX, verify_global_bounds
        src_rec = torch.tensor([F64_MAX] * 3 + [-F64_MAX] * 3, dtype=torch.float64, device="cuda")
        if n:
            pa.calculate_bounds_async(src, src_rec.data_ptr())
        torch.cuda.synchronize()
        affine = args.workload == "convert_affine_bounds"
        if os.environ.get("PASTURE_BENCH_FAULT") == str(rank):  # test hook: this rank holds a record that is off in the last bits
            final = final.clone()
            final[4] += 1e-9
        # a failed check does not swallow the line: it is printed with "verified": false and the error, and THEN every rank exits non-zero
        try:
            self_check = verify_global_bounds(src_rec, final, SCALE if affine else (1.0, 1.0, 1.0), OFFSET if affine else (0.0, 0.0, 0.0))
        except AssertionError as e:
            self_check = {"verified": False, "ranks": world, "error": str(e)[:2000]}
        self_check["comm_size"] = transport.size() if transport is not None else world
        if self_check["comm_size"] != world:
            self_check["verified"] = False
            self_check.setdefault("error", f"pst_comm_size = {self_check['comm_size']}, expected {world}")

    if final is not None:
        rec = final
    elif has_reduction and ring.i:
        rec = ring.recs[(ring.i - 1) % len(ring.recs)]
    if final is None and getattr(ring, "encoded", False):  # (the ring's records are kept as {min, -max}: pst_bounds_record_set_form)
        rec = rec.cpu()
        rec[3:] = -rec[3:]
    result = bounds_from_record(rec.cpu())
    # BASELINE.json configs[3] made driver-visible: a driver that passes only `--gpus N` gets the weak-scaling line above AND this leg --
    # ONE 10^9-point cloud sharded by index range over the N ranks (strong scaling), the same fused step, the same exchange per step,
    # timed like the main region (barrier + synchronize before; each rank stops its clock when its steps and exchanges are done, then the barrier; max over ranks); never folded into `value`
    configs3 = None
    # (a single rank started under torchrun with PASTURE_FORCE_DIST=1 runs this leg too: the way the N > 1 code path is exercised on a 1-GPU box)
    if (distributed and (world > 1 or os.environ.get("PASTURE_FORCE_DIST") == "1") and args.workload == "convert_affine_bounds"
            and not args.global_points and not args.no_configs3):
        from pasture_amd.distributed import shard_range
        g3 = args.configs3_points
        sh = shard_range(g3, rank, world)
        src = dst = None
        s_src = pa.HashMapBuffer.new_from_layout(layout)
        s_src.resize(len(sh))
        s_src.synth_fill(SEED, sh.start)
        s_dst = pa.HashMapBuffer.new_from_layout(layout)
        s_dst.resize(len(sh))
        ring3 = BoundsExchange(make_rec, transport, depth=4) if transport is not None else PipelinedBoundsReduce(make_rec, depth=4)
        c3_steps = 10

        def c3_step():
            conv.convert_into_with_bounds_async(s_src, s_dst, ring3.current().data_ptr())
            ring3.submit()
        for _ in range(2):
            c3_step()
        ring3.finish()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        # per-rank HIP events around each step's conversion launch (north_star-style): the first hardware run separates kernel time from
        # exchange / launch time without a second run -- wall time between the barriers minus the slowest rank's kernel time is what the 48-byte
        # all-reduce and the launch path cost
        c3_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(c3_steps)]
        t3 = time.perf_counter()
        for e0, e1 in c3_ev:
            e0.record(stream)
            conv.convert_into_with_bounds_async(s_src, s_dst, ring3.current().data_ptr())
            e1.record(stream)
            ring3.submit()
        ring3.wait()
        torch.cuda.synchronize()
        c3_mine = time.perf_counter() - t3  # (this rank's clock stops when its steps and exchanges are done; the barrier follows, the MAX over ranks is the job's time)
        dist.barrier()
        torch.cuda.synchronize()
        rec3 = ring3.finish()
        t = torch.tensor([c3_mine], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c3_elapsed = float(t.item())
        c3_kernel = [a.elapsed_time(b) for a, b in c3_ev]
        mine3 = torch.tensor([sum(c3_kernel) / len(c3_kernel), min(c3_kernel), float(len(sh))], dtype=torch.float64, device=ctl)
        rows3 = [torch.zeros(3, dtype=torch.float64, device=ctl) for _ in range(world)]
        dist.all_gather(rows3, mine3)
        # the same self-check for the sharded 10^9-point cloud: global AABB == affine(union of the shards' source bounds), on every rank
        from pasture_amd.distributed import F64_MAX as _F64_MAX, verify_global_bounds as _verify
        s_rec = torch.tensor([_F64_MAX] * 3 + [-_F64_MAX] * 3, dtype=torch.float64, device="cuda")
        if len(sh):
            pa.calculate_bounds_async(s_src, s_rec.data_ptr())
        torch.cuda.synchronize()
        try:
            c3_check = _verify(s_rec, rec3, SCALE, OFFSET)
        except AssertionError as e:
            c3_check = {"verified": False, "ranks": world, "error": str(e)[:2000]}
        configs3 = {"global_points": g3, "n_gpus": n_ranks_seen, "scaling": "strong", "steps": c3_steps,
                    "ms_per_step": round(c3_elapsed / c3_steps * 1e3, 4), "value": round(g3 * c3_steps / c3_elapsed / 1e6, 2), "unit": "Mpoints/s",
                    "aggregate_GBps": round(bytes_per_point * g3 * c3_steps / c3_elapsed / 1e9, 1),
                    "frac_of_aggregate_peak": round(bytes_per_point * g3 * c3_steps / c3_elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
                    "points_rank0": len(sh), "bounds": bounds_from_record(rec3.cpu()), "self_check": c3_check,
                    "per_rank": {"points": [int(r[2]) for r in rows3], "kernel_ms_avg": [round(float(r[0]), 4) for r in rows3], "kernel_ms_min": [round(float(r[1]), 4) for r in rows3],
                                 "kernel_frac_of_peak": [round(bytes_per_point * float(r[2]) / (float(r[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if float(r[0]) > 0 else None for r in rows3]},
                    "exchange_and_launch_ms_per_step": round(c3_elapsed / c3_steps * 1e3 - max(float(r[0]) for r in rows3), 4),
                    "note": "BASELINE.json configs[3]: ONE cloud sharded by index range, rank r owns [r*ceil(G/N), min(G,(r+1)*ceil(G/N))); "
                            "one AABB all-reduce per step; wall time from the opening barrier to each rank's completion, max over ranks"}
        del s_src, s_dst

    # BASELINE.json configs[2] and configs[4] made driver-visible (N = 1, default workload): measured after the timed region, never folded into `value`
    extra_legs, extra_checks = {}, {}
    if world == 1 and not distributed and args.workload == "convert_affine_bounds" and not args.global_points and not args.no_extra_legs:
        src = dst = None
        try:
            rep, smp = leg_configs2(pa, las, cv, torch, stream, n, SEED)
            extra_legs["configs2_las0_to_columns"], extra_checks["configs2"] = rep, smp
        except Exception as e:  # noqa: BLE001  (a failed leg is reported, it does not take the headline with it)
            extra_legs["configs2_las0_to_columns"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        try:
            rep, smp = leg_configs4(pa, torch, stream, n, SEED, pmc=not args.no_traffic_run)
            extra_legs["configs4_knn16"], extra_checks["configs4"] = rep, smp
            from pasture_amd.algorithms import release_scratch
            release_scratch()
        except Exception as e:  # noqa: BLE001
            extra_legs["configs4_knn16"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        try:
            rep, smp = leg_configs4(pa, torch, stream, n, SEED, n_queries=16, sheet=True)
            extra_legs["configs4_knn16_sheet"], extra_checks["configs4_sheet"] = rep, smp
            from pasture_amd.algorithms import release_scratch
            release_scratch()
        except Exception as e:  # noqa: BLE001
            extra_legs["configs4_knn16_sheet"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        try:
            extra_legs["dropin_convert_then_bounds"] = leg_dropin(pa, cv, torch, stream, n, SEED, result)
        except Exception as e:  # noqa: BLE001
            extra_legs["dropin_convert_then_bounds"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        try:
            rep, smp = leg_chunked(pa, las, cv, torch, stream, SEED)
            extra_legs["chunked_rawlas_1MiB"], extra_checks["chunked"] = rep, smp
        except Exception as e:  # noqa: BLE001
            extra_legs["chunked_rawlas_1MiB"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        src = dst = None

    # north_star size made driver-visible: the same fused convert + AABB over 10^9 points (24 GB in, 24 GB out) on ONE GPU, measured in this
    # run after the timed region (N = 1, default workload only); reported beside the headline, never folded into `value`
    north_star = None
    if (world == 1 and not args.no_north_star and args.workload == "convert_affine_bounds" and not args.global_points
            and args.north_star_points > n):
        nn = args.north_star_points
        free_b, _total_b = torch.cuda.mem_get_info()
        if free_b > 2 * 24 * nn + (8 << 30):
            src = dst = None
            big_src = pa.HashMapBuffer.new_from_layout(layout)
            big_src.resize(nn)
            big_src.synth_fill(SEED, 0)
            big_dst = pa.HashMapBuffer.new_from_layout(layout)
            big_dst.resize(nn)
            big_rec = torch.empty(6, dtype=torch.float64, device="cuda")
            ns_steps = 10
            for _ in range(2):
                conv.convert_into_with_bounds_async(big_src, big_dst, big_rec.data_ptr())
            torch.cuda.synchronize()
            ns_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ns_steps)]
            for e0, e1 in ns_ev:
                e0.record(stream)
                conv.convert_into_with_bounds_async(big_src, big_dst, big_rec.data_ptr())
                e1.record(stream)
            torch.cuda.synchronize()
            ns_all = [e0.elapsed_time(e1) for e0, e1 in ns_ev]
            ns_ms, ns_min = sum(ns_all) / ns_steps, min(ns_all)
            ns_gbs = bytes_per_point * nn / (ns_ms * 1e-3) / 1e9
            ns_traffic = None
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("convert_affine_bounds_1e9")
                if t and t.get("points") == nn:
                    ns_traffic = {"bytes_per_launch": t.get("bytes_per_launch"), "algorithmic_bytes_per_launch": bytes_per_point * nn,
                                  "source": f"profiles/hbm_traffic.json (round {t.get('round')}, separate rocprofv3 --pmc passes of the 10^9-point run); NOT measured in this run"}
            except Exception:
                pass
            north_star = {"points": nn, "steps": ns_steps, "ms_per_step": round(ns_ms, 4), "ms_per_step_min": round(ns_min, 4),
                          "value": round(nn / (ns_ms * 1e-3) / 1e6, 2),
                          "unit": "Mpoints/s", "achieved_GBps": round(ns_gbs, 1), "frac": round(ns_gbs / HBM_PEAK_GBS, 4),
                          "frac_best_step": round(bytes_per_point * nn / (ns_min * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "traffic": ns_traffic, "bounds": bounds_from_record(big_rec.cpu()),
                          "note": "north_star size (10^9 points, 1 GPU): same kernel, same fused step, HIP events around each of 10 steps (avg and min)"}
            del big_src, big_dst

    if rank == 0:
        total_points = global_points * args.steps
        value = total_points / elapsed / 1e6
        achieved = bytes_per_point * n / (kernel_ms_avg * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath)).get(args.workload)
                if t and t.get("points") == n:
                    traffic = t.get("bytes_per_launch")
                    traffic_src = (f"profiles/hbm_traffic.json (round {t.get('round')}; separate rocprofv3 --pmc passes, fetch factor "
                                   f"{t.get('fetch_factor', 2)}: {t.get('calibrated_on', 'wide coalesced reads')}); NOT measured in this run")
            except Exception:
                pass
        if world == 1 and not distributed and args.workload == "convert_affine_bounds" and not args.no_traffic_run and not args.no_extra_legs and not args.global_points:  # (the default driver run only: the tools pass --no-extra-legs, some of them run under rocprofv3 themselves)
            m = measure_traffic_in_run(args.workload, n, "vec3f64_stream2_kernel")
            if m is not None:
                traffic, traffic_src = m
        line = {
            "metric": "Mpoints/sec + achieved HBM GB/s (% of peak), 10^8-pt POSITION_3D convert+AABB",
            "value": round(value, 2), "unit": "Mpoints/s", "n_gpus": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "points_per_gpu": n, "global_points": global_points,
                       "layout": "columnar Vec3f64" if args.workload in ("convert_affine_bounds", "bounds", "narrow_f64_f32", "normals_knn16", "normals_knn16_sheet") else "LAS format 0",
                       "parallelism": (f"index-range shard x{world} of one {global_points}-point cloud (configs[3]), one all-reduce of the 6-f64 AABB" if args.global_points
                                       else f"index-range shard x{world}, one all-reduce of the 6-f64 AABB") if distributed else "1 GPU",
                       "seed": SEED, "bounds": result, "plan": plan_kinds, "plan_requested": args.plan,
                       **({"family_measured": family_rep} if family_rep is not None else {})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_point": bytes_per_point, "kernel_ms_avg": round(kernel_ms_avg, 4),
                         "kernel_ms_min": round(min(kernel_ms), 4), "kernel_ms_samples": len(kernel_ms),
                         "note": "HIP events around one step's launches on the launch stream (conversion kernel + the AABB fold kernels where fused), "
                                 f"on every {every}. step of the timed region"},
        }
        if args.workload.startswith("normals_knn"):
            # The kNN call is bound by its vector-instruction count, not by HBM: beside the HBM lower bound above, the
            # bound that binds -- vector (wave64) instructions of the dominant kernel per launch, from the committed SQ_INSTS_VALU pass,
            # over the chip's issue rate (256 CUs x 4 SIMDs, one wave instruction per 4 cycles at 2.4 GHz).
            try:
                v = json.load(open(os.path.join(ROOT, "profiles", "knn_valu.json"))).get(args.workload)
            except Exception:
                v = None
            if v and v.get("points") == n:
                issue_rate = 256 * 4 * 2.4e9 / 4.0
                bound_ms = v["valu_wave_instructions_per_launch"] / issue_rate * 1e3
                line["roofline"]["bound_valu"] = {
                    "kernel": v["kernel"], "valu_wave_instructions_per_launch": v["valu_wave_instructions_per_launch"],
                    "valu_wave_instructions_per_point": round(v["valu_wave_instructions_per_launch"] / n, 2),
                    "issue_rate_wave_instructions_per_s": issue_rate, "bound_ms": round(bound_ms, 3),
                    "kernel_ms": v.get("kernel_ms"), "frac_of_kernel": round(bound_ms / v["kernel_ms"], 4) if v.get("kernel_ms") else None,
                    "frac_of_step": round(bound_ms / (kernel_ms_avg), 4),
                    "source": f"profiles/knn_valu.json (round {v.get('round')}: rocprofv3 --pmc SQ_INSTS_VALU and --kernel-trace passes of this workload); NOT measured in this run",
                    "note": "frac_of_kernel = time the dominant kernel's vector instructions need at full issue rate / its measured duration; "
                            "frac_of_step = the same against the whole call (index build, sort, fallback searches included)"}
        if distributed:
            line["config"]["collective"] = transport.name if transport is not None else "torch.distributed.all_reduce (two 3 x f64 collectives: MIN of the minima, MAX of the maxima)"
            if collective_note:
                line["config"]["collective_note"] = collective_note
            line["config"]["timing"] = ("barrier + synchronize, K steps, each rank stops its clock when its steps and its exchanges are complete (exchange stream dry, "
                                        "device synchronized), then the closing barrier; value uses the MAX over ranks")
            if args.rehearse_on_one_gpu:
                line["config"]["rehearsal"] = "every rank on GPU 0 over gloo: the N > 1 logic with real kernels; timings are meaningless"
            if per_rank is not None:
                line["per_rank"] = per_rank
            if self_check is not None:
                line["self_check"] = self_check
        if configs3 is not None:
            line["configs3_1e9"] = configs3
        if north_star is not None:
            line["north_star_1e9"] = north_star
        for name, rep in extra_legs.items():
            line[name] = rep
        if world == 1 and not args.no_cpu_baseline and args.workload == "convert_affine_bounds":
            cb = cpu_baseline(args.workload, args.cpu_sample_points, extra_checks)
            if cb is not None:
                line["cpu_baseline"] = cb
                # the headline's own check: the AABB of the 10^8 converted points on the GPU == the oracle's over the same synthetic points, digit for digit
                if args.cpu_sample_points == n and not args.global_points:
                    gpu_b = [float(x) for x in result[0]] + [float(x) for x in result[1]] if result else None
                    line["verified"] = bool(gpu_b is not None and gpu_b == [float(x) for x in cb["bounds"]])
                    line["verification"] = "config.bounds (GPU, fused convert + AABB of the timed steps) == cpu_baseline.bounds (oracle, same 10^8 synthetic points), all six doubles"
                sc = cb.get("spot_checks") if isinstance(cb.get("spot_checks"), dict) else {}
                if "configs2" in sc and "configs2_las0_to_columns" in line:
                    line["configs2_las0_to_columns"]["verified"] = bool(sc["configs2"].get("verified"))
                if "configs4_sheet" in sc and "configs4_knn16_sheet" in line:
                    ck = line["configs4_knn16_sheet"].get("checks", {})
                    line["configs4_knn16_sheet"]["verified"] = bool(sc["configs4_sheet"].get("verified")) and bool(ck.get("columns_equal_f64_results_narrowed")) \
                        and ck.get("neighbour_lists_exact") == ck.get("neighbour_lists_checked")
                if "chunked" in sc and "chunked_rawlas_1MiB" in line:
                    leg = line["chunked_rawlas_1MiB"]
                    leg["verified"] = bool(sc["chunked"].get("verified"))
                    orc_us = sc["chunked"].get("oracle_us_per_call_by_points", {})
                    leg["oracle_us_per_call_by_points"] = orc_us
                    gpu_us = leg.get("us_per_call_by_points", {})
                    faster = [int(k) for k in gpu_us if k in orc_us and gpu_us[k] < orc_us[k]]
                    leg["gpu_call_overtakes_the_oracle_from_points"] = min(faster) if faster else None
                if "configs4" in sc and "configs4_knn16" in line:
                    ck = line["configs4_knn16"].get("checks", {})
                    line["configs4_knn16"]["verified"] = bool(sc["configs4"].get("verified")) and bool(ck.get("columns_equal_f64_results_narrowed")) \
                        and ck.get("neighbour_lists_exact") == ck.get("neighbour_lists_checked")
        print(json.dumps(line), flush=True)
    failed_check = (self_check is not None and not self_check["verified"]) or (configs3 is not None and not configs3["self_check"]["verified"])
    if failed_check and rank == 0:
        sys.stderr.write("bench.py: the in-run self-check of the sharded AABB FAILED (see self_check in the line above): exit 3\n")
    if distributed:
        dist.barrier()
        if transport is not None:
            transport.close()
        dist.destroy_process_group()
    if failed_check:
        sys.exit(3)


if __name__ == "__main__":
    main()
