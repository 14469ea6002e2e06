#!/usr/bin/env python3
"""Probe (round 6): the disk cache of the plan-specialised kernels when its files are damaged, truncated, replaced by another plan's code object, or the
directory cannot be written.  Every child process converts a random packed layout records -> columns with PST_JIT=sync and compares with numpy; the family
the library reports must be the plan-specialised one every time."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, ROOT_DIR); sys.path.insert(0, ROOT_DIR + "/tests")
from harness import random_records
from pasture_amd import conversion as cv
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter
from pasture_amd.layout import PointAttributeDataType as T, PointLayout, attributes as A
seed = int(sys.argv[1])
layout = PointLayout.from_attributes_packed([A.GPS_TIME, A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION.with_custom_datatype(T.U32), A.COLOR_RGB][: 3 + seed % 3], 1)
n = 300_007
rec = random_records(layout, n, seed)
src = VectorBuffer.from_numpy(rec, layout)
out = BufferLayoutConverter.for_layouts(layout, layout).convert(src, HashMapBuffer)
kinds = cv.last_plan_kinds()
ok = all(out.view_attribute(a.attribute_definition()).tobytes() == np.ascontiguousarray(rec[a.name()]).tobytes() for a in layout.attributes())
print("RESULT", "ok" if ok else "WRONG", kinds, cv.jit_stats())
'''.replace("ROOT_DIR", repr(ROOT))


def child(seed, cache_dir):
    env = dict(os.environ, PST_JIT="sync", PST_JIT_CACHE_DIR=cache_dir, PST_QUIET="1")
    r = subprocess.run([sys.executable, "-c", CHILD, str(seed)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    return r.returncode, (line[-1] if line else r.stderr[-400:])


def main():
    d = "/tmp/pst_cache_probe"
    subprocess.run(["rm", "-rf", d])
    bad = 0
    for what in ("cold", "warm", "truncated", "garbage", "swapped", "unwritable"):
        if what == "truncated":
            for f in glob.glob(d + "/*"):
                data = open(f, "rb").read()
                open(f, "wb").write(data[: len(data) // 2])
        elif what == "garbage":
            for f in glob.glob(d + "/*"):
                open(f, "wb").write(os.urandom(os.path.getsize(f) or 4096))
        elif what == "swapped":  # a VALID code object of another plan under this plan's name
            child(1, d)
            files = sorted(glob.glob(d + "/*"), key=os.path.getmtime)
            if len(files) >= 2:
                a, b = open(files[0], "rb").read(), open(files[-1], "rb").read()
                open(files[0], "wb").write(b)
                open(files[-1], "wb").write(a)
        elif what == "unwritable":
            subprocess.run(["rm", "-rf", d])
            os.makedirs(d)
            os.chmod(d, 0o555)
        rc, line = child(0, d)
        good = rc == 0 and "RESULT ok" in line and "'jit'" in line
        bad += 0 if good else 1
        print(f"{what:11s} rc={rc} {line[:300]}")
    os.chmod(d, 0o755)
    print("all behaved" if not bad else f"{bad} BAD")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
