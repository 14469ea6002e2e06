#!/usr/bin/env python3
"""A process that ends while the compiler thread is inside hipRTC (round 6: one such process dumped core at exit).  Queues the background compilation
of a fresh plan (a random packed layout, so that neither the in-memory nor the on-disk cache has it) and leaves at once; run N times by
tools/r06_calls/r06_gpu5.sh, which counts the exit statuses."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PST_JIT_CACHE"] = "0"
import numpy as np  # noqa: E402

import pasture_amd as pa  # noqa: E402
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
types = [T.U8, T.U16, T.I32, T.F32, T.F64, T.Vec3f32, T.Vec3u16, T.I64]
attrs = [PointAttributeDefinition(f"A{k}", types[int(rng.integers(len(types)))]) for k in range(int(rng.integers(3, 9)))]
layout = PointLayout.from_attributes_packed(attrs, 1)
n = 1 << 21
src = pa.VectorBuffer.new_from_layout(layout)
src.resize(n)
dst = pa.HashMapBuffer.new_from_layout(layout)
dst.resize(n)
conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
conv.convert_into_range_async(src, range(0, n), dst, range(0, n))  # interpreted now, the specialised kernel queued for the compiler thread
print("queued", seed, flush=True)
