"""The C-ABI library loads without a GPU and exports every symbol include/pasture_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pasture_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pst_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pasture_amd._capi import LIB_PATH, PRODUCT_SYMBOLS
    assert os.path.exists(LIB_PATH), "libpasture_amd.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pasture_amd.h but not exported"
    # and the Python binding covers the whole header
    assert sorted("pst_" + s for s in PRODUCT_SYMBOLS) == syms


def test_no_cpu_fallback_without_device():
    """Compute entry points must fail loudly when no GPU is usable (this container) instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pasture_amd as pa
    from pasture_amd.layout import attributes as A
    layout = pa.PointLayout.from_attributes([A.POSITION_3D])
    buf = pa.HashMapBuffer.new_from_layout(layout)
    with pytest.raises(pa.PastureError) as e:
        buf.resize(16)
    assert e.value.code == 21 and "no CPU fallback" in str(e.value)
    assert pa.calculate_bounds(buf) is None  # empty buffer: answered on the host like bounds.rs:12-14


def test_product_never_references_oracle():
    """The product path must not import / link / execute anything under oracle/."""
    pkg = os.path.join(ROOT, "pasture_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "pasture_oracle" not in src and "oracle_capi" not in src, os.path.join(dirpath, f)
