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


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(ROOT, "pasture_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", lib_dir,
           "-lpasture_amd", f"-Wl,-rpath,{lib_dir}", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_plain_c_program_links_against_the_boundary_and_fails_loudly_without_gpu(tmp_path):
    """The boundary is usable from plain C (no C++ / torch types): examples/c_abi_demo.c compiles with -Wall -Werror against
    include/pasture_amd.h and links to libpasture_amd.so.  Without a GPU its first compute call must fail with
    PST_ERR_NO_DEVICE (exit code 77 of the demo) — never fall back to a CPU path."""
    import subprocess
    import torch
    exe = _build_c_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked run of the demo")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 77, r.stderr
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_program_runs_on_the_gpu(tmp_path):
    import subprocess
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("OK: 100000 points")


def test_rust_repr_c_layouts_match_the_header(tmp_path):
    """No rustc in the image: the layouts rust/pasture-amd-sys declares (#[repr(C)]) are pinned mechanically instead.  tools/gen_rust_sys.py
    writes the Rust structs AND tests/abi/rust_layout_asserts.c (_Static_assert of sizeof / alignof / offsetof per field) from one model;
    the C file must compile against include/pasture_amd.h, the committed generated files must be what the generator produces now, and every
    struct the header defines must be in the model."""
    import shutil
    import subprocess
    gen = os.path.join(ROOT, "tools", "gen_rust_sys.py")
    rs = os.path.join(ROOT, "rust", "pasture-amd-sys", "src", "lib.rs")
    cf = os.path.join(ROOT, "tests", "abi", "rust_layout_asserts.c")
    before = (open(rs).read(), open(cf).read())
    try:
        subprocess.check_call(["python3", gen], stdout=subprocess.DEVNULL)
        after = (open(rs).read(), open(cf).read())
    finally:
        open(rs, "w").write(before[0])
        open(cf, "w").write(before[1])
    assert after == before, "rust/pasture-amd-sys/src/lib.rs or tests/abi/rust_layout_asserts.c is stale: run tools/gen_rust_sys.py"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-c", "-I", os.path.join(ROOT, "include"), cf, "-o", str(tmp_path / "asserts.o")])
    # a wrong number must actually fail (the check is not vacuous)
    bad = tmp_path / "bad.c"
    bad.write_text(before[1].replace("sizeof(pst_datatype) == 40", "sizeof(pst_datatype) == 48"))
    assert "sizeof(pst_datatype) == 48" in bad.read_text()
    r = subprocess.run(["gcc", "-std=c11", "-c", "-I", os.path.join(ROOT, "include"), str(bad), "-o", str(tmp_path / "bad.o")], capture_output=True, text=True)
    assert r.returncode != 0 and "sizeof(pst_datatype)" in r.stderr
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pasture_amd.h")).read(), flags=re.S)
    defined = set(re.findall(r"typedef struct(?:\s+\w+)?\s*\{[^}]*\}\s*(pst_\w+)\s*;", hdr))
    assert defined and all(f"sizeof({s})" in before[1] for s in defined), defined
    assert shutil.which("gcc")
