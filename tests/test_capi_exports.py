"""The C-ABI library must load without a GPU and export every symbol the public headers declare."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for hdr in ("hydrochrono_amd.h", "hydrochrono_amd_host.h", "hydrochrono_amd_yaml.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(hc_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    from hydrochrono_amd import capi
    lib = capi.load()
    decl = declared_symbols()
    assert len(decl) >= 45
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert decl == set(capi.SIGNATURES), (decl ^ set(capi.SIGNATURES))
    assert b"gfx950" in lib.hc_version()


def test_no_cpu_fallback_without_device():
    """On a box without a GPU, creating a context must fail loudly (HC_ERR_DEVICE), not fall back."""
    from hydrochrono_amd import capi
    lib = capi.load()
    if lib.hc_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    ctx = ctypes.c_void_p()
    rc = lib.hc_create(1, 0, ctypes.byref(ctx))
    assert rc == capi.HC_ERR_DEVICE and not ctx.value
    assert b"no CPU fallback" in lib.hc_last_error(None)


def test_product_never_touches_the_oracle():
    """Nothing under hydrochrono_amd/ may import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "hydrochrono_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "hc_oracle" not in text and "oracle/" not in text and "import oracle" not in text, f


def test_cpp_mirror_header_and_example_compile_and_link(tmp_path):
    """The header-only C++ mirror of the reference's plugin surface (hydro_forces_amd.hpp, Chrono-free part) and the
    example driver build with plain g++ against the C-ABI library.  Running them needs a GPU (tests/test_gpu_cpp_mirror.py)."""
    import subprocess
    from hydrochrono_amd import build as hb
    hb.build()
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    out = str(tmp_path / "sphere_mock_chrono")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "examples", "sphere_mock_chrono.cpp"), "-o", out,
                    "-L", libdir, "-lhydrochrono_amd", f"-Wl,-rpath,{libdir}"], check=True)
    assert os.path.exists(out)


def test_kernel_code_object_is_built_next_to_the_library():
    """build() also leaves the kernels as a stand-alone gfx950 code object: hc_step loads it through the HSA loader and writes
    its AQL packets itself (hydrochrono_amd/csrc/hc_direct.hpp).  Without the file the library falls back to HIP launches, so
    its absence would go unnoticed on the GPU -- it is checked here."""
    from hydrochrono_amd import build as hb
    assert os.path.exists(hb.KERNEL_CO), "run __graft_entry__.build()"
    blob = open(hb.KERNEL_CO, "rb").read()
    assert blob[:4] == b"\x7fELF" and len(blob) > 100_000
    for name in (b"finalize_kernelILi4ELb0E", b"finalize_kernelILi4ELb1E", b"scatter_kernelE", b"reduce_block_kernelE", b"conv_block_kernelILi6ELi4ELi2ELi1E",
                 b"conv_block_kernelILi6ELi3ELi1ELi2E", b"conv_step_kernelILi4ELi2E", b"added_mass_mv_tagged_kernelE",
                 b"conv_block_kernelILi4ELi4ELi1ELi2E", b"near_split_kernelE", b"wide_step_kernelE"):
        assert name in blob, name
    assert os.path.getmtime(hb.KERNEL_CO) >= os.path.getmtime(os.path.join(hb.CSRC, "hc_kernels.hip"))


def test_wait_result_buffer_is_host_only():
    """hc_wait_result_buffer (the reader side of the shared-memory host gather) needs no GPU: granules {value, sequence} of the
    half that belongs to the sequence number's parity are copied out; a sequence that never arrives ends with HC_ERR_DEVICE."""
    import ctypes as C

    import numpy as np

    from hydrochrono_amd import capi
    lib = capi.load()
    rows = 5
    buf = np.zeros(2 * 2 * rows, dtype=np.uint64)  # two halves of [rows][2]
    vals = np.array([1.5, -2.25, 3.0e10, 0.0, -7.0])
    for seq in (7, 8):
        half = buf[(seq & 1) * 2 * rows:(seq & 1) * 2 * rows + 2 * rows]
        half[0::2] = (vals * seq).view(np.uint64)
        half[1::2] = seq
    out = np.empty(rows)
    dp = out.ctypes.data_as(capi.c_double_p)
    for seq in (7, 8):
        assert lib.hc_wait_result_buffer(buf.ctypes.data, rows, seq, dp, 1.0) == capi.HC_OK
        assert np.array_equal(out, vals * seq)
    buf[1] = 6  # one granule of the even half still carries an older sequence number
    assert lib.hc_wait_result_buffer(buf.ctypes.data, rows, 8, dp, 0.05) == capi.HC_ERR_DEVICE
    assert lib.hc_wait_result_buffer(None, rows, 8, dp, 0.05) == capi.HC_ERR_INVALID
    assert lib.hc_step_sequence(None, C.byref(C.c_ulonglong())) == capi.HC_ERR_INVALID
