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
    """The header-only C++ mirror of the reference's plugin surface (include/hydroc_amd/*.h, Chrono-free part) and the
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
    assert os.path.exists(hb.TUNING_CO) and os.path.exists(hb.TUNING_LIB), "the tuning build is missing: run __graft_entry__.build()"
    for name in (b"finalize_kernelILi4ELb0E", b"finalize_kernelILi4ELb1E", b"scatter_kernelE", b"reduce_block_kernelE", b"conv_block_kernelILi6ELi4ELi2ELi1E",
                 b"conv_block_kernelILi6ELi3ELi1ELi2E", b"conv_step_kernelILi4ELi2E", b"added_mass_mv_tagged_kernelE",
                 b"conv_block_kernelILi4ELi4ELi1ELi2E", b"near_split_kernelE", b"wide_step_kernelE"):
        assert name in blob, name
    assert os.path.getmtime(hb.KERNEL_CO) >= os.path.getmtime(os.path.join(hb.CSRC, "hc_kernels.hip"))


OPERATIONAL_SWITCHES = {"HC_DIRECT", "HC_ARM", "HC_PASS_AHEAD", "HC_PASS_AHEAD_GAP_US", "HC_PASS_CONCURRENT", "HC_DEVICE_SHARED", "HC_MULTI_THREADS",
                        "HC_MULTI_SPIN_US", "HC_STEP_TIMEOUT_S", "HC_QUEUE_DEV_MEM", "HC_MULTI_PIN", "HC_HDP_FLUSH"}


def _hc_names(path):
    blob = open(path, "rb").read()
    return {m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])(HC_[A-Z][A-Z0-9_]+)\x00", blob)}


def test_release_library_reads_operational_switches_only():
    """The shipped library knows twelve environment variables, all operational and all listed in INTEGRATION.md; every sweep / A-B /
    fault-injection switch lives in the tuning build (-DHC_TUNING) only.  Checked on the strings of the built binaries: a name
    that is not in the release library cannot be read by it."""
    from hydrochrono_amd import build as hb
    names = _hc_names(hb.MAIN_LIB)
    env_like = {n for n in names if not n.startswith(("HC_ERR", "HC_OK", "HC_API", "HC_HIP", "HC_HOST", "HC_FANOUT", "HC_TUNE"))}
    assert env_like == OPERATIONAL_SWITCHES, sorted(env_like ^ OPERATIONAL_SWITCHES)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in OPERATIONAL_SWITCHES:
        assert n in doc, f"{n} is read by the release library but not documented in INTEGRATION.md"
    tuning = _hc_names(hb.TUNING_LIB)
    for n in ("HC_SUB_BLOCK", "HC_MINI_NARROW", "HC_WIDE_FUSED", "HC_SLOT_STATE", "HC_STEP_CANARY", "HC_FAULT_STALE_STATE_AT", "HC_BLOCK_MT", "HC_BLOCK_V32"):
        assert n in tuning and n not in names, n


def _kernel_notes(path):
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        import pytest
        pytest.skip("llvm-readelf not found")
    txt = subprocess.run([readelf, "--notes", path], capture_output=True, text=True, check=True).stdout
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        out[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    return out


def test_release_code_object_has_no_spills_no_scratch_and_no_rejected_variants():
    """Every kernel of the shipped code object keeps its registers (no spilled VGPR, no scratch: a kernel with scratch cannot go to
    the direct queue at all), and the variants that were measured and not taken -- 12 row tiles per pass workgroup (148 spilled
    VGPRs at depth 32), the depth-64 pass, the HC_BLOCK_V32 / unroll sweeps -- are not in it (they live in the tuning build)."""
    from hydrochrono_amd import build as hb
    rel = _kernel_notes(hb.KERNEL_CO)
    assert len(rel) >= 20
    for name, (scratch, vgpr, spills) in rel.items():
        assert scratch == 0 and spills == 0, (name, scratch, spills)
    assert not any("conv_block_kernelILi12E" in n for n in rel), "the 12-tile pass variant ships"
    assert not any(re.search(r"conv_block_kernelILi\dELi\dELi4E", n) for n in rel), "a depth-64 pass variant ships"
    assert not any(re.search(r"conv_step_kernelILi\dELi[13]E", n) for n in rel), "an unroll-sweep variant of the plain kernel ships"
    assert not any("finalize_pre_kernel" in n for n in rel), "the kernel-argument-preload variant of the step kernel ships"
    assert sum("step_hot_kernel" in n for n in rel) == 2, "the step kernel of the common block step (one and two own IRF samples) is missing"
    tun = _kernel_notes(hb.TUNING_CO)
    assert len(tun) > len(rel) and any(re.search(r"conv_block_kernelILi\dELi\dELi4E", n) for n in tun)


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
