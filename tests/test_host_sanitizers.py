"""AddressSanitizer + UBSan over the host-only sources of the product (the hydro.yaml reader and the init-time host math).
GPU sanitizers are not available on the pool, so the device code is covered by the parity tests only."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hydrochrono_amd", "csrc")
FLAGS = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def build(out, sources, extra):
    r = subprocess.run(FLAGS + extra + sources + ["-o", out], capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower():
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr
    return out


def run_clean(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-2000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-2000:]
    return text


def test_host_math_under_sanitizers(tmp_path):
    exe = build(str(tmp_path / "host_math"), [os.path.join(ROOT, "tests", "cpp", "host_math_driver.cpp"), os.path.join(CSRC, "hc_host_math.cpp")],
                ["-I", CSRC])
    assert "host math ok" in run_clean([exe])


def test_yaml_reader_under_sanitizers(tmp_path):
    from test_hydro_yaml import CASES
    files = []
    for name, text in CASES.items():
        p = tmp_path / f"{name}.hydro.yaml"
        with open(p, "w", newline="") as fh:
            fh.write(text)
        files.append(str(p))
    for i, f in enumerate(sorted(glob.glob("/root/reference/**/*.hydro.yaml", recursive=True))):
        dst = tmp_path / f"ref_{i}.hydro.yaml"
        shutil.copy(f, dst)
        files.append(str(dst))
    # hc_create_from_hydro_yaml (the one function of hc_yaml.cpp that calls into the GPU library) is not exercised here
    exe = build(str(tmp_path / "yaml"), [os.path.join(ROOT, "tests", "cpp", "yaml_driver.cpp"), os.path.join(CSRC, "hc_yaml.cpp"),
                                         os.path.join(CSRC, "hc_host_math.cpp")],
                ["-I", os.path.join(ROOT, "include"), "-Wl,--unresolved-symbols=ignore-all"])
    out = run_clean([exe] + files)
    assert out.count(": ok bodies=") + out.count(": error ") == len(files)


@pytest.mark.parametrize("name", ["plan_test", "history_test"])
def test_planner_and_history_index_under_sanitizers(tmp_path, name):
    """The host-only headers of the step path (look-ahead planner incl. the two-level form, history bookkeeping incl. rewinds)."""
    exe = build(str(tmp_path / name), [os.path.join(ROOT, "tests", "cpp", name + ".cpp")], [])
    assert " 0 failures" in run_clean([exe]) or "0 failures" in run_clean([exe])
