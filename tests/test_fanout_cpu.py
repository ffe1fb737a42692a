"""CPU test of the worker hand-off of hc_step_multi / hc_added_mass_mv_multi (hydrochrono_amd/csrc/hc_fanout.hpp, host-only C++):
plain build, and under ThreadSanitizer (GPU-side race detection is not available on the pool; the hand-off is the one piece of
the multi-GPU step that is multi-threaded, and it has no HIP dependency)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "fanout_test.cpp")


def test_fanout_hand_off(tmp_path):
    exe = str(tmp_path / "fanout_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", SRC, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith(" 0 failures"), r.stdout[-3000:]


def test_fanout_hand_off_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "fanout_test_tsan")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", SRC, "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "tsan" in (b.stderr or "").lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    text = r.stdout + r.stderr
    if "FATAL: ThreadSanitizer" in text and "unexpected memory mapping" in text:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert "WARNING: ThreadSanitizer" not in text, text[-4000:]
    assert r.returncode == 0 and " 0 failures" in r.stdout, text[-3000:]
