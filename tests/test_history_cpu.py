"""CPU test of the velocity-history bookkeeping (hydrochrono_amd/csrc/hc_history.hpp, host-only C++): push / prune exactly as
TestHydro does (src/hydro_forces.cpp:327-340,559-574), ring-slot addressing of kept and retired samples through ring growth, and
steps BACK in time (drop the abandoned samples, re-admit retired ones) against the reference's rule applied from scratch.
Built with plain g++ -- no GPU, no HIP."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_history_index_through_rewinds_and_ring_growth(tmp_path):
    exe = str(tmp_path / "history_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "history_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.splitlines()[-1] == "0 failures"
