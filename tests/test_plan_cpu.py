"""CPU test of the look-ahead planner (hydrochrono_amd/csrc/hc_plan.hpp, host-only C++): the pass / scatter-target / own-entry
partition of the radiation sum reproduces the direct evaluation (bracket search + interpolation weights of
src/hydro_forces.cpp:343-381,589-647) for step sizes equal to, below and above the IRF spacing, both block lengths, and
histories shorter than the IRF window (deferred-sample rule).  Built with plain g++ -- no GPU, no HIP."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_partition_reproduces_the_direct_sum(tmp_path):
    exe = str(tmp_path / "plan_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "plan_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout.splitlines()[-1]
