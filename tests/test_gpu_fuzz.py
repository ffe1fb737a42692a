"""A short leg of the randomised differential run (profiles/fuzz_parity.py) in the suite, so that the harness itself stays alive: drawn
configurations, stepping patterns and mid-run reconfigurations through the C ABI against the oracle, 15 s on each build of the library.
The long runs live in profiles/r05/fuzz_parity*.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("release", [False, True], ids=["tuning_build", "release_build"])
def test_randomised_differential_run_short_leg(release):
    env = dict(os.environ)
    env.pop("HYDROCHRONO_AMD_FLAVOR", None)
    if release:
        env["FUZZ_RELEASE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "fuzz_parity.py"), "15", "500001"], env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "fuzz ok:" in r.stdout, tail
    cases = int(r.stdout.split("fuzz ok:")[1].split("cases")[0])
    assert cases >= 5, tail
    print(r.stdout.strip().splitlines()[-1])
