"""Legs of the randomised differential run (profiles/fuzz_parity.py) in the suite: drawn configurations, stepping patterns and mid-run
reconfigurations through the C ABI against the oracle, every step's total and components.  Round 5's three defects all came out of this
harness, so the driver's suite now runs the draws that found them -- the default draw on both builds, the wide-only draw (6N >= 1024:
column slices, fused wide step, two-level form) and the shard-only draw (2-4 row shards behind hc_step_multi) on the SHIPPED library --
from fixed seeds, plus one leg from a time-derived seed (printed, and named in the failure) so that successive runs of the suite cover
new cases.  The long runs live in profiles/r05/fuzz_parity*.txt and profiles/r06/."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LEGS = [  # id, environment, seconds, first seed (None: derived from the clock), fewest cases expected
    ("tuning_build", {}, 15, 500001, 5),
    ("release_build", {"FUZZ_RELEASE": "1"}, 15, 500001, 5),
    ("release_build_wide_only", {"FUZZ_RELEASE": "1", "FUZZ_WIDE": "1"}, 30, 510001, 3),
    ("release_build_shards_only", {"FUZZ_RELEASE": "1", "FUZZ_SHARDS": "1"}, 30, 520001, 5),
    ("release_build_seed_from_the_clock", {"FUZZ_RELEASE": "1"}, 20, None, 3),
]


@pytest.mark.gpu
@pytest.mark.parametrize("leg", LEGS, ids=[leg[0] for leg in LEGS])
def test_randomised_differential_run_leg(leg):
    name, extra, seconds, seed, fewest = leg
    if seed is None:
        seed = 600000000 + int(time.time()) % 100000000
    env = dict(os.environ)
    for k in ("HYDROCHRONO_AMD_FLAVOR", "FUZZ_RELEASE", "FUZZ_WIDE", "FUZZ_SHARDS", "FUZZ_FLAVOR"):
        env.pop(k, None)
    env.update(extra)
    print(f"fuzz leg {name}: {seconds} s from seed {seed}")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "fuzz_parity.py"), str(seconds), str(seed)], env=env, capture_output=True, text=True, timeout=900)
    tail = f"leg {name}, first seed {seed} (re-run: {' '.join(k + '=' + v for k, v in extra.items())} python profiles/fuzz_parity.py {seconds} {seed})\n" + (r.stdout + r.stderr)[-1800:]
    assert r.returncode == 0 and "fuzz ok:" in r.stdout, tail
    cases = int(r.stdout.split("fuzz ok:")[1].split("cases")[0])
    assert cases >= fewest, tail
    print(r.stdout.strip().splitlines()[-1])
