"""The C++ host-side mirror of the reference's plugin surface (include/hydroc_amd/hydro_forces.h) driven by a
Chrono-free C++ program (examples/sphere_mock_chrono.cpp): BEMIO-HDF5 ingest -> TestHydro / wave classes ->
CoordinateFuncForBody callbacks -> heave trajectory, compared with the reference's golden files."""
import os
import subprocess

import numpy as np
import pytest

from cases import GOLDEN_DIR, goldens

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from hydrochrono_amd import build as hb
    hb.build()
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    out = str(tmp_path_factory.mktemp("cpp") / "sphere_mock_chrono")
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "examples", "sphere_mock_chrono.cpp"), "-o", out,
                    "-L", libdir, "-lhydrochrono_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return out


def run(exe, mode, nsteps):
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "sphere.h5"), mode, str(nsteps)], check=True, capture_output=True, text=True)
    a = np.array([[float(x) for x in line.split()] for line in r.stdout.strip().splitlines()])
    assert a.shape == (nsteps, 2)
    return a[:, 1]


def test_cpp_decay(exe):
    ref = goldens()["decay_z_um"] * 1e-6
    z = run(exe, "decay", len(ref))
    assert np.max(np.abs(z - ref)) <= 1.01e-6  # both sides print 6 decimals


def test_cpp_regular(exe):
    ref = goldens()["reg_waves_1_z_um"][:3000] * 1e-6
    assert np.max(np.abs(run(exe, "regular", 3000) - ref)) <= 1.01e-6


def test_cpp_irregular(exe):
    ref = goldens()["irreg_waves_z_um"][:6000] * 1e-6
    d = np.abs(run(exe, "irregular", 6000) - ref)
    assert d.max() <= 1e-4 and d[5000:].max() <= 6e-6


def test_cpp_array_example_on_several_shards(tmp_path):
    """examples/array_multi_gpu.cpp: a coupled four-body array through the Chrono-free C++ mirror with a DEVICE LIST -- one, two and
    four row shards (all on the one GPU here); the printed trajectories do not depend on the shard count."""
    from hydrochrono_amd import build as hb
    hb.build()
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    exe = str(tmp_path / "array_multi_gpu")
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "array_multi_gpu.cpp"), "-o", exe,
                    "-L", libdir, "-lhydrochrono_amd", f"-Wl,-rpath,{libdir}"], check=True)
    outs = []
    for devs in ("0", "0,0", "0,0,0,0"):
        r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "four_body.h5"), "4", "400", devs], check=True, capture_output=True, text=True)
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2]
    a = np.array([[float(x) for x in line.split()] for line in outs[0].strip().splitlines()])
    assert a.shape == (400, 5) and np.all(np.isfinite(a)) and np.max(np.abs(a[:, 1:])) < 5.0 and np.ptp(a[:, 1]) > 0.05
