"""The guarded Project Chrono adapter block of hydrochrono_amd/csrc/hydro_forces_amd.hpp (ChronoBody, ComponentFunc,
ChLoadAddedMass, ChronoHydroSystem -- the counterparts of src/hydro_forces.cpp:63-168,223-234 and
src/chloadaddedmass.cpp:27-70) compiled against minimal stand-in Chrono headers (tests/cpp/chrono_stub/, test
infrastructure only) and driven through ChForce -> ComponentFunc::GetVal x6 per step and ChLoadAddedMass.

Chrono itself is not installed in the image, so this pins nothing about Chrono -- it keeps the adapter code compiling
and running against the interface shapes the reference uses."""
import os
import subprocess

import numpy as np
import pytest

from cases import GOLDEN_DIR, goldens

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "chrono_adapter_test.cpp")
STUB = os.path.join(ROOT, "tests", "cpp", "chrono_stub")


def build(out, opt="-O1"):
    from hydrochrono_amd import build as hb
    hb.build()
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    subprocess.run(["g++", "-std=c++17", opt, "-Wall", "-Werror", "-I", STUB, SRC, "-o", out, "-L", libdir, "-lhydrochrono_amd",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return out


def test_adapter_block_compiles_and_links(tmp_path):
    """CPU: the HYDROCHRONO_AMD_WITH_CHRONO block builds warning-free against the stand-in headers."""
    assert os.path.exists(build(str(tmp_path / "chrono_adapter_test")))


@pytest.mark.gpu
def test_sphere_decay_through_chforce_and_added_mass_load(tmp_path):
    from hydrochrono_amd import build as hb
    exe = build(str(tmp_path / "chrono_adapter_test"), "-O2")
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    ref = goldens()["decay_z_um"] * 1e-6
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "sphere.h5"), str(len(ref))], check=True, capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("MV_CHECK")
    assert float(lines[-1].split()[1]) <= 1e-13  # R += c*M*w through LoadIntLoadResidual_Mv == the Jacobian block times w
    z = np.array([float(line.split()[1]) for line in lines[:-1]])
    assert z.shape == ref.shape
    assert np.max(np.abs(z - ref)) <= 5.1e-7  # the reference's golden decay trajectory (6 printed decimals)
