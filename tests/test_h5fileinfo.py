"""include/hydroc_amd/h5fileinfo.h: the reference's HydroData / H5FileInfo (include/hydroc/h5fileinfo.h:35-260) over the library's
HDF5 reader, host side only (no GPU): every getter against the committed fixtures of the same files -- the sphere's BEMIO file of
the reference and the generated three-body file (tests/golden/make_fixtures.py, make_multibody_bemio.py) -- with the scaling rules of
src/h5fileinfo.cpp (:60-61 added mass x rho, :73-75 magnitudes x rho g, :89-90 excitation IRF x rho g, :310-320 per-access factors)."""
import os
import subprocess

import numpy as np
import pytest

from cases import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from hydrochrono_amd import build as hb
    hb.build()
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    out = str(tmp_path_factory.mktemp("h5fileinfo") / "h5fileinfo_test")
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "h5fileinfo_test.cpp"), "-o", out, "-L", libdir, "-lhydrochrono_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return out


def fixture_of(name, n):
    """(rho, g, depth, w, per-body dict) of a committed fixture, whichever of the two key styles it uses."""
    z = np.load(os.path.join(GOLDEN_DIR, name))
    if "rho" in z.files:
        sc = lambda k: float(z[k].ravel()[0])
        key = {"vol": "disp_vol", "cg": "cg", "cb": "cb", "lin": "linear_restoring_stiffness", "am": "added_mass_inf_freq", "t": "rirf_t", "K": "rirf_K",
               "mag": "excitation_mag", "ph": "excitation_phase", "et": "excitation_irf_t", "ef": "excitation_irf_f"}
        rho, g, depth, w = sc("rho"), sc("g"), sc("water_depth"), z["w"].ravel()
    else:
        key = {"vol": "properties/disp_vol", "cg": "properties/cg", "cb": "properties/cb", "lin": "hydro_coeffs/linear_restoring_stiffness",
               "am": "hydro_coeffs/added_mass/inf_freq", "t": "hydro_coeffs/radiation_damping/impulse_response_fun/t",
               "K": "hydro_coeffs/radiation_damping/impulse_response_fun/K", "mag": "hydro_coeffs/excitation/mag", "ph": "hydro_coeffs/excitation/phase",
               "et": "hydro_coeffs/excitation/impulse_response_fun/t", "ef": "hydro_coeffs/excitation/impulse_response_fun/f"}
        rho, g, w = float(z["simulation_parameters/rho"]), float(z["simulation_parameters/g"]), z["simulation_parameters/w"].ravel()
        depth = np.inf  # the generated files say "infinite" (a string, src/h5fileinfo.cpp:207-220)
    bodies = [{k: np.asarray(z[f"body{b + 1}/{v}"], dtype=np.float64) for k, v in key.items()} for b in range(n)]
    return rho, g, depth, w, bodies


@pytest.mark.parametrize("h5, npz, n", [("sphere.h5", "sphere_bemio.npz", 1), ("three_body.h5", "three_body_bemio.npz", 3),
                                        ("three_body_vlen.h5", "three_body_bemio.npz", 3), ("three_body.h5", "three_body_bemio.npz", 2)])
def test_every_getter_against_the_fixture(exe, tmp_path, h5, npz, n):
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, h5), str(n), out], capture_output=True, text=True)
    if n == 2:
        # two bodies of a three-body file: K is {6, 18, S}, not {6, 12, S} -- the reader refuses (the reference would read a tensor
        # of the file's shape and index it as if it had 12 columns)
        assert r.returncode != 0 and "added_mass/inf_freq must be 6 x 6N" in r.stderr
        return
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "End"
    rho, g, depth, w, bodies = fixture_of(npz, n)
    d = np.fromfile(out)
    pos = 0

    def take(k):
        nonlocal pos
        v = d[pos:pos + k]
        pos += k
        return v

    D, S = 6 * n, bodies[0]["t"].size
    assert take(1)[0] == rho and take(1)[0] == g
    assert take(1)[0] == depth
    assert list(take(3)) == [6, D, S]
    assert np.array_equal(take(S), bodies[0]["t"])
    for b, q in enumerate(bodies):
        assert take(1)[0] == float(q["vol"].ravel()[0])
        assert np.array_equal(take(3), q["cg"]) and np.array_equal(take(3), q["cb"])
        assert np.array_equal(take(36), q["lin"].ravel())
        assert np.array_equal(take(36), q["lin"].ravel() * rho * g)      # GetHydrostaticStiffnessVal
        assert list(take(2)) == [6, D]
        assert np.array_equal(take(6 * D), q["am"].ravel() * rho)        # scaled when read
        assert np.array_equal(take(6 * D * S), q["K"].ravel() * rho)     # GetRIRFVal: per access
        assert take(1)[0] == b
        assert take(1)[0] == q["t"][1] - q["t"][0]
        assert take(1)[0] == 1.0
        nw = int(take(1)[0])
        assert nw == w.size and np.array_equal(take(nw), w)
        assert np.array_equal(take(6 * nw), q["mag"].reshape(6, nw).ravel() * (rho * g))
        assert np.array_equal(take(6 * nw), q["ph"].reshape(6, nw).ravel())
        L = int(take(1)[0])
        assert L == q["et"].size and np.array_equal(take(L), q["et"])
        assert np.array_equal(take(6 * L), q["ef"].reshape(6, L).ravel() * (rho * g))
        assert take(1)[0] == 1.0
    assert pos == d.size


def test_errors_are_the_reference_exception_types(exe):
    r = subprocess.run([exe, "--errors", os.path.join(GOLDEN_DIR, "sphere.h5")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines())
    assert lines["MISSING"].startswith("Unable to open/read HDF5 hydro data file: /nonexistent/dir/nothing.h5")  # src/h5fileinfo.cpp:167-179
    assert "no-throw" not in lines["TOO_MANY"] and "no-throw" not in lines["RANGE"]


def test_spectrum_helpers(exe):
    """PiersonMoskowitzSpectrumHz / JONSWAPSpectrumHz of hydroc_amd/wave_types.h against the formulas of src/wave_types.cpp:679-715
    (S_PM = 1.25 Tp^-4 (Hs/2)^2 f^-5 exp(-1.25 Tp^-4 f^-4); x gamma^exp(-(f Tp - 1)^2 / (2 sigma^2)), sigma = 0.07 up to 1/Tp, 0.09
    above; optional x (1 - 0.287 ln gamma)); the frequency vector comes back sorted."""
    r = subprocess.run([exe, "--spectrum"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    a = np.array([[float(x) for x in ln.split()] for ln in r.stdout.strip().splitlines()])
    f = a[:, 0]
    assert list(f) == sorted([0.31, 0.05, 0.125, 0.2, 0.08])
    Hs, Tp = 2.0, 8.0
    pm = 1.25 * Tp ** -4 * (Hs / 2) ** 2 * f ** -5.0 * np.exp(-1.25 * Tp ** -4 * f ** -4.0)
    sigma = np.where(f <= 1 / Tp, 0.07, 0.09)
    peak = np.exp(-(f * Tp - 1) ** 2 / (2 * sigma ** 2))
    assert np.allclose(a[:, 1], pm, rtol=1e-14, atol=0)
    assert np.allclose(a[:, 2], pm * 3.3 ** peak, rtol=1e-14, atol=0)
    assert np.allclose(a[:, 3], pm * 2.0 ** peak * (1 - 0.287 * np.log(2.0)), rtol=1e-14, atol=0)
