#!/usr/bin/env python3
"""Extract parity fixtures from the reference's own test data (run in the BUILD container only).

Produces (committed, small, data only -- no reference source text):
  tests/golden/sphere_bemio.npz     flat float64 arrays of the BEMIO datasets that
                                    H5FileInfo::ReadH5Data reads (src/h5fileinfo.cpp:35-90)
                                    from demos/sphere/hydroData/sphere.h5, file order kept.
  tests/golden/sphere_goldens.npz   heave trajectories of the reference regression suite
                                    (tests/regression/reference_data/sphere/**/hc_ref_*.txt),
                                    stored as int32 micro-metres (the files print 6 decimals).
  tests/golden/iea_sphere_decay.npz time / heave position, velocity, acceleration of the YAML-runner regression case
                                    tests/regression/run_hydrochrono/iea_sphere/decay/expected/results.still.h5
                                    (dt = 0.01 != dt_rirf = 0.015, gravity 9.8, HHT; its iea_sphere.h5 is byte-identical
                                    to demos/sphere/hydroData/sphere.h5, i.e. sphere_bemio.npz).

Needs h5dump (HDF5 1.10 tools, /opt/conda/bin) and /root/reference; neither exists on the GPU box,
which only ever reads the .npz files.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = os.environ.get("HC_REFERENCE_ROOT", "/root/reference")
H5 = os.path.join(REF, "demos/sphere/hydroData/sphere.h5")
HERE = os.path.dirname(os.path.abspath(__file__))
H5DUMP = os.environ.get("H5DUMP", "/opt/conda/bin/h5dump")

DATASETS = {
    # key in npz                      : (hdf5 path, shape)
    "rho": ("/simulation_parameters/rho", (1,)),
    "g": ("/simulation_parameters/g", (1,)),
    "water_depth": ("/simulation_parameters/water_depth", (1,)),
    "w": ("/simulation_parameters/w", (240,)),
    "body1/disp_vol": ("/body1/properties/disp_vol", (1,)),
    "body1/cg": ("/body1/properties/cg", (3,)),
    "body1/cb": ("/body1/properties/cb", (3,)),
    "body1/linear_restoring_stiffness": ("/body1/hydro_coeffs/linear_restoring_stiffness", (6, 6)),
    "body1/added_mass_inf_freq": ("/body1/hydro_coeffs/added_mass/inf_freq", (6, 6)),
    "body1/rirf_t": ("/body1/hydro_coeffs/radiation_damping/impulse_response_fun/t", (1001,)),
    "body1/rirf_K": ("/body1/hydro_coeffs/radiation_damping/impulse_response_fun/K", (6, 6, 1001)),
    "body1/excitation_mag": ("/body1/hydro_coeffs/excitation/mag", (6, 1, 240)),
    "body1/excitation_phase": ("/body1/hydro_coeffs/excitation/phase", (6, 1, 240)),
    "body1/excitation_irf_t": ("/body1/hydro_coeffs/excitation/impulse_response_fun/t", (1001,)),
    "body1/excitation_irf_f": ("/body1/hydro_coeffs/excitation/impulse_response_fun/f", (6, 1, 1001)),
}


def dump(path, shape):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", path, "-b", "LE", "-o", tmp.name, H5],
                       check=True, stdout=subprocess.DEVNULL)
        arr = np.fromfile(tmp.name, dtype="<f8")
    assert arr.size == int(np.prod(shape)), (path, arr.size, shape)
    return arr.reshape(shape)


def dump_file(h5file, path, shape):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", path, "-b", "LE", "-o", tmp.name, h5file], check=True, stdout=subprocess.DEVNULL)
        arr = np.fromfile(tmp.name, dtype="<f8")
    assert arr.size == int(np.prod(shape)), (path, arr.size, shape)
    return arr.reshape(shape)


def iea_sphere_decay():
    base = os.path.join(REF, "tests/regression/run_hydrochrono/iea_sphere")
    with open(os.path.join(base, "assets/hydroData/iea_sphere.h5"), "rb") as a, open(H5, "rb") as b:
        assert a.read() == b.read(), "iea_sphere.h5 is expected to be the sphere BEMIO file"
    res = os.path.join(base, "decay/expected/results.still.h5")
    n = 4000
    out = {"time": dump_file(res, "/results/time/time", (n,))}
    for name in ("position", "velocity", "acceleration"):
        a = dump_file(res, f"/results/model/bodies/body1/{name}", (n, 3))
        assert np.all(a[:, :2] == 0.0)  # prismatic heave joint
        out[name + "_z"] = a[:, 2].copy()
    # case parameters, transcribed as data from the case's model / simulation yaml
    out["mass"], out["gravity_z"], out["time_step"] = np.array(261800.0), np.array(-9.8), np.array(0.01)
    np.savez_compressed(os.path.join(HERE, "iea_sphere_decay.npz"), **out)


def read_heave(relpath, skip):
    rows = []
    with open(os.path.join(REF, relpath)) as fh:
        for i, line in enumerate(fh):
            if i < skip:
                continue
            parts = line.split()
            if len(parts) != 2:
                continue
            rows.append((float(parts[0]), float(parts[1])))
    a = np.array(rows)
    t = a[:, 0]
    z = np.rint(a[:, 1] * 1e6).astype(np.int32)
    assert np.max(np.abs(z * 1e-6 - a[:, 1])) < 1e-9
    return t, z


def main():
    bem = {k: dump(p, s) for k, (p, s) in DATASETS.items()}
    np.savez_compressed(os.path.join(HERE, "sphere_bemio.npz"), **bem)

    gold = {}
    base = "tests/regression/reference_data/sphere"
    t, z = read_heave(f"{base}/decay/hc_ref_sphere_decay.txt", 1)
    gold["decay_t0"], gold["decay_dt"], gold["decay_z_um"] = t[0], 0.015, z
    assert np.allclose(t, 0.015 * np.arange(1, len(t) + 1), atol=1e-9)
    for k in range(1, 11):
        t, z = read_heave(f"{base}/reg_waves/hc_ref_sphere_reg_waves_{k}.txt", 5)
        assert np.allclose(t, 0.015 * np.arange(1, len(t) + 1), atol=1e-9), k
        gold[f"reg_waves_{k}_z_um"] = z
    t, z = read_heave(f"{base}/irreg_waves/hc_ref_sphere_irreg_waves.txt", 2)
    assert np.allclose(t, 0.015 * np.arange(1, len(t) + 1), atol=1e-9)
    gold["irreg_waves_z_um"] = z
    # Case parameters, transcribed as data from tests/regression/sphere/reg_waves/sphere_reg_waves_test.cpp:23-30
    gold["reg_wave_amp"] = np.array([0.177, 0.314, 0.380, 0.491, 0.706, 0.961, 1.256, 1.589, 1.962, 2.374])
    gold["reg_wave_omega"] = np.array([2.094395102, 1.570796327, 1.427996661, 1.256637061, 1.047197551,
                                       0.897597901, 0.785398163, 0.698131701, 0.628318531, 0.571198664])
    gold["reg_wave_pto_damping"] = np.array([398736.034, 118149.758, 90080.857, 161048.558, 322292.419,
                                             479668.979, 633979.761, 784083.286, 932117.647, 1077123.445])
    np.savez_compressed(os.path.join(HERE, "sphere_goldens.npz"), **gold)
    iea_sphere_decay()
    for f in ("sphere_bemio.npz", "sphere_goldens.npz", "iea_sphere_decay.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
