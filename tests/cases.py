"""Test inputs shared by the oracle tests and the GPU parity tests.

A *case* is the set of raw BEMIO arrays H5FileInfo::ReadH5Data reads (src/h5fileinfo.cpp:35-90),
unscaled and in file order, so the oracle and the product ingest exactly the same numbers.
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SPHERE_MASS = 261.8e3  # tests/regression/sphere/demo_sphere_decay.cpp: SetMass(261.8e3)
SPHERE_G = 9.81        # system.SetGravitationalAcceleration(0,0,-9.81)
SPHERE_DT = 0.015


def sphere_case():
    z = np.load(os.path.join(GOLDEN_DIR, "sphere_bemio.npz"))
    body = dict(
        disp_vol=float(z["body1/disp_vol"][0]), cg=z["body1/cg"], cb=z["body1/cb"],
        lin=z["body1/linear_restoring_stiffness"], added_mass_inf=z["body1/added_mass_inf_freq"],
        rirf_t=z["body1/rirf_t"], rirf_K=z["body1/rirf_K"], w=z["w"],
        ex_mag=z["body1/excitation_mag"], ex_phase=z["body1/excitation_phase"],
        ex_irf_t=z["body1/excitation_irf_t"], ex_irf_f=z["body1/excitation_irf_f"])
    return dict(N=1, rho=float(z["rho"][0]), g=float(z["g"][0]), water_depth=float(z["water_depth"][0]), bodies=[body])


def three_body_case():
    """The generated three-body BEMIO fixture (tests/golden/make_multibody_bemio.py): dataset shapes of a multi-body BEMIO
    file (added_mass/inf_freq {6,18}, impulse_response_fun/K {6,18,S}) and water_depth = "infinite"."""
    return multi_body_fixture("three_body", 3)


def four_body_case():
    """four_body.h5 of the same generator (D = 24): read by the multi-shard C++ test with 1, 2 and 4 row shards."""
    return multi_body_fixture("four_body", 4)


def multi_body_fixture(stem, N):
    z = np.load(os.path.join(GOLDEN_DIR, stem + "_bemio.npz"))
    bodies = []
    for b in range(1, N + 1):
        p = f"body{b}/"
        bodies.append(dict(
            disp_vol=float(z[p + "properties/disp_vol"]), cg=z[p + "properties/cg"], cb=z[p + "properties/cb"],
            lin=z[p + "hydro_coeffs/linear_restoring_stiffness"], added_mass_inf=z[p + "hydro_coeffs/added_mass/inf_freq"],
            rirf_t=z[p + "hydro_coeffs/radiation_damping/impulse_response_fun/t"],
            rirf_K=z[p + "hydro_coeffs/radiation_damping/impulse_response_fun/K"], w=z["simulation_parameters/w"],
            ex_mag=z[p + "hydro_coeffs/excitation/mag"], ex_phase=z[p + "hydro_coeffs/excitation/phase"],
            ex_irf_t=z[p + "hydro_coeffs/excitation/impulse_response_fun/t"],
            ex_irf_f=z[p + "hydro_coeffs/excitation/impulse_response_fun/f"]))
    return dict(N=N, rho=float(z["simulation_parameters/rho"]), g=float(z["simulation_parameters/g"]), water_depth=float("inf"),
                bodies=bodies)


def goldens():
    return np.load(os.path.join(GOLDEN_DIR, "sphere_goldens.npz"))


def iea_sphere_decay():
    """The YAML-runner regression case tests/regression/run_hydrochrono/iea_sphere/decay (reference's expected
    results.still.h5): recorded heave position / velocity / acceleration at dt = 0.01 (!= dt_rirf = 0.015), gravity 9.8,
    HHT integrator; its BEMIO file is byte-identical to the sphere's."""
    return dict(np.load(os.path.join(GOLDEN_DIR, "iea_sphere_decay.npz")))


def iea_sphere_residual(forces_z, rec, case):
    """Soft check of the true-interpolation branch: Newton's law of the recorded motion,
    (m + rho*Ainf_33) a_z + m g, against hs_z - rad_z recomputed from the recorded position and velocity columns.
    Returns max |difference| as an acceleration.  HHT's alpha-weighting and the reference evaluating the force at each
    step's predictor state limit the agreement to ~7.4e-3 m/s^2 (3.8e-3 of max |a|); a wrong gravity source, sign
    or interpolation is 10x-1000x worse."""
    m, g = float(rec["mass"]), -float(rec["gravity_z"])
    a33 = case["rho"] * np.asarray(case["bodies"][0]["added_mass_inf"]).reshape(6, 6)[2, 2]
    return float(np.max(np.abs(rec["acceleration_z"] - (np.asarray(forces_z) - m * g) / (m + a33))))


def load_into_oracle(case, oracle_cls=None):
    from oracle import Oracle
    o = (oracle_cls or Oracle)(case["N"])
    o.set_simulation_parameters(case["rho"], case["g"], case["water_depth"])
    for b, bd in enumerate(case["bodies"]):
        o.set_body(b, bd["disp_vol"], bd["cg"], bd["cb"], bd["lin"], bd["added_mass_inf"], bd["rirf_t"], bd["rirf_K"])
        if "w" in bd:
            o.set_body_excitation_rao(b, bd["w"], bd["ex_mag"], bd["ex_phase"])
        if "ex_irf_t" in bd:
            o.set_body_excitation_irf(b, bd["ex_irf_t"], bd["ex_irf_f"])
    o.construct()
    if "g_sys" in case:
        o.set_gravity(case["g_sys"])
    return o
