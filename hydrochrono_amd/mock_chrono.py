"""Stand-ins for the Project Chrono time loop that drives the force path (Chrono itself is not available here).

`run_heave_1dof` reproduces what Chrono's default Euler-implicit-linearized stepper does for a single heaving body:
the force used for step n -> n+1 is F(x_n, v_n, t_n), then symplectic Euler.  The added-mass load
(ChLoadAddedMass) enters through the mass matrix: m + rho*Ainf_33.  Works with any object exposing
`step(t,pos,rpy,linvel,angvel)` and `added_mass_matrix()` -- the GPU path (HydroForces) and the CPU oracle alike.
`PrescribedMotion` is the benchmark driver: q_{b,d}(t) = q0 + 0.1 sin(w t + phi), analytic velocity.
"""
import numpy as np


def run_heave_1dof(hydro, mass, g, pto_damping, z0, dt, nsteps):
    a33 = hydro.added_mass_matrix()[2, 2]
    z, v = float(z0), 0.0
    out = np.empty(nsteps)
    pos, rpy, lv, av = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
    for n in range(nsteps):
        pos[2], lv[2] = z, v
        fz = hydro.step(n * dt, pos, rpy, lv, av)[2]
        F = fz - mass * g - pto_damping * v
        v += dt * F / (mass + a33)
        z += dt * v
        out[n] = z
    return out


class PrescribedMotion:
    """Deterministic 6N-DoF motion about the bodies' rest poses."""

    def __init__(self, num_bodies, rest_pos, seed=20251031, amplitude=0.1):
        rng = np.random.default_rng(seed)
        self.N = num_bodies
        self.q0 = np.zeros((num_bodies, 6))
        self.q0[:, :3] = np.asarray(rest_pos, dtype=np.float64).reshape(num_bodies, 3)
        self.omega = rng.uniform(0.4, 2.5, size=(num_bodies, 6))
        self.phi = rng.uniform(0.0, 2 * np.pi, size=(num_bodies, 6))
        self.amp = amplitude * np.ones((num_bodies, 6))
        self.amp[:, 3:] *= 0.5  # radians

    def state(self, t):
        """Returns pos[N,3], rpy[N,3], linvel[N,3], angvel[N,3]."""
        q = self.q0 + self.amp * np.sin(self.omega * t + self.phi)
        v = self.amp * self.omega * np.cos(self.omega * t + self.phi)
        return q[:, :3].copy(), q[:, 3:].copy(), v[:, :3].copy(), v[:, 3:].copy()

    def packed(self, t):
        """12N vector pos|rpy|linvel|angvel, the layout hc_step_device expects."""
        p, r, lv, av = self.state(t)
        return np.concatenate([p.reshape(-1), r.reshape(-1), lv.reshape(-1), av.reshape(-1)])

    def velocity6(self, t):
        """[6N] velocity vector in DoF order (what the history stores)."""
        v = self.amp * self.omega * np.cos(self.omega * t + self.phi)
        return v.reshape(-1)
