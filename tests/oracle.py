"""ctypes binding of the CPU oracle (oracle/libhc_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (hydrochrono_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "libhc_oracle.so")

_dp = C.POINTER(C.c_double)


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def build_oracle(force=False):
    src = os.path.join(_ROOT, "oracle", "hc_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(_ROOT, "oracle"), "-B", "libhc_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_wave_number.restype = C.c_double
        _lib.orc_last_error.argtypes = [C.c_void_p]
        _lib.orc_destroy.argtypes = [C.c_void_p]
        # the reference's static schedule over IRF steps gets slower beyond a few dozen threads (see bench.py's sweep);
        # tests pin a small count, bench.py's cpu_baseline sets its own
        _lib.orc_set_num_threads(C.c_int(min(8, os.cpu_count() or 1)))
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[oracle rc={code}] {msg}")
        self.code = code


class Oracle:
    """One reference `TestHydro` instance (src/hydro_forces.cpp) restated on the CPU."""

    def __init__(self, num_bodies):
        self.L = lib()
        self.N = int(num_bodies)
        self.D = 6 * self.N
        self.ctx = C.c_void_p(self.L.orc_create(self.N))

    def close(self):
        if self.ctx:
            self.L.orc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc, self.L.orc_last_error(self.ctx).decode())

    # ---- ingest ----
    def set_simulation_parameters(self, rho, g, water_depth):
        self._chk(self.L.orc_set_simulation_parameters(self.ctx, C.c_double(rho), C.c_double(g), C.c_double(water_depth)))

    def set_body(self, b, disp_vol, cg, cb, lin, added_mass_inf, rirf_t, rirf_K):
        cg = np.ascontiguousarray(cg, dtype=np.float64).reshape(3)
        cb = np.ascontiguousarray(cb, dtype=np.float64).reshape(3)
        lin = np.ascontiguousarray(lin, dtype=np.float64).reshape(36)
        A = np.ascontiguousarray(added_mass_inf, dtype=np.float64).reshape(6 * self.D)
        t = np.ascontiguousarray(rirf_t, dtype=np.float64).reshape(-1)
        K = np.ascontiguousarray(rirf_K, dtype=np.float64)
        assert K.size == 6 * self.D * t.size
        self._chk(self.L.orc_set_body(self.ctx, b, C.c_double(disp_vol), _p(cg), _p(cb), _p(lin), _p(A), _p(t),
                                      C.c_int(t.size), _p(K.reshape(-1))))

    def set_body_excitation_rao(self, b, w, mag, phase):
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
        mag = np.ascontiguousarray(mag, dtype=np.float64).reshape(-1)
        phase = np.ascontiguousarray(phase, dtype=np.float64).reshape(-1)
        assert mag.size == 6 * w.size and phase.size == 6 * w.size
        self._chk(self.L.orc_set_body_excitation_rao(self.ctx, b, _p(w), C.c_int(w.size), _p(mag), _p(phase)))

    def set_body_excitation_irf(self, b, t, f):
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
        f = np.ascontiguousarray(f, dtype=np.float64).reshape(-1)
        assert f.size == 6 * t.size
        self._chk(self.L.orc_set_body_excitation_irf(self.ctx, b, _p(t), C.c_int(t.size), _p(f)))

    def construct(self):
        self._chk(self.L.orc_construct(self.ctx))

    def set_gravity(self, g3):
        g3 = np.ascontiguousarray(g3, dtype=np.float64).reshape(3)
        self._chk(self.L.orc_set_gravity(self.ctx, _p(g3)))

    # ---- waves ----
    def add_waves_none(self, num_bodies=None):
        self._chk(self.L.orc_add_waves_none(self.ctx, C.c_int(self.N if num_bodies is None else num_bodies)))

    def add_waves_regular(self, amplitude, omega, num_bodies=None):
        self._chk(self.L.orc_add_waves_regular(self.ctx, C.c_int(self.N if num_bodies is None else num_bodies),
                                               C.c_double(amplitude), C.c_double(omega)))

    def add_waves_irregular(self, simulation_dt, simulation_duration, ramp_duration=0.0, wave_height=0.0,
                            wave_period=0.0, frequency_min=0.001, frequency_max=1.0, nfrequencies=0,
                            peak_enhancement_factor=1.0, is_normalized=False, seed=1, num_bodies=None):
        self._chk(self.L.orc_add_waves_irregular(
            self.ctx, C.c_int(self.N if num_bodies is None else num_bodies), C.c_double(simulation_dt),
            C.c_double(simulation_duration), C.c_double(ramp_duration), C.c_double(wave_height), C.c_double(wave_period),
            C.c_double(frequency_min), C.c_double(frequency_max), C.c_double(nfrequencies),
            C.c_double(peak_enhancement_factor), C.c_int(int(is_normalized)), C.c_int(seed)))

    def set_convolution_mode(self, mode):
        self._chk(self.L.orc_set_convolution_mode(self.ctx, C.c_int(mode)))

    def set_tapered_direct_options(self, smoothing=0, window_length=5, rirf_end_time=-1.0, taper_start_percent=0.8,
                                   taper_end_percent=1.0, taper_final_amplitude=0.0):
        self._chk(self.L.orc_set_tapered_direct_options(
            self.ctx, C.c_int(smoothing), C.c_int(window_length), C.c_double(rirf_end_time),
            C.c_double(taper_start_percent), C.c_double(taper_end_percent), C.c_double(taper_final_amplitude)))

    def set_diagnostics(self, export_plot_csv, directory=""):
        """opts.export_plot_csv + SetDiagnosticsOutputDirectory (include/hydroc/hydro_forces.h:258,269)."""
        self._chk(self.L.orc_set_diagnostics(self.ctx, C.c_int(int(export_plot_csv)), str(directory).encode()))

    # ---- stepping ----
    def step(self, t, pos, rpy, linvel, angvel):
        out = np.empty(self.D)
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (pos, rpy, linvel, angvel)]
        assert all(x.size == 3 * self.N for x in a)
        self._chk(self.L.orc_step(self.ctx, C.c_double(t), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out)))
        return out

    def components(self):
        hs, rad, wv = np.empty(self.D), np.empty(self.D), np.empty(self.D)
        self._chk(self.L.orc_get_force_components(self.ctx, _p(hs), _p(rad), _p(wv)))
        return hs, rad, wv

    def compute_radiation(self, t, linvel, angvel):
        out = np.empty(self.D)
        lv = np.ascontiguousarray(linvel, dtype=np.float64).reshape(-1)
        av = np.ascontiguousarray(angvel, dtype=np.float64).reshape(-1)
        self._chk(self.L.orc_compute_radiation(self.ctx, C.c_double(t), _p(lv), _p(av), _p(out)))
        return out

    def rirf_val(self, row, col, st):
        v = C.c_double()
        self._chk(self.L.orc_get_rirf_val(self.ctx, row, col, st, C.byref(v)))
        return v.value

    def rirf_width(self, S):
        out = np.empty(S)
        self._chk(self.L.orc_get_rirf_width(self.ctx, _p(out)))
        return out

    def history_size(self):
        return self.L.orc_history_size(self.ctx)

    def prefill_history(self, times_newest_first, vel):
        t = np.ascontiguousarray(times_newest_first, dtype=np.float64)
        v = np.ascontiguousarray(vel, dtype=np.float64)
        assert v.shape == (t.size, self.D)
        self._chk(self.L.orc_prefill_history(self.ctx, C.c_int(t.size), _p(t), _p(v.reshape(-1))))

    # ---- init products ----
    def irreg_sizes(self):
        L, nf, nt = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.orc_irreg_sizes(self.ctx, C.byref(L), C.byref(nf), C.byref(nt)))
        return L.value, nf.value, nt.value

    def irreg_irf(self, b=0):
        Lb = C.c_int()
        self._chk(self.L.orc_irreg_irf_size(self.ctx, C.c_int(b), C.byref(Lb)))
        L = Lb.value
        t, w, v = np.empty(L), np.empty(L), np.empty((6, L))
        self._chk(self.L.orc_irreg_get_irf(self.ctx, b, _p(t), _p(w), _p(v.reshape(-1))))
        return t, w, v

    def irreg_spectrum(self):
        _, nf, _ = self.irreg_sizes()
        arrs = [np.empty(nf) for _ in range(5)]
        self._chk(self.L.orc_irreg_get_spectrum(self.ctx, *[_p(a) for a in arrs]))
        return dict(zip(("f", "S", "df", "phase", "k"), arrs))

    def irreg_eta(self):
        _, _, nt = self.irreg_sizes()
        t, e = np.empty(nt), np.empty(nt)
        self._chk(self.L.orc_irreg_get_eta(self.ctx, _p(t), _p(e)))
        return t, e

    def regular_coeffs(self):
        mag, ph, k = np.empty(self.D), np.empty(self.D), C.c_double()
        self._chk(self.L.orc_regular_get_coeffs(self.ctx, _p(mag), _p(ph), C.byref(k)))
        return mag, ph, k.value

    def added_mass_matrix(self):
        M = np.empty((self.D, self.D))
        self._chk(self.L.orc_added_mass_matrix(self.ctx, _p(M.reshape(-1))))
        return M

    def added_mass_mv(self, R, w, c):
        R = np.ascontiguousarray(R, dtype=np.float64).copy()
        w = np.ascontiguousarray(w, dtype=np.float64)
        self._chk(self.L.orc_added_mass_mv(self.ctx, _p(R), _p(w), C.c_double(c), C.c_int(R.size)))
        return R

    # ---- optimised CPU variant (bench cpu_baseline only) ----
    def flat_prepare(self):
        self._chk(self.L.orc_flat_prepare(self.ctx))

    def flat_step(self, t, pos, rpy, linvel, angvel):
        out = np.empty(self.D)
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (pos, rpy, linvel, angvel)]
        self._chk(self.L.orc_flat_step(self.ctx, C.c_double(t), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out)))
        return out

    def run_heave_1dof(self, mass, g, pto_damping, z0, dt, nsteps, want_force=False):
        z = np.empty(nsteps)
        fz = np.empty(nsteps) if want_force else None
        self._chk(self.L.orc_run_heave_1dof(self.ctx, C.c_double(mass), C.c_double(g), C.c_double(pto_damping),
                                            C.c_double(z0), C.c_double(dt), C.c_int(nsteps), _p(z), _p(fz)))
        return (z, fz) if want_force else z


# ---- free functions (restated third-party arithmetic) ----
def dense_mv_seconds(M, w, c, reps):
    """Seconds per product of the oracle's `R += c * M * w` loop (src/chloadaddedmass.cpp:55-70) on an arbitrary D x D matrix."""
    import time
    M = np.ascontiguousarray(M, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    D = w.size
    R = np.zeros(D)
    L = lib()
    L.orc_dense_mv.restype = None
    L.orc_dense_mv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
    L.orc_dense_mv(M.ctypes.data, D, w.ctypes.data, float(c), R.ctypes.data, max(1, reps // 10))  # warm
    a = time.perf_counter()
    L.orc_dense_mv(M.ctypes.data, D, w.ctypes.data, float(c), R.ctypes.data, reps)
    return (time.perf_counter() - a) / reps, R


def linspaced(n, lo, hi):
    out = np.empty(n)
    lib().orc_linspaced(C.c_int(n), C.c_double(lo), C.c_double(hi), _p(out))
    return out


def mt19937_raw(seed, n):
    out = np.empty(n, dtype=np.uint32)
    lib().orc_mt19937_raw(C.c_uint(seed), C.c_int(n), out.ctypes.data_as(C.POINTER(C.c_uint)))
    return out


def uniform_phases(seed, n):
    out = np.empty(n)
    lib().orc_uniform_phases(C.c_uint(seed), C.c_int(n), _p(out))
    return out


def spline_resample(vals, n_new):
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    assert vals.shape[0] == 6
    out = np.empty((6, n_new))
    rc = lib().orc_spline_resample(C.c_int(vals.shape[1]), _p(vals.reshape(-1)), C.c_int(n_new), _p(out.reshape(-1)))
    if rc:
        raise OracleError(rc, "spline resample failed")
    return out


def get_lower_index(value, ticks):
    ticks = np.ascontiguousarray(ticks, dtype=np.float64)
    out = C.c_long()
    rc = lib().orc_get_lower_index(C.c_double(value), _p(ticks), C.c_int(ticks.size), C.byref(out))
    if rc:
        raise OracleError(rc, "get_lower_index threw")
    return out.value


def wave_number(omega, depth, g):
    return lib().orc_wave_number(C.c_double(omega), C.c_double(depth), C.c_double(g))


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(n))
