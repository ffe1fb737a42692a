"""HydroForces -- Python mirror of the reference's `TestHydro` surface (include/hydroc/hydro_forces.h:164-285),
implemented purely by calls into the C ABI (include/hydrochrono_amd.h).  No arithmetic happens here."""
import ctypes as C

import numpy as np

from . import capi


class HydroError(RuntimeError):
    """A non-zero hc_status; .status holds it (1 = the reference throws std::runtime_error, 2 = std::out_of_range)."""

    def __init__(self, status, message):
        super().__init__(f"[hc_status={status}] {message}")
        self.status = status


def _dp(a):
    return None if a is None else a.ctypes.data_as(capi.c_double_p)


def _arr(x, n=None):
    a = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class HydroForces:
    def __init__(self, num_bodies, device=0, body_range=None):
        self.lib = capi.load()
        self.N = int(num_bodies)
        self.D = 6 * self.N
        self.b0, self.b1 = (0, self.N) if body_range is None else (int(body_range[0]), int(body_range[1]))
        self.n_local = self.b1 - self.b0
        self.D_local = 6 * self.n_local
        ctx = C.c_void_p()
        rc = self.lib.hc_create_sharded(self.N, self.b0, self.b1, int(device), C.byref(ctx))
        if rc != capi.HC_OK:
            raise HydroError(rc, self.lib.hc_last_error(None).decode())
        self.ctx = ctx

    # -- plumbing --
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.hc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != capi.HC_OK:
            raise HydroError(rc, self.lib.hc_last_error(self.ctx).decode())

    # -- ingest (H5FileInfo::ReadH5Data) --
    def load_bemio_h5(self, path):
        self._chk(self.lib.hc_load_bemio_h5(self.ctx, str(path).encode()))

    def set_simulation_parameters(self, rho, g, water_depth):
        self._chk(self.lib.hc_set_simulation_parameters(self.ctx, rho, g, water_depth))

    def set_body(self, b, disp_vol, cg, cb, lin, added_mass_inf, rirf_t, rirf_K):
        cg, cb, lin = _arr(cg, 3), _arr(cb, 3), _arr(lin, 36)
        A = _arr(added_mass_inf, 6 * self.D)
        t = _arr(rirf_t)
        K = _arr(rirf_K, 6 * self.D * t.size)
        self._chk(self.lib.hc_set_body_properties(self.ctx, b, float(disp_vol), _dp(cg), _dp(cb)))
        self._chk(self.lib.hc_set_hydrostatic_stiffness(self.ctx, b, _dp(lin)))
        self._chk(self.lib.hc_set_added_mass_inf(self.ctx, b, _dp(A)))
        self._chk(self.lib.hc_set_rirf(self.ctx, b, _dp(t), t.size, _dp(K)))

    def set_body_excitation_rao(self, b, w, mag, phase):
        w = _arr(w)
        mag, phase = _arr(mag, 6 * w.size), _arr(phase, 6 * w.size)
        self._chk(self.lib.hc_set_excitation_rao(self.ctx, b, _dp(w), w.size, _dp(mag), _dp(phase)))

    def set_body_excitation_irf(self, b, t, f):
        t = _arr(t)
        f = _arr(f, 6 * t.size)
        self._chk(self.lib.hc_set_excitation_irf(self.ctx, b, _dp(t), t.size, _dp(f)))

    def synth_fill(self, seed, S, dt_rirf, n_exc=0, dt_exc=0.0):
        self._chk(self.lib.hc_synth_fill(self.ctx, seed, S, dt_rirf, n_exc, dt_exc))

    def finalize(self):
        self._chk(self.lib.hc_finalize(self.ctx))

    @classmethod
    def from_case(cls, case, device=0, body_range=None):
        """Build from a raw-array case dict (see tests/cases.py, hydrochrono_amd/synthetic.py)."""
        h = cls(case["N"], device=device, body_range=body_range)
        h.set_simulation_parameters(case["rho"], case["g"], case["water_depth"])
        for b, bd in enumerate(case["bodies"]):
            h.set_body(b, bd["disp_vol"], bd["cg"], bd["cb"], bd["lin"], bd["added_mass_inf"], bd["rirf_t"], bd["rirf_K"])
            if "w" in bd:
                h.set_body_excitation_rao(b, bd["w"], bd["ex_mag"], bd["ex_phase"])
            if "ex_irf_t" in bd:
                h.set_body_excitation_irf(b, bd["ex_irf_t"], bd["ex_irf_f"])
        h.finalize()
        if "g_sys" in case:
            h.set_gravity(case["g_sys"])
        return h

    @classmethod
    def from_hydro_yaml(cls, yaml_path, system_body_names, timestep, sim_duration, ramp_duration=0.0, device=0):
        """ReadHydroYAML + SetupHydroFromYAML (src/hydro_yaml_parser.cpp, src/setup_hydro_from_yaml.cpp:126-193).
        Returns (HydroForces, matched_index) where matched_index[k] is the position of hydro body k in system_body_names."""
        lib = capi.load()
        cfg = C.c_void_p()
        err = C.create_string_buffer(2048)
        rc = lib.hc_yaml_read(str(yaml_path).encode(), C.byref(cfg), err, 2048)
        if rc != capi.HC_OK:
            raise HydroError(rc, err.value.decode())
        try:
            names = (C.c_char_p * len(system_body_names))(*[n.encode() for n in system_body_names])
            matched = (C.c_int * max(1, len(system_body_names)))()
            nm = C.c_int()
            ctx = C.c_void_p()
            rc = lib.hc_create_from_hydro_yaml(cfg, names, len(system_body_names), timestep, sim_duration, ramp_duration, int(device),
                                               C.byref(ctx), matched, C.byref(nm), err, 2048)
            if rc != capi.HC_OK:
                raise HydroError(rc, err.value.decode())
        finally:
            lib.hc_yaml_free(cfg)
        self = cls.__new__(cls)
        self.lib, self.ctx = lib, ctx
        self.N = nm.value
        self.D = 6 * self.N
        self.b0, self.b1, self.n_local, self.D_local = 0, self.N, self.N, self.D
        return self, list(matched[: nm.value])

    # -- configuration --
    def set_gravity(self, g3):
        g3 = _arr(g3, 3)
        self._chk(self.lib.hc_set_gravity(self.ctx, _dp(g3)))

    def add_waves_none(self, num_bodies=None):
        self._chk(self.lib.hc_set_wave_none(self.ctx, self.N if num_bodies is None else int(num_bodies)))

    def add_waves_regular(self, amplitude, omega, num_bodies=None):
        self._chk(self.lib.hc_set_wave_regular(self.ctx, self.N if num_bodies is None else int(num_bodies), amplitude, omega))

    def add_waves_irregular(self, simulation_dt, simulation_duration, ramp_duration=0.0, wave_height=0.0, wave_period=0.0,
                            frequency_min=0.001, frequency_max=1.0, nfrequencies=0, peak_enhancement_factor=1.0,
                            is_normalized=False, seed=1, num_bodies=None, spectral=False):
        p = capi.IrregularWaveParams()
        self.lib.hc_irregular_wave_params_default(C.byref(p))
        p.num_bodies = self.N if num_bodies is None else int(num_bodies)
        p.simulation_dt, p.simulation_duration, p.ramp_duration = simulation_dt, simulation_duration, ramp_duration
        p.wave_height, p.wave_period = wave_height, wave_period
        p.frequency_min, p.frequency_max, p.nfrequencies = frequency_min, frequency_max, nfrequencies
        p.peak_enhancement_factor, p.is_normalized, p.seed = peak_enhancement_factor, int(is_normalized), int(seed)
        fn = self.lib.hc_set_wave_irregular_spectral if spectral else self.lib.hc_set_wave_irregular
        self._chk(fn(self.ctx, C.byref(p)))

    def set_eta_synthesis(self, mode):
        """0 = direct FP64 sum (default), 1 = rocFFT chirp-z."""
        self._chk(self.lib.hc_set_eta_synthesis(self.ctx, int(mode)))

    def set_convolution_mode(self, mode):
        self._chk(self.lib.hc_set_convolution_mode(self.ctx, int(mode)))

    def set_tapered_direct_options(self, smoothing=0, window_length=5, rirf_end_time=-1.0, taper_start_percent=0.8,
                                   taper_end_percent=1.0, taper_final_amplitude=0.0, export_plot_csv=False):
        o = capi.TaperedDirectOptions(int(smoothing), int(window_length), rirf_end_time, taper_start_percent,
                                      taper_end_percent, taper_final_amplitude, int(export_plot_csv))
        self._chk(self.lib.hc_set_tapered_direct_options(self.ctx, C.byref(o)))

    def set_diagnostics_output_directory(self, directory):
        self._chk(self.lib.hc_set_diagnostics_output_directory(self.ctx, str(directory).encode()))

    # -- per step --
    def step(self, t, pos, rpy, linvel, angvel):
        n3 = 3 * self.N
        a = [x if (type(x) is np.ndarray and x.dtype == np.float64 and x.size == n3 and x.flags.c_contiguous) else _arr(x, n3)
             for x in (pos, rpy, linvel, angvel)]
        out = np.empty(self.D_local)
        # raw addresses through a c_void_p prototype: this call sits in per-step loops
        rc = capi.step_raw(self.lib)(self.ctx, t, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, out.ctypes.data)
        if rc:
            self._chk(rc)
        return out

    def step_many(self, times, states, forces=None, seconds=None):
        """hc_step_many: one synchronous hc_step per row -- times [n], states [n][12N] packed pos | rpy | linvel | angvel
        (mock_chrono.PrescribedMotion.packed), both float64 C-contiguous -- in ONE call of the C ABI (a prescribed-motion driver's loop
        without an interpreter between the steps).  Returns (forces [n][D_local], seconds per call [n]); a failing step raises after
        the rows before it have been filled."""
        times = np.ascontiguousarray(times, dtype=np.float64)
        states = np.ascontiguousarray(states, dtype=np.float64)
        n = times.size
        if states.shape != (n, 12 * self.N):
            raise ValueError(f"states must be [{n}][{12 * self.N}]")
        forces = np.empty((n, self.D_local)) if forces is None else forces
        seconds = np.empty(n) if seconds is None else seconds
        # the C side writes n * D_local and n doubles through these pointers: anything but a writeable C-contiguous float64 array of
        # exactly that shape would be a heap overflow (or a silently ignored result), so refuse it -- also under python -O
        for name, arr, shape in (("forces", forces, (n, self.D_local)), ("seconds", seconds, (n,))):
            if not isinstance(arr, np.ndarray) or arr.dtype != np.float64 or arr.shape != shape or not arr.flags.c_contiguous or not arr.flags.writeable:
                raise ValueError(f"{name} must be a writeable C-contiguous float64 array of shape {shape}")
        done = C.c_int(0)
        rc = self.lib.hc_step_many(self.ctx, n, _dp(times), _dp(states), _dp(forces), _dp(seconds), C.byref(done))
        if rc:
            self._chk(rc)
        return forces, seconds

    def set_result_buffer(self, addr, size):
        """hc_set_result_buffer: the tagged {value, sequence} granules of hc_step go to caller memory at `addr` (e.g. a shared-memory
        segment other processes map, host_exchange.HostExchange); None / 0 hands the buffer back to the library."""
        self._chk(self.lib.hc_set_result_buffer(self.ctx, addr or None, int(size)))

    def step_sequence(self):
        """hc_step_sequence: the sequence number of the step begun last (what its granules are tagged with)."""
        s = C.c_ulonglong()
        self.lib.hc_step_sequence(self.ctx, C.byref(s))
        return s.value

    def step_device(self, t, state_ptr, out_ptr, stream_ptr=None):
        """state_ptr / out_ptr: integer device addresses (e.g. torch.Tensor.data_ptr()).

        stream_ptr: hipStream_t as an integer.  None or 0 is NOT the legacy default stream (whose handle is 0, e.g.
        torch.cuda.current_stream() outside a stream context) but the context's own non-blocking stream, which other
        work is not ordered against: pass an explicit stream (torch.cuda.Stream().cuda_stream) when device work of
        the caller -- a collective, a copy -- has to follow the step."""
        # plain ints go straight through the declared c_void_p argtypes (this call sits in per-step loops)
        rc = self.lib.hc_step_device(self.ctx, t, state_ptr, out_ptr, stream_ptr or None)
        if rc:
            self._chk(rc)

    def components(self):
        hs, rad, wv = (np.empty(self.D_local) for _ in range(3))
        self._chk(self.lib.hc_get_force_components(self.ctx, _dp(hs), _dp(rad), _dp(wv)))
        return hs, rad, wv

    def compute_radiation(self, t, linvel, angvel):
        lv, av = _arr(linvel, 3 * self.N), _arr(angvel, 3 * self.N)
        out = np.empty(self.D_local)
        self._chk(self.lib.hc_compute_radiation(self.ctx, float(t), _dp(lv), _dp(av), _dp(out)))
        return out

    def compute_hydrostatics(self, pos, rpy):
        p, r = _arr(pos, 3 * self.N), _arr(rpy, 3 * self.N)
        out = np.empty(self.D_local)
        self._chk(self.lib.hc_compute_hydrostatics(self.ctx, _dp(p), _dp(r), _dp(out)))
        return out

    def compute_waves(self, t):
        out = np.empty(self.D_local)
        self._chk(self.lib.hc_compute_waves(self.ctx, float(t), _dp(out)))
        return out

    def set_lookahead(self, steps):
        """0 = plain per-step evaluation; > 0 = 16-step look-ahead blocking (the default)."""
        self._chk(self.lib.hc_set_lookahead(self.ctx, int(steps)))

    def set_pass_schedule(self, one_block_ahead, slices=0):
        """hc_set_pass_schedule: 0 = the pass of a look-ahead block when the block starts, 1 = one block ahead, in `slices` launches
        (0: chosen by the library) behind the first steps of the block before -- for callers that leave the GPU idle between force
        evaluations for less than a pass takes; None or -1 = the library's default: ADAPTIVE -- per block, from the gaps the caller
        left between the synchronous steps of the block before (back to back: at block start; away for more than a few
        microseconds: one block ahead; systems below 256 MB of K always at block start; HC_PASS_AHEAD=0/1 pins it)."""
        mode = -1 if one_block_ahead is None or int(one_block_ahead) < 0 else int(bool(one_block_ahead))
        self._chk(self.lib.hc_set_pass_schedule(self.ctx, mode, int(slices)))

    def schedule(self):
        """hc_get_schedule: {"lookahead": 0 | 16 | 32 (what hc_set_lookahead made of its argument), "pass_schedule": -1 adaptive | 0 | 1,
        "ahead_now": the adaptive rule's current answer, "slices"}."""
        v = [C.c_int() for _ in range(4)]
        self._chk(self.lib.hc_get_schedule(self.ctx, *[C.byref(x) for x in v]))
        return dict(zip(("lookahead", "pass_schedule", "ahead_now", "slices"), (x.value for x in v)))

    def rirf_value(self, row_local, col, st):
        """TestHydro::GetRIRFval(row, col, st) for a local row (src/hydro_forces.cpp:693-711)."""
        v = C.c_double()
        self._chk(self.lib.hc_get_rirf_value(self.ctx, int(row_local), int(col), int(st), C.byref(v)))
        return v.value

    def direct_dispatch(self):
        """(active, reason): whether hc_step writes AQL packets itself instead of calling hipLaunchKernelGGL."""
        return bool(self.lib.hc_direct_dispatch_active(self.ctx)), self.lib.hc_dispatch_mode_reason(self.ctx).decode()

    def reset_history(self):
        self._chk(self.lib.hc_reset_history(self.ctx))

    def set_history(self, times_newest_first, vel):
        t = _arr(times_newest_first)
        v = _arr(vel, t.size * self.D)
        self._chk(self.lib.hc_set_history(self.ctx, t.size, _dp(t), _dp(v)))

    def get_history(self):
        n = C.c_int()
        self._chk(self.lib.hc_get_history(self.ctx, C.byref(n), None, None))
        t, v = np.empty(n.value), np.empty((n.value, self.D))
        self._chk(self.lib.hc_get_history(self.ctx, C.byref(n), _dp(t), _dp(v.reshape(-1))))
        return t, v

    # -- added mass (ChLoadAddedMass) --
    def added_mass_matrix(self):
        M = np.empty((self.D_local, self.D))
        self._chk(self.lib.hc_added_mass_matrix(self.ctx, _dp(M.reshape(-1))))
        return M

    def added_mass_mv(self, R, w, c):
        R = _arr(R).copy()
        w = _arr(w)
        self._chk(self.lib.hc_added_mass_mv(self.ctx, _dp(w), float(c), _dp(R), R.size))
        return R

    # -- introspection --
    def sizes(self):
        v = [C.c_int() for _ in range(8)]
        self._chk(self.lib.hc_get_sizes(self.ctx, *[C.byref(x) for x in v]))
        return dict(zip(("N", "n_local", "S", "L", "nf", "nt", "H", "Hcap"), (x.value for x in v)))

    def enable_profiling(self, on=1):
        """on = n > 0: HIP events around the kernels of every n-th step; 0/False: off."""
        self._chk(self.lib.hc_enable_profiling(self.ctx, int(on)))

    def reset_profile(self):
        self._chk(self.lib.hc_reset_profile(self.ctx))

    def profile(self):
        p = capi.ProfileStats()
        self._chk(self.lib.hc_get_profile(self.ctx, C.byref(p)))
        return {k: getattr(p, k) for k, _ in capi.ProfileStats._fields_}

    def init_stats(self):
        """hc_get_init_stats: what the init half of the path cost this context, stage by stage (seconds and bytes)."""
        st = capi.InitStats()
        self._chk(self.lib.hc_get_init_stats(self.ctx, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_ if k != "pad_"}

    def rirf_width(self):
        w = np.empty(self.sizes()["S"])
        self._chk(self.lib.hc_get_rirf_width(self.ctx, _dp(w)))
        return w

    def rirf_effective(self):
        S = self.sizes()["S"]
        out = np.empty((self.D_local, self.D, S))
        self._chk(self.lib.hc_get_rirf_effective(self.ctx, _dp(out.reshape(-1))))
        return out

    def irreg_irf(self, b=0):
        Lb = C.c_int()
        self._chk(self.lib.hc_get_excitation_irf_size(self.ctx, b, C.byref(Lb)))
        L = Lb.value
        t, w, v = np.empty(L), np.empty(L), np.empty((6, L))
        self._chk(self.lib.hc_get_excitation_irf_resampled(self.ctx, b, _dp(t), _dp(w), _dp(v.reshape(-1))))
        return t, w, v

    def irreg_spectrum(self):
        nf = self.sizes()["nf"]
        arrs = [np.empty(nf) for _ in range(5)]
        self._chk(self.lib.hc_get_spectrum(self.ctx, *[_dp(a) for a in arrs]))
        return dict(zip(("f", "S", "df", "phase", "k"), arrs))

    def irreg_eta(self):
        nt = self.sizes()["nt"]
        t, e = np.empty(nt), np.empty(nt)
        self._chk(self.lib.hc_get_eta_table(self.ctx, _dp(t), _dp(e)))
        return t, e

    def export_irregular_inputs_h5(self, path):
        self._chk(self.lib.hc_export_irregular_inputs_h5(self.ctx, str(path).encode()))

    def regular_coeffs(self):
        mag, ph, k = np.empty(self.D), np.empty(self.D), C.c_double()
        self._chk(self.lib.hc_get_regular_coeffs(self.ctx, _dp(mag), _dp(ph), C.byref(k)))
        return mag, ph, k.value


class HydroGroup:
    """G row-sharded contexts of ONE coupled N-body system driven by one host process through hc_step_multi /
    hc_added_mass_mv_multi (drop-in variant: host holds all state -> a state store per GPU -> host gather).
    `shards` are HydroForces objects created with body_range=... that together cover bodies [0, N)."""

    def __init__(self, shards):
        self.shards = list(shards)
        self.N = self.shards[0].N
        self.D = 6 * self.N
        self.lib = self.shards[0].lib
        covered = sorted((h.b0, h.b1) for h in self.shards)
        if covered[0][0] != 0 or covered[-1][1] != self.N or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
            raise ValueError("the shards do not partition bodies [0, N)")
        self._ctxs = (C.c_void_p * len(self.shards))(*[h.ctx for h in self.shards])
        self._step = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(
            ("hc_step_multi", self.lib))

    @classmethod
    def from_case(cls, case, n_shards, devices=None):
        from .parallel_split import body_shard
        devices = devices or [0] * n_shards
        return cls([HydroForces.from_case(case, device=devices[g], body_range=body_shard(case["N"], n_shards, g)) for g in range(n_shards)])

    def __getattr__(self, name):
        # configuration calls (add_waves_*, set_lookahead, set_history, ...) go to every shard
        if name.startswith(("add_waves", "set_", "reset_", "enable_")):
            def fan_out(*a, **k):
                for h in self.shards:
                    getattr(h, name)(*a, **k)
            return fan_out
        raise AttributeError(name)

    def step(self, t, pos, rpy, linvel, angvel):
        n3 = 3 * self.N
        a = [_arr(x, n3) for x in (pos, rpy, linvel, angvel)]
        out = np.empty(self.D)
        rc = self._step(self._ctxs, len(self.shards), t, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, out.ctypes.data)
        if rc:
            raise HydroError(rc, self.lib.hc_last_error(self.shards[0].ctx).decode())
        return out

    def components(self):
        parts = [h.components() for h in sorted(self.shards, key=lambda h: h.b0)]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    def added_mass_mv(self, R, w, c):
        R = _arr(R).copy()
        w = _arr(w)
        rc = self.lib.hc_added_mass_mv_multi(self._ctxs, len(self.shards), _dp(w), float(c), _dp(R), R.size)
        if rc:
            raise HydroError(rc, self.lib.hc_last_error(self.shards[0].ctx).decode())
        return R

    def close(self):
        for h in self.shards:
            h.close()
