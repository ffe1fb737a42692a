"""A stand-in for hydrochrono_amd.hydro.HydroForces WITHOUT a GPU -- test infrastructure only, loaded by `bench.py --stub-context`
(tests/test_parallel_gloo.py::test_bench_multi_rank_control_flow_world8_stub): it lets the N > 1 control flow of the bench --
rendezvous, body-row shards, shared-memory result buffers and their tagged granules, the host gather, the device-path all-gather,
the after-run exchange check, the per-rank diagnostics, the JSON line -- run on CPU over gloo with world size 8, so that the first
run on a real 8-GPU node cannot fail on an argument or a buffer shape.  It computes NO physics: a row's "force" is a fixed function
of (time, state, row), the same on every rank, so that gathered vectors can be checked bit for bit."""
import ctypes as C

import numpy as np

from hydrochrono_amd import capi


def fake_forces(t, state, rows):
    """rows: global row indices; state: the packed 12N doubles of the step."""
    s = float(np.sum(state * np.cos(np.arange(state.size) * 0.37)))
    return np.sin(0.1 * rows + t) * (1.0 + 0.001 * s) + 1e-3 * rows


class StubShard:
    def __init__(self, num_bodies, device=0, body_range=None):
        self.N = int(num_bodies)
        self.b0, self.b1 = body_range if body_range else (0, self.N)
        self.D, self.D_local = 6 * self.N, 6 * (self.b1 - self.b0)
        self.rows = np.arange(6 * self.b0, 6 * self.b1, dtype=np.float64)
        self.ctx = id(self) & 0x7FFFFFFF  # an opaque, non-zero "handle"
        self.lib = capi.load()            # host-only entry points (hc_wait_result_buffer) work without a device
        self.seq = 0
        self.buf_addr, self.buf_size = 0, 0
        self._own = np.zeros(2 * 2 * self.D_local, dtype=np.uint64)  # tagged granules when no caller buffer is set
        self._last = np.zeros(self.D_local)
        self.lookahead = 32
        self.calls = {"begin": 0, "end": 0, "device": 0}

    # ---- configuration: accepted and ignored ----
    def synth_fill(self, *a, **k): pass
    def finalize(self): pass
    def add_waves_irregular(self, **k): pass
    def add_waves_none(self, *a): pass
    def set_history(self, t_hist, v_hist):
        assert np.asarray(v_hist).shape == (len(t_hist), self.D), "history must be [n][6N]"
    def set_lookahead(self, steps): self.lookahead = int(steps)
    def set_pass_schedule(self, *a): pass
    def enable_profiling(self, on=1): pass
    def reset_profile(self): pass
    def close(self): pass

    def _chk(self, rc):
        if rc:
            raise RuntimeError(f"stub context: status {rc}")

    def schedule(self):
        return {"lookahead": self.lookahead, "pass_schedule": -1, "ahead_now": 0, "slices": 4}

    def step(self, t, pos, rpy, linvel, angvel):
        state = np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in (pos, rpy, linvel, angvel)])
        return fake_forces(t, state, self.rows)

    def sizes(self):
        return {"N": self.N, "n_local": self.b1 - self.b0, "S": 1024, "L": 1023, "nf": 512, "nt": 0, "H": 0, "Hcap": 0}

    def profile(self):
        p = {name: 0 for name, _ in capi.ProfileStats._fields_}
        p["conv_kernel_bytes"] = p["block_kernel_bytes"] = p["block_kernel_bytes_once"] = 8.0 * self.D_local * self.D * 1024
        return p

    def direct_dispatch(self):
        return False, "stub context (no GPU)"

    # ---- the result buffer of hc_step (hc_set_result_buffer / hc_step_sequence) ----
    def set_result_buffer(self, addr, size):
        if addr:
            assert size >= 4 * self.D_local * 8, "result buffer too small: 2 x 16 bytes per owned row"
            assert addr % 16 == 0
        self.buf_addr, self.buf_size = int(addr or 0), int(size)

    def step_sequence(self):
        return self.seq

    def _publish(self, values):
        seq = self.seq
        g = np.empty(2 * self.D_local, dtype=np.uint64)
        g[0::2] = values.view(np.uint64)
        g[1::2] = seq
        half = (seq & 1) * 2 * self.D_local * 8
        if self.buf_addr:
            C.memmove(self.buf_addr + half, g.ctypes.data, g.nbytes)
        else:
            self._own[(seq & 1) * 2 * self.D_local:(seq & 1) * 2 * self.D_local + 2 * self.D_local] = g

    # ---- hc_step_begin / hc_step_end with the raw-address signature bench.py uses ----
    def begin_raw(self, ctx, t, p_pos, p_rpy, p_lin, p_ang):
        assert ctx == self.ctx
        n3 = 3 * self.N
        assert p_rpy - p_pos == 8 * n3 and p_lin - p_rpy == 8 * n3 and p_ang - p_lin == 8 * n3, "state must be packed pos | rpy | linvel | angvel"
        state = np.ctypeslib.as_array((C.c_double * (4 * n3)).from_address(p_pos)).copy()
        self.seq += 1
        self._last = fake_forces(t, state, self.rows)
        self._publish(self._last)
        self.calls["begin"] += 1
        return 0

    def end_raw(self, ctx, out_addr):
        assert ctx == self.ctx
        C.memmove(out_addr, self._last.ctypes.data, self._last.nbytes)
        self.calls["end"] += 1
        return 0

    def step_device(self, t, state_ptr, out_ptr, stream_ptr=None):
        state = np.ctypeslib.as_array((C.c_double * (12 * self.N)).from_address(state_ptr)).copy()
        f = fake_forces(t, state, self.rows)
        C.memmove(out_ptr, f.ctypes.data, f.nbytes)
        self.calls["device"] += 1
