"""ctypes binding of the C ABI in include/hydrochrono_amd.h (libhydrochrono_amd.so).

This module only declares signatures and loads the library.  It fails loudly if the HIP library is missing:
the hydro-force path has no CPU fallback.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libhydrochrono_amd.so")
# the same sources built with -DHC_TUNING (hydrochrono_amd/build.py): reads the sweep / A-B / fault-injection switches of
# csrc/hc_internal.hpp (HC_TUNE_INT) and holds the kernel variants that were measured and not taken; loaded by tests and probes only
TUNING_LIB_PATH = os.path.join(_PKG, "lib", "libhydrochrono_amd_tuning.so")

HC_OK, HC_ERR_RUNTIME, HC_ERR_OUT_OF_RANGE, HC_ERR_INVALID, HC_ERR_DEVICE, HC_ERR_UNSUPPORTED = range(6)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class IrregularWaveParams(C.Structure):
    """hc_irregular_wave_params == IrregularWaveParams (include/hydroc/wave_types.h:277-292)."""
    _fields_ = [("num_bodies", C.c_int), ("simulation_dt", C.c_double), ("simulation_duration", C.c_double),
                ("ramp_duration", C.c_double), ("wave_height", C.c_double), ("wave_period", C.c_double),
                ("frequency_min", C.c_double), ("frequency_max", C.c_double), ("nfrequencies", C.c_double),
                ("peak_enhancement_factor", C.c_double), ("is_normalized", C.c_int), ("seed", C.c_int)]


class TaperedDirectOptions(C.Structure):
    """hc_tapered_direct_options == TestHydro::TaperedDirectOptions (include/hydroc/hydro_forces.h:246-259)."""
    _fields_ = [("smoothing", C.c_int), ("window_length", C.c_int), ("rirf_end_time", C.c_double),
                ("taper_start_percent", C.c_double), ("taper_end_percent", C.c_double),
                ("taper_final_amplitude", C.c_double), ("export_plot_csv", C.c_int)]


class ProfileStats(C.Structure):
    _fields_ = [("hydrostatics_seconds", C.c_double), ("radiation_seconds", C.c_double), ("waves_seconds", C.c_double),
                ("hydrostatics_calls", C.c_int), ("radiation_calls", C.c_int), ("waves_calls", C.c_int),
                ("conv_kernel_seconds", C.c_double), ("conv_kernel_launches", C.c_longlong),
                ("conv_kernel_bytes", C.c_double), ("block_kernel_seconds", C.c_double),
                ("block_kernel_launches", C.c_longlong), ("block_kernel_bytes", C.c_double),
                ("block_kernel_bytes_once", C.c_double),
                ("step_kernel_seconds", C.c_double), ("step_kernel_launches", C.c_longlong),
                ("scatter_kernel_seconds", C.c_double), ("scatter_kernel_launches", C.c_longlong),
                ("direct_dispatches", C.c_longlong), ("hip_launches", C.c_longlong), ("history_rewinds", C.c_longlong),
                ("mini_pass_seconds", C.c_double), ("mini_pass_launches", C.c_longlong), ("queue_parkings", C.c_longlong),
                ("ahead_pass_slices", C.c_longlong), ("ahead_blocks", C.c_longlong),
                ("pass_lane_launches", C.c_longlong),
                ("multi_doorbell_offset_last", C.c_double), ("multi_doorbell_offset_sum", C.c_double), ("multi_calls", C.c_longlong),
                ("slot_state_steps", C.c_longlong), ("wide_fused_steps", C.c_longlong),
                ("schedule_blocks_ahead", C.c_longlong), ("schedule_blocks_at_start", C.c_longlong), ("ring_grows_for_pass", C.c_longlong),
                ("hot_steps", C.c_longlong)]


class InitStats(C.Structure):
    _fields_ = [("h5_read_seconds", C.c_double), ("h5_read_bytes", C.c_double), ("rirf_h2d_seconds", C.c_double), ("rirf_h2d_bytes", C.c_double),
                ("rirf_relayout_seconds", C.c_double), ("rirf_relayout_bytes", C.c_double), ("finalize_seconds", C.c_double),
                ("direct_setup_seconds", C.c_double), ("wave_resample_seconds", C.c_double), ("wave_spectrum_seconds", C.c_double),
                ("wave_eta_seconds", C.c_double), ("wave_eta_samples", C.c_longlong), ("wave_eta_components", C.c_longlong),
                ("wave_eta_mode", C.c_int), ("pad_", C.c_int), ("wave_upload_seconds", C.c_double), ("wave_upload_bytes", C.c_double),
                ("wave_total_seconds", C.c_double), ("taper_seconds", C.c_double), ("taper_bytes", C.c_double),
                ("synth_seconds", C.c_double), ("synth_bytes", C.c_double)]


# name -> (restype, argtypes); every symbol include/hydrochrono_amd.h declares
SIGNATURES = {
    "hc_version": (C.c_char_p, []),
    "hc_device_count": (C.c_int, []),
    "hc_device_local_cpus": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "hc_bind_thread_to_device": (C.c_int, [C.c_int]),
    "hc_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "hc_create_sharded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "hc_destroy": (None, [C.c_void_p]),
    "hc_last_error": (C.c_char_p, [C.c_void_p]),
    "hc_set_simulation_parameters": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "hc_set_body_properties": (C.c_int, [C.c_void_p, C.c_int, C.c_double, c_double_p, c_double_p]),
    "hc_set_hydrostatic_stiffness": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "hc_set_added_mass_inf": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "hc_set_rirf": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int, c_double_p]),
    "hc_set_excitation_rao": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p]),
    "hc_set_excitation_irf": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int, c_double_p]),
    "hc_load_bemio_h5": (C.c_int, [C.c_void_p, C.c_char_p]),
    "hc_finalize": (C.c_int, [C.c_void_p]),
    "hc_set_gravity": (C.c_int, [C.c_void_p, c_double_p]),
    "hc_set_wave_none": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_set_wave_regular": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double]),
    "hc_irregular_wave_params_default": (None, [C.POINTER(IrregularWaveParams)]),
    "hc_set_wave_irregular": (C.c_int, [C.c_void_p, C.POINTER(IrregularWaveParams)]),
    "hc_set_eta_synthesis": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_set_wave_irregular_spectral": (C.c_int, [C.c_void_p, C.POINTER(IrregularWaveParams)]),
    "hc_set_convolution_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_tapered_direct_options_default": (None, [C.POINTER(TaperedDirectOptions)]),
    "hc_set_tapered_direct_options": (C.c_int, [C.c_void_p, C.POINTER(TaperedDirectOptions)]),
    "hc_set_diagnostics_output_directory": (C.c_int, [C.c_void_p, C.c_char_p]),
    "hc_get_shard": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "hc_get_excitation_irf_size": (C.c_int, [C.c_void_p, C.c_int, c_int_p]),
    "hc_step": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "hc_step_begin": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p]),
    "hc_step_end": (C.c_int, [C.c_void_p, c_double_p]),
    "hc_step_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "hc_added_mass_mv_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_double_p, C.c_double, c_double_p, C.c_int]),
    "hc_set_result_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hc_step_sequence": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "hc_wait_result_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_ulonglong, c_double_p, C.c_double]),
    "hc_step_device": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_get_force_components": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "hc_compute_radiation": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p, c_double_p]),
    "hc_compute_hydrostatics": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "hc_compute_waves": (C.c_int, [C.c_void_p, C.c_double, c_double_p]),
    "hc_set_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_set_pass_schedule": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hc_get_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hc_direct_dispatch_active": (C.c_int, [C.c_void_p]),
    "hc_dispatch_mode_reason": (C.c_char_p, [C.c_void_p]),
    "hc_reset_history": (C.c_int, [C.c_void_p]),
    "hc_set_history": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    "hc_get_history": (C.c_int, [C.c_void_p, c_int_p, c_double_p, c_double_p]),
    "hc_added_mass_matrix": (C.c_int, [C.c_void_p, c_double_p]),
    "hc_added_mass_mv": (C.c_int, [C.c_void_p, c_double_p, C.c_double, c_double_p, C.c_int]),
    "hc_enable_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_get_profile": (C.c_int, [C.c_void_p, C.POINTER(ProfileStats)]),
    "hc_get_init_stats": (C.c_int, [C.c_void_p, C.POINTER(InitStats)]),
    "hc_reset_profile": (C.c_int, [C.c_void_p]),
    "hc_get_sizes": (C.c_int, [C.c_void_p] + [c_int_p] * 8),
    "hc_get_rirf_width": (C.c_int, [C.c_void_p, c_double_p]),
    "hc_get_rirf_effective": (C.c_int, [C.c_void_p, c_double_p]),
    "hc_get_rirf_value": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p]),
    "hc_step_many": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p]),
    "hc_h5_read": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hc_h5_free": (None, [C.c_void_p]),
    "hc_h5_get_sizes": (C.c_int, [C.c_void_p, c_int_p, c_double_p, c_double_p, c_double_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "hc_h5_get_body": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "hc_h5_get_rirf": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "hc_h5_get_excitation_rao": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "hc_h5_get_excitation_irf": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    "hc_get_excitation_irf_resampled": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "hc_get_spectrum": (C.c_int, [C.c_void_p] + [c_double_p] * 5),
    "hc_get_eta_table": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "hc_export_irregular_inputs_h5": (C.c_int, [C.c_void_p, C.c_char_p]),
    "hc_get_regular_coeffs": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "hc_synth_fill": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_int, C.c_double, C.c_int, C.c_double]),
}

# include/hydrochrono_amd_host.h
SIGNATURES.update({
    "hc_host_linspaced": (None, [C.c_int, C.c_double, C.c_double, c_double_p]),
    "hc_host_trapezoid_widths": (None, [c_double_p, C.c_int, c_double_p]),
    "hc_host_jonswap_spectrum_hz": (None, [c_double_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p]),
    "hc_host_random_phases": (None, [C.c_int, C.c_int, c_double_p]),
    "hc_host_wave_number": (C.c_double, [C.c_double, C.c_double, C.c_double]),
    "hc_host_resample_irf": (C.c_int, [c_double_p, C.c_int, C.c_int, c_double_p]),
})

# include/hydrochrono_amd_yaml.h
SIGNATURES.update({
    "hc_yaml_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "hc_yaml_free": (None, [C.c_void_p]),
    "hc_yaml_num_bodies": (C.c_int, [C.c_void_p]),
    "hc_yaml_body_string": (C.c_char_p, [C.c_void_p, C.c_int, C.c_char_p]),
    "hc_yaml_body_number": (C.c_double, [C.c_void_p, C.c_int, C.c_char_p]),
    "hc_yaml_string": (C.c_char_p, [C.c_void_p, C.c_char_p]),
    "hc_yaml_number": (C.c_double, [C.c_void_p, C.c_char_p]),
    "hc_yaml_period_values": (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
    "hc_create_from_hydro_yaml": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                            C.POINTER(C.c_void_p), c_int_p, c_int_p, C.c_char_p, C.c_size_t]),
    "hc_create_from_hydro_yaml_sharded": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_double, C.c_double, C.c_double, c_int_p, C.c_int,
                                                    C.POINTER(C.c_void_p), c_int_p, c_int_p, C.c_char_p, C.c_size_t]),
})

_libs = {}        # flavour -> loaded library
_step_raws = {}   # flavour -> hc_step bound with integer arguments
_flavor = "tuning" if os.environ.get("HYDROCHRONO_AMD_FLAVOR", "release") == "tuning" else "release"  # (subprocesses of tests / sweeps)


def flavor():
    return _flavor


class use_flavor:
    """`with capi.use_flavor("tuning"):` -- objects created inside (HydroForces, ...) bind to libhydrochrono_amd_tuning.so and keep it
    for their lifetime; the release library stays the default outside.  Both may be loaded in one process: they share nothing but
    the HIP runtime (each is linked -Bsymbolic and loaded RTLD_LOCAL)."""

    def __init__(self, name):
        if name not in ("release", "tuning"):
            raise ValueError(name)
        self.name = name

    def __enter__(self):
        global _flavor
        self.prev, _flavor = _flavor, self.name
        return load()

    def __exit__(self, *exc):
        global _flavor
        _flavor = self.prev
        return False


def step_raw(lib=None):
    """hc_step bound with integer (address) arguments: skips the per-call ctypes pointer conversions."""
    lib = lib or load()
    fn = _step_raws.get(id(lib))
    if fn is None:
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
        fn = _step_raws[id(lib)] = proto(("hc_step", lib))
    return fn


def load(which=None):
    """Load libhydrochrono_amd.so (or, inside use_flavor("tuning") / with which="tuning", its tuning build); raises if it has not
    been built (no fallback)."""
    which = which or _flavor
    if which in _libs:
        return _libs[which]
    path = LIB_PATH if which == "release" else TUNING_LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first (python -m hydrochrono_amd.build, or "
            "__graft_entry__.build()).  The hydro-force path has no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _libs[which] = lib
    return lib
