/* hydrochrono_amd.h -- C ABI of the MI355X-native HydroChrono hydro-force path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry point names
 * the reference interface (file:line relative to the HydroChrono tree @2025-10-31) it replaces.
 * INTEGRATION.md shows the Chrono-side binding (ChFunction / ChLoadCustomMultiple subclasses) that
 * forwards to these calls; the headers under include/hydroc_amd are that binding.
 *
 * Conventions
 *   N   = number of hydro bodies of the whole system, D = 6N degrees of freedom.
 *   A context may own only the output rows of bodies [body_begin, body_end) (multi-GPU row sharding,
 *   SURVEY 8e); n_local = body_end - body_begin, D_local = 6*n_local.  Per-step INPUTS are always
 *   the full N-body state, per-step OUTPUTS have D_local entries.
 *   All arrays are IEEE double, row-major ("file order" of the BEMIO HDF5 datasets) unless stated.
 *   Every function returns an hc_status; hc_last_error() holds the message of the last failure.
 *   One host thread drives a context (same contract as TestHydro: its per-time cache is
 *   unsynchronised, src/hydro_forces.cpp:742-748).
 *   The library computes on the GPU only.  There is no CPU fallback: hc_create fails with
 *   HC_ERR_DEVICE when no gfx950 device is usable.
 */
#ifndef HYDROCHRONO_AMD_H
#define HYDROCHRONO_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hc_ctx hc_ctx;

typedef enum hc_status {
    HC_OK               = 0,
    HC_ERR_RUNTIME      = 1, /* the reference throws std::runtime_error here */
    HC_ERR_OUT_OF_RANGE = 2, /* the reference throws std::out_of_range here  */
    HC_ERR_INVALID      = 3, /* bad argument / call order                     */
    HC_ERR_DEVICE       = 4, /* HIP runtime failure or no usable GPU         */
    HC_ERR_UNSUPPORTED  = 5  /* optional component not built (e.g. HDF5)      */
} hc_status;

/* ------------------------------------------------------------------------------------------------
 * Lifecycle.  Replaces TestHydro::TestHydro / ~TestHydro (src/hydro_forces.cpp:170-242).
 * ---------------------------------------------------------------------------------------------- */
const char* hc_version(void);
int hc_device_count(void); /* number of visible HIP devices; does not initialise a context */
/* Where the host thread that steps should run.  A synchronous step is a round trip between one host thread and one GPU: a doorbell
 * store into the GPU's MMIO page, stores through its BAR, and a spin on pinned memory the GPU writes.  From a core of the OTHER
 * socket every one of them crosses the socket interconnect: 64 bodies 11.5-11.8 us per hc_step against 10.0-10.3 us from a core of the
 * GPU's own NUMA node (profiles/r06/affinity_ahead_probe.txt) -- and an unpinned thread lands on either.
 * hc_device_local_cpus: the CPUs local to the device's PCIe root as the kernel lists them ("0-63,128-191"; "" when unknown).
 * hc_bind_thread_to_device: restricts the CALLING thread to those CPUs (sched_setaffinity; threads it creates afterwards inherit the
 * mask -- a host that wants its worker pool on all sockets creates it first, or binds only its stepping thread).  HC_ERR_UNSUPPORTED
 * when the CPUs are not known.  The worker threads of hc_step_multi bind themselves to their context's device (HC_MULTI_PIN=0: no). */
int hc_device_local_cpus(int device_id, char* out, size_t out_bytes);
int hc_bind_thread_to_device(int device_id);
int hc_create(int num_bodies, int device_id, hc_ctx** out);
/* Row-sharded context: owns output rows of bodies [body_begin, body_end) only. */
int hc_create_sharded(int num_bodies, int body_begin, int body_end, int device_id, hc_ctx** out);
void hc_destroy(hc_ctx* ctx);
/* The bodies [body_begin, body_end) whose output rows the context owns (either pointer may be NULL). */
int hc_get_shard(hc_ctx* ctx, int* body_begin, int* body_end);
const char* hc_last_error(const hc_ctx* ctx); /* ctx may be NULL: error of the last failed hc_create */

/* ------------------------------------------------------------------------------------------------
 * Ingest.  Replaces H5FileInfo::ReadH5Data (src/h5fileinfo.cpp:27-180); the raw-array setters take
 * exactly the datasets it reads, UNSCALED (the rho / rho*g scaling of :60-61,73-75,89-90 is applied
 * inside), so tests need no HDF5.
 * ---------------------------------------------------------------------------------------------- */
/* simulation_parameters/{rho,g,water_depth}  (src/h5fileinfo.cpp:35-37; "infinite" -> +inf, :207-220) */
int hc_set_simulation_parameters(hc_ctx* ctx, double rho, double g, double water_depth);
/* bodyK/properties/{disp_vol,cg,cb}  (:48,56-57) */
int hc_set_body_properties(hc_ctx* ctx, int body, double disp_vol, const double cg[3], const double cb[3]);
/* bodyK/hydro_coeffs/linear_restoring_stiffness {6,6}  (:58-59) */
int hc_set_hydrostatic_stiffness(hc_ctx* ctx, int body, const double lin[36]);
/* bodyK/hydro_coeffs/added_mass/inf_freq {6,D}  (:60-61) */
int hc_set_added_mass_inf(hc_ctx* ctx, int body, const double* A_6xD);
/* bodyK/hydro_coeffs/radiation_damping/impulse_response_fun/{t,K}; K is {6,D,S} (:49-50,62-63).
 * All bodies must carry the same t within 1e-10 (HydroData::GetRIRFTimeVector, :329-343). */
int hc_set_rirf(hc_ctx* ctx, int body, const double* t, int S, const double* K_6xDxS);
/* simulation_parameters/w and bodyK/hydro_coeffs/excitation/{mag,phase} {6,1,nw}  (:68-78) */
int hc_set_excitation_rao(hc_ctx* ctx, int body, const double* w, int nw, const double* mag_6x1xnw,
                          const double* phase_6x1xnw);
/* bodyK/hydro_coeffs/excitation/impulse_response_fun/{t,f}; f is {6,1,n}  (:83-90) */
int hc_set_excitation_irf(hc_ctx* ctx, int body, const double* t, int n, const double* f_6x1xn);
/* Reads all of the above for bodies "body1".."bodyN" from a BEMIO HDF5 file (needs libhdf5 at build
 * time, else HC_ERR_UNSUPPORTED). */
int hc_load_bemio_h5(hc_ctx* ctx, const char* path);
/* The same file WITHOUT a device context: H5FileInfo(path, num_bodies).ReadH5Data() (include/hydroc/h5fileinfo.h:230-260,
 * src/h5fileinfo.cpp:27-153) for callers that only want to look at the data (the reference's tests/h5fileinfo_t01.cpp,
 * chloadaddedmass_t01.cpp; the C++ face is include/hydroc_amd/h5fileinfo.h).  Host side only, no GPU is touched.  Values come back as the
 * reference's HydroData holds them: added mass x rho (:60-61), excitation magnitude x rho g (:73-75), excitation IRF x rho g (:89-90);
 * K, the stiffness matrix and everything else as in the file.  Errors: status code + hc_last_error(NULL). */
typedef struct hc_h5data hc_h5data;
int hc_h5_read(const char* path, int num_bodies, hc_h5data** out);
void hc_h5_free(hc_h5data* data);
/* S radiation samples, nw RAO frequencies, L excitation-IRF samples of (0-based) body `body`; any pointer may be NULL */
int hc_h5_get_sizes(const hc_h5data* data, int* num_bodies, double* rho, double* g, double* water_depth, int body, int* S, int* nw, int* L);
/* HydroData::BodyInfo (include/hydroc/h5fileinfo.h:37-47); ainf is {6, 6N} row-major, rirf_t has S entries; any pointer may be NULL */
int hc_h5_get_body(const hc_h5data* data, int body, double* disp_vol, double cg[3], double cb[3], double lin[36], double* ainf_6xD, double* rirf_t_S);
/* bodyK/.../impulse_response_fun/K in file order {6, 6N, S} (HydroData::GetRIRFVal(b, dof, col, s), src/h5fileinfo.cpp:321-323) */
int hc_h5_get_rirf(const hc_h5data* data, int body, double* K_6xDxS);
/* HydroData::RegularWaveInfo (:56-60): freq_list {nw}, magnitude {6, nw} x rho g, phase {6, nw} */
int hc_h5_get_excitation_rao(const hc_h5data* data, int body, double* w_nw, double* mag_6xnw, double* phase_6xnw);
/* HydroData::IrregularWaveInfo (:61-72): excitation_irf_time {L}, excitation_irf_matrix {6, L} x rho g */
int hc_h5_get_excitation_irf(const hc_h5data* data, int body, double* t_L, double* f_6xL);
/* End of ingest = rest of the TestHydro constructor (src/hydro_forces.cpp:176-238): trapezoid widths,
 * equilibrium, cb-cg, K re-laid-out into HBM, added-mass assembly (src/chloadaddedmass.cpp:12-25),
 * default NoWave. */
int hc_finalize(hc_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Configuration.
 * ---------------------------------------------------------------------------------------------- */
/* ChSystem::GetGravitationalAcceleration(), read by ComputeForceHydrostatics (src/hydro_forces.cpp:268).
 * Default (0,0,-9.81). */
int hc_set_gravity(hc_ctx* ctx, const double g[3]);
/* TestHydro::AddWaves(std::make_shared<NoWave>(num_bodies_arg))  (src/hydro_forces.cpp:244-261,
 * src/wave_types.cpp:257-264).  num_bodies_arg < N reproduces the reference's short force vector as an
 * error (HC_ERR_RUNTIME at the first step) instead of an out-of-bounds read. */
int hc_set_wave_none(hc_ctx* ctx, int num_bodies_arg);
/* AddWaves(RegularWave(num_bodies_arg)) with regular_wave_amplitude_/regular_wave_omega_
 * (src/wave_types.cpp:274-352). */
int hc_set_wave_regular(hc_ctx* ctx, int num_bodies_arg, double amplitude, double omega);

/* IrregularWaveParams (include/hydroc/wave_types.h:277-292); eta_file_path_ is not supported (that
 * reference branch is undefined behaviour), wave_stretching_ does not affect forces. */
typedef struct hc_irregular_wave_params {
    int num_bodies;
    double simulation_dt;
    double simulation_duration;
    double ramp_duration;
    double wave_height;
    double wave_period;
    double frequency_min;           /* default 0.001 */
    double frequency_max;           /* default 1.0   */
    double nfrequencies;            /* 0 = ceil((fmax-fmin)*duration) */
    double peak_enhancement_factor; /* 1.0 = Pierson-Moskowitz */
    int is_normalized;
    int seed;                       /* default 1 */
} hc_irregular_wave_params;
void hc_irregular_wave_params_default(hc_irregular_wave_params* p);
/* AddWaves(IrregularWaves(params)): excitation-IRF resampling, spectrum, phases, eta(t) table
 * (src/wave_types.cpp:432-459,572-606,643-676,717-774). */
int hc_set_wave_irregular(hc_ctx* ctx, const hc_irregular_wave_params* params);

/* How hc_set_wave_irregular builds the free-surface table eta(t) (GetEtaIrregularTimeSeries, src/wave_types.cpp:27-59):
 * 0 = direct FP64 sum of the Nf cosines per time sample on the GPU (default; same summation order as the reference),
 * 1 = chirp-z transform on rocFFT (three FFTs of length >= nt + nf - 1; agrees with the direct sum to ~1e-12 of max|eta|). */
int hc_set_eta_synthesis(hc_ctx* ctx, int mode);

/* Spectral (component-sum) excitation -- NOT in the reference (whose irregular waves are the excitation-IRF convolution
 * above); it is the mode BASELINE.json's north_star words literally: the same spectrum / phases as hc_set_wave_irregular,
 *   f[row](t) = ramp(t) * sum_i |X_row(w_i)| * a_i * cos(w_i t - phi_i + arg X_row(w_i)),   a_i = sqrt(2 S_i df_i),
 * with the excitation RAO interpolated per component by RegularWave's interpolator (src/wave_types.cpp:329-352, held
 * constant outside the BEM frequency range) and per-body phases.  Agrees with the IRF convolution up to the IRF's
 * truncation / resampling error (a few per cent on the sphere data), so it is validated at a looser tolerance. */
int hc_set_wave_irregular_spectral(hc_ctx* ctx, const hc_irregular_wave_params* params);

/* TestHydro::SetRadiationConvolutionMode: 0 = Baseline, 1 = TaperedDirect (include/hydroc/hydro_forces.h:234-243) */
int hc_set_convolution_mode(hc_ctx* ctx, int mode);
/* TestHydro::TaperedDirectOptions (include/hydroc/hydro_forces.h:246-259) */
typedef struct hc_tapered_direct_options {
    int smoothing;                /* 0 = "sg" (Savitzky-Golay 5), 1 = "moving_average" */
    int window_length;            /* moving average only, >= 3 */
    double rirf_end_time;         /* <= 0: full length */
    double taper_start_percent;   /* 0.8 */
    double taper_end_percent;     /* 1.0 */
    double taper_final_amplitude; /* 0.0 */
    int export_plot_csv;          /* 0; 1: write rirf_body<b>_summary.csv (b 0-based, one per owned body; columns
                                     step,time,k_before,k_after of channel row 0 / column 0, src/hydro_forces.cpp:509-531) into the
                                     diagnostics directory when the kernel is processed */
} hc_tapered_direct_options;
void hc_tapered_direct_options_default(hc_tapered_direct_options* o);
int hc_set_tapered_direct_options(hc_ctx* ctx, const hc_tapered_direct_options* opts);
/* TestHydro::SetDiagnosticsOutputDirectory (include/hydroc/hydro_forces.h:269): where the export_plot_csv files go; "" (the
 * default) = the current working directory, as in the reference (src/hydro_forces.cpp:513).  Export errors are ignored (:529). */
int hc_set_diagnostics_output_directory(hc_ctx* ctx, const char* dir);

/* ------------------------------------------------------------------------------------------------
 * Per-step force evaluation.  Replaces the 6N ComponentFunc::GetVal -> ForceFunc6d::CoordinateFunc ->
 * TestHydro::CoordinateFuncForBody callbacks of one Chrono update (src/hydro_forces.cpp:79-85,136-144,
 * 727-767): total = hydrostatic - radiation + waves, evaluated once per distinct time `t` and cached.
 *   pos     [N][3]  ChBody::GetPos()
 *   rpy     [N][3]  ChBody::GetRot().GetCardanAnglesXYZ()
 *   linvel  [N][3]  ChBody::GetPosDt()
 *   angvel  [N][3]  ChBody::GetAngVelParent()
 *   force_out [D_local] world-frame force (x,y,z) and torque (x,y,z) per owned body.
 * Errors kept from the reference: excitation window exceeded (src/wave_types.cpp:833-840) -> HC_ERR_RUNTIME.
 * A step BACK in time (t below the newest history sample: an integrator that rejected a step and retries from an earlier time)
 * drops the history samples at times >= t -- they belong to the abandoned attempt -- and continues from the history as it was
 * at t.  Deviation: the reference has no such rule; it inserts the earlier time in front of its newest-first history
 * (src/hydro_forces.cpp:559-574), keeps the abandoned samples and interpolates in a non-monotone list from then on.
 * hc_step is synchronous (the forces are in force_out when it returns) but does not synchronise the stream: work that later
 * steps need may still be running on the context's stream.
 * ---------------------------------------------------------------------------------------------- */
int hc_step(hc_ctx* ctx, double t, const double* pos, const double* rpy, const double* linvel, const double* angvel,
            double* force_out);
/* The two halves of hc_step for a host that has other work between handing the state to the GPU and needing the forces:
 * hc_step_begin stores the state, hands the step kernel to the GPU (and enqueues what later steps need) and returns;
 * hc_step_end waits for the results of that step.  Exactly one hc_step_end per hc_step_begin; no other per-step call on
 * the context in between.  Same cache and error rules as hc_step (a failure of either half leaves nothing pending). */
int hc_step_begin(hc_ctx* ctx, double t, const double* pos, const double* rpy, const double* linvel, const double* angvel);
int hc_step_end(hc_ctx* ctx, double* force_out);
/* A prescribed-motion driver's loop ("the build's own counterpart is a mock Chrono loop"): n synchronous evaluations one
 * after the other -- hc_step(ctx, t_n[k], state k, force k) for k = 0 .. n-1 with state k the packed block
 * [pos 3N | rpy 3N | linvel 3N | angvel 3N] at states_nx12N + 12N*k and force k the D_local totals at forces_nxDlocal + D_local*k.
 * seconds_n (may be NULL) receives the wall time of every call (std::chrono::steady_clock around hc_step).  Stops at the first failing
 * step and returns its status; *done (may be NULL) is the number of steps completed.  Same results as n calls of hc_step: it is n calls. */
int hc_step_many(hc_ctx* ctx, int n, const double* t_n, const double* states_nx12N, double* forces_nxDlocal, double* seconds_n, int* done);
/* Multi-GPU inside ONE host process -- the reference is one C++ object in one Chrono process (src/hydro_forces.cpp:170-242),
 * evaluated by one call per time (:727-767); SURVEY 8e, drop-in variant.  ctxs[0..n_ctx) are row-sharded contexts of the same
 * N-body system (hc_create_sharded, any devices, any split of the bodies); the call stores the state into every context and
 * rings every GPU's doorbell BEFORE it enqueues anything else or waits, then gathers each shard's rows:
 *   force_out[6*body_begin(g) .. 6*body_end(g)) = the totals of context g          (force_out has 6N entries)
 * No collective, no device-to-device traffic: inputs are 12N doubles per GPU, outputs 6*n_local doubles per GPU, through the
 * PCIe BAR / mapped host memory like hc_step.  The gathered vector is bitwise the one an unsharded context returns.  On
 * failure the status of the first failing context is returned, hc_last_error() of every context of the group holds its
 * message, and no context is left with a step pending. */
int hc_step_multi(hc_ctx* const* ctxs, int n_ctx, double t, const double* pos, const double* rpy, const double* linvel,
                  const double* angvel, double* force_out);
/* One process per GPU (an MPI-style host, or this repo's benchmark under torch.distributed.run): the host gather without a collective.
 * hc_set_result_buffer makes the step kernel deliver this context's tagged results -- 16-byte {value, step sequence number} granules,
 * 2 x D_local of them, the two halves used by consecutive steps in turn -- into memory the caller provides, e.g. a POSIX
 * shared-memory segment every process of the node maps (the library registers it with the GPU; NULL returns to the internal buffer;
 * the memory must stay mapped until then or until hc_destroy).  `rows` of hc_wait_result_buffer = D_local of the shard that writes the buffer.
 * After hc_step_begin each process collects every shard's rows with hc_wait_result_buffer (host-only: it spins on the granules of the
 * given sequence number -- hc_step_sequence of the own context; contexts driven in lockstep count alike -- and copies the values
 * out; HC_ERR_DEVICE after timeout_seconds, <= 0: HC_STEP_TIMEOUT_S / 20 s), then completes its own step with hc_step_end. */
int hc_set_result_buffer(hc_ctx* ctx, void* host_buffer, size_t bytes);
int hc_step_sequence(const hc_ctx* ctx, unsigned long long* seq);
int hc_wait_result_buffer(const void* host_buffer, int rows, unsigned long long seq, double* values_out, double timeout_seconds);
/* Same evaluation with the body state already in HBM and the result left in HBM:
 *   d_state     device pointer, 12N doubles = pos[3N] | rpy[3N] | linvel[3N] | angvel[3N]
 *   d_force_out device pointer, D_local doubles
 *   stream      hipStream_t (NULL = the context's own stream).  Asynchronous: returns after enqueue; d_state and
 *               d_force_out must stay valid until the work enqueued for this step has run.  The steps of one context
 *               normally stay on one stream (the velocity ring is updated in stream order); a step that goes to another
 *               stream than the one before it -- hc_step included, which uses the context's stream -- is ordered behind
 *               it with an event.  hc_step and hc_step_device share the per-time cache.  On a caller's stream that is
 *               idle when the call arrives (a caller that waits for every step) only the step kernel is enqueued there;
 *               the work later steps need (scatter, look-ahead pass) runs on the context's own stream behind an event,
 *               and the next step waits for it -- what the caller enqueues next on its stream (e.g. the all-gather of the
 *               force rows of a row-sharded array) follows the step kernel directly.  A caller that enqueues steps ahead
 *               of the GPU gets all of it on its stream, in order. */
int hc_step_device(hc_ctx* ctx, double t, const double* d_state, double* d_force_out, void* stream);
/* force_hydrostatic_, force_radiation_damping_, force_waves_ of the last evaluated step (D_local each;
 * any pointer may be NULL).  Synchronises the context's stream. */
int hc_get_force_components(hc_ctx* ctx, double* hydrostatic, double* radiation, double* waves);
/* TestHydro::ComputeForceRadiationDampingConv called directly (src/hydro_forces.cpp:537-691): pushes
 * (t, velocities) into the history and returns the radiation term only.  Calling it twice with the
 * same t is the reference's duplicate-time error (:555-557) -> HC_ERR_RUNTIME. */
int hc_compute_radiation(hc_ctx* ctx, double t, const double* linvel, const double* angvel, double* rad_out);
/* TestHydro::ComputeForceHydrostatics (:263-322) and ComputeForceWaves (:713-725) on their own. */
int hc_compute_hydrostatics(hc_ctx* ctx, const double* pos, const double* rpy, double* hs_out);
int hc_compute_waves(hc_ctx* ctx, double t, double* waves_out);
/* Multi-step look-ahead (on by default, 32 steps): when the caller steps on a uniform time grid, one blocked pass over K
 * precomputes, for the next 32 (or 16) predicted step times, what the history known so far contributes to the radiation sum,
 * so K leaves HBM once per block; a step inside a block is ONE kernel launch that adds its own newest-sample part, and what
 * later steps of the block need from it is enqueued behind it (off the caller's critical path).  A step whose time deviates
 * from the prediction (> max(1e-9 of the step size, 64 ulp)) falls back to the plain per-step evaluation: a missed prediction
 * costs speed, never accuracy.  A step whose time is accepted is evaluated on the PREDICTED time grid (t0 + m*dt), so its
 * interpolation weights differ from those of the caller's t by at most that tolerance / dt relative (1e-15 .. 1e-12 observed;
 * contract 1e-6).  Wide systems (6N >= 1024) use a two-level form: sub-blocks of 8 steps with a short pass over the head of K
 * after each.  steps = 0 disables it (every step streams K), 1..16 selects blocks of 16, more blocks of 32 (a depth-64 pass exists in
 * the tuning build only: measured and not taken, EXPERIMENTS.md). */
int hc_set_lookahead(hc_ctx* ctx, int steps);
/* When the pass of a look-ahead block runs.  one_block_ahead < 0 (the default of every context): ADAPTIVE -- the library counts,
 * per block, how many of the gaps between the caller's synchronous steps (end of one hc_step / hc_step_multi to the begin of the
 * next) were longer than a threshold (HC_PASS_AHEAD_GAP_US; default 4 us, for wide systems 0 up to 12 GB of K in the context and a
 * tenth of the pass's cost per step above that) and runs the pass of the following block at block
 * start when the caller steps back to back, one block ahead when it is away between steps -- as a Chrono loop is; systems below
 * 256 MB of K (6N * 6N * S * 8 bytes) always run it at block start (their pass takes microseconds).  hc_step_multi measures the
 * gap once for its group of contexts, so the shards of one array decide alike.  The decisions are counted in
 * hc_profile_stats::schedule_blocks_ahead / schedule_blocks_at_start.
 * one_block_ahead = 0: when the block starts.  A caller that comes back before the pass has
 * finished waits for it on the first step of the block (190 us at 64 bodies; 1.55 ms for the 64-body row shard of a 512-body
 * array); a caller that stays away longer than that never notices it.  one_block_ahead = 1: the pass of the NEXT block is computed
 * from the history known when the current block starts, in `slices` launches (<= 0: chosen from the size of K, 2 .. 8) issued
 * behind the first steps of the current block, and what the current block's own samples add to the next block's steps follows in
 * short passes over the head of K.  With direct dispatch (and the device to itself) all of that runs BESIDE the steps, on a queue
 * of its own whose CU mask leaves a few compute units of every XCD to the step kernels; contexts that share a device, and steps
 * that go through HIP launches, issue the same launches on the step path's own queue / stream (the same sums; a back-to-back
 * caller then pays 10-30 % for them).  No step waits for a whole pass.
 * Measured on an MI355X, mean hc_step latency of a C++ caller with 30 / 100 / 300 us of host work between calls: 64 bodies
 * 18.9 -> 14.2 / 16.8 -> 14.1 / 14.0 -> 14.1 us (p99 at 30 us: 174 -> 19 us); the 64-of-512-body shard - / 66.8 -> 30.0 /
 * 60.1 -> 20.7 us (p99 at 300 us: 1306 -> 26 us).  A caller that steps back to back has nothing to hide the pass behind:
 * 19.5 -> 20.2 us at 64 bodies (the short pass towards the next block and the two queues sharing the chip), 74.7 -> 71.3 us for the shard.  Used once the history covers the IRF window; results are those of
 * schedule 0 up to the rounding of a different summation grouping (same 1e-6 contract, same tolerance on the predicted times).
 * The schedule is part of the configuration: the row shards of one array must use the same one (and the same slice count) to
 * stay bitwise equal to the unsharded context -- pin it (0 or 1) where that matters across separately driven contexts; under the
 * adaptive schedule two runs agree to rounding (1e-15), not bit for bit, when their callers' gaps differ.
 * HC_PASS_AHEAD=0/1 in the environment pins the default of new contexts (HC_PASS_CONCURRENT=0: no queue of its own); the slice count
 * is set through this call only.
 * REPRODUCIBILITY: with the schedule pinned (0 or 1) a run is bitwise repeatable whatever the caller's timing; under the adaptive
 * default it is repeatable to rounding ("Reproducibility"). */
int hc_set_pass_schedule(hc_ctx* ctx, int one_block_ahead, int slices);
/* What is in force: the look-ahead depth (0, 16 or 32: hc_set_lookahead clamps what it is given to what this build of the library
 * holds), the pass schedule (-1 adaptive, 0 at block start, 1 one block ahead), under the adaptive schedule the rule's current answer
 * (1: the next pass goes one block ahead), and the slice count of a pass made ahead.  Any pointer may be NULL.  For hosts that drive
 * the row shards of one array from several processes and want them to run ONE schedule: read it on one rank, pin it on all. */
int hc_get_schedule(const hc_ctx* ctx, int* lookahead, int* pass_schedule, int* ahead_now, int* slices);
/* How hc_step hands its kernels to the GPU.  1: as AQL packets written straight into an HSA queue of the library's own (kernel
 * arguments stored through the PCIe BAR) -- the default when the stand-alone code object hc_kernels.co lies next to the library,
 * the device's memory is host-addressable and the start-up self-tests pass (a dispatch completes; memory and argument slots the
 * host re-writes are re-read, not served stale); it saves the 2.4-3.3 us a hipLaunchKernelGGL call costs the host on the critical
 * path of every step (all kernels of hc_step / hc_step_begin / hc_step_multi and hc_added_mass_mv go this way, for systems of
 * every size).  0: through HIP launches on the context's stream (HC_DIRECT=0 forces this); hc_dispatch_mode_reason then says why.
 * Between steps of a caller that stays away for a while (a Chrono loop integrating) the library leaves its queue parked on a
 * barrier packet, so that the next step's kernel starts without the ~6 us an idle queue needs (HC_ARM=0 disables it).
 * A dispatch that never completes, or a queue the runtime reports broken, ends the wait after HC_STEP_TIMEOUT_S (default 20 s)
 * with HC_ERR_DEVICE; the context then fails every later step the same way.  The kernels and the results are the same either way.  hc_step_device always uses HIP,
 * and so does hc_step while hc_enable_profiling is on under a tool that intercepts HSA queues (rocprofv3): the tool sees direct
 * dispatches too, but the completion signals the library's own timings rest on are then the tool's. */
int hc_direct_dispatch_active(const hc_ctx* ctx);
const char* hc_dispatch_mode_reason(const hc_ctx* ctx);
/* Forget the velocity history and the per-time cache (fresh TestHydro state). */
int hc_reset_history(hc_ctx* ctx);
/* Injects a history as if those steps had been evaluated (times newest first, vel [n][D]); used to start
 * benchmarks in the steady state.  The only cross-step state of the path is this history plus the cached
 * time (checkpoint/resume). */
int hc_set_history(hc_ctx* ctx, int n, const double* times_newest_first, const double* vel_nxD);
int hc_get_history(hc_ctx* ctx, int* n, double* times_newest_first, double* vel_nxD); /* NULL arrays: size query */

/* ------------------------------------------------------------------------------------------------
 * Added mass.  Replaces ChLoadAddedMass (src/chloadaddedmass.cpp:12-70).
 * ---------------------------------------------------------------------------------------------- */
/* infinite_added_mass: D_local x D row-major, rho-scaled -- what ComputeJacobian puts top-left in M (:27-44) */
int hc_added_mass_matrix(hc_ctx* ctx, double* M_DlocalxD);
/* LoadIntLoadResidual_Mv: R[row0 + i] += c * sum_j M[i][j] * w[j], i < D_local (:55-70); w has >= D entries,
 * R has n_sys entries (n_sys >= D), row0 = 6*body_begin. */
int hc_added_mass_mv(hc_ctx* ctx, const double* w, double c, double* R_inout, int n_sys);
/* The same product for a row-sharded system held by n_ctx contexts of one process (see hc_step_multi): all shards are handed
 * to their GPUs first, then each shard's rows [6*body_begin, 6*body_end) of R are collected. */
int hc_added_mass_mv_multi(hc_ctx* const* ctxs, int n_ctx, const double* w, double c, double* R_inout, int n_sys);

/* ------------------------------------------------------------------------------------------------
 * Introspection (exporter / diagnostics parity).
 * ---------------------------------------------------------------------------------------------- */
/* HydroProfileStats (include/hydroc/hydro_forces.h:153-160), filled from HIP events; *_seconds are GPU
 * time of the kernels of each term.  conv_kernel_* describe the radiation GEMV kernel alone. */
typedef struct hc_profile_stats {
    /* HydroProfileStats: GPU seconds per term.  A launch that carries two terms (the convolution kernels stream K and,
     * for irregular waves, Kex) is apportioned by algorithmic bytes; hydrostatics_seconds is the step kernel (reduction,
     * the step's own newest-sample part, hydrostatics, regular / spectral wave term, total). */
    double hydrostatics_seconds, radiation_seconds, waves_seconds;
    int hydrostatics_calls, radiation_calls, waves_calls;
    double conv_kernel_seconds; /* sum of HIP-event durations of the plain per-step convolution launches */
    long long conv_kernel_launches;
    double conv_kernel_bytes;   /* algorithmic bytes of one step (8*D_local*D*S + vectors) */
    double block_kernel_seconds; /* look-ahead kernel launches (one covers a block of 32 or 16 steps) */
    long long block_kernel_launches;
    double block_kernel_bytes;  /* algorithmic bytes of the last pass: sum over its steps of the share of K (and of the
                                   velocity vector) that the pass computes for that step, i.e. IRF samples s >= s_cut[j] */
    double block_kernel_bytes_once; /* bytes the last LAUNCH of the pass kernel has to move once: live part of K, Kex, staged vectors
                                     * (the whole pass with the pass at block start, one slice of it under the schedule "one block
                                     * ahead"; block_kernel_bytes is scaled the same way) */
    double step_kernel_seconds;  /* the step kernel (finalize_kernel): the one launch on the critical path of a block step; wide
                                    systems (6N >= 1024): two launches per step, near_split_kernel + finalize_kernel, both counted */
    long long step_kernel_launches;
    double scatter_kernel_seconds; /* scatter launches (after a block step has delivered its forces) */
    long long scatter_kernel_launches;
    /* how the kernels of the per-step path reached the GPU since the last reset (counted whether or not profiling is on):
     * AQL packets written by the library itself / hipLaunchKernelGGL calls */
    long long direct_dispatches, hip_launches;
    long long history_rewinds;     /* steps back in time handled by dropping the newer history samples (see hc_step) */
    double mini_pass_seconds;      /* short passes of the two-level look-ahead of wide systems (one per sub-block of 8 steps) */
    long long mini_pass_launches;
    long long queue_parkings;      /* times the direct queue was left parked on a barrier packet after a step / an added-mass product */
    long long ahead_pass_slices;   /* pass schedule "one block ahead": launches of passes of a NEXT block (counted in block_passes too) */
    long long ahead_blocks;        /* ... and blocks that started with their rows already there (no pass at block start) */
    long long pass_lane_launches;  /* passes / short passes dispatched to the pass lane of the direct queue (they run beside the steps) */
    /* hc_step_multi: when this context's step kernel was handed to its GPU, measured from the entry of the call (seconds; the value
     * of the last call, the sum over all calls and their number) -- the fan-out cost of a multi-GPU step as each GPU sees it */
    double multi_doorbell_offset_last, multi_doorbell_offset_sum;
    long long multi_calls;
    long long slot_state_steps;    /* steps whose body state travelled behind the step kernel's argument block (direct dispatch, one-launch steps
                                    * of systems of up to 170 bodies, i.e. every system that is not wide): the kernel requests it together with its arguments, not after them */
    long long wide_fused_steps;    /* block steps of a wide system (6N >= 1024) whose own-sample slices and step kernel went out as ONE launch (wide_step_kernel) */
    long long schedule_blocks_ahead, schedule_blocks_at_start; /* look-ahead blocks at whose start the pass schedule answered "the pass of
                                    * the NEXT block runs one block ahead" / "at block start" (hc_set_pass_schedule; under the adaptive
                                    * schedule this is the rule's answer block by block) */
    long long ring_grows_for_pass; /* times the history ring was re-allocated so that a pass one block ahead can read its view of the
                                    * history while the block's steps push their samples (steps well below the IRF spacing) */
    long long hot_steps;           /* of slot_state_steps: block steps that went to the step kernel of the common case (step_hot_kernel: the
                                    * step's own IRF samples against its own velocity only, no plain partials, no spectral wave mode) */
} hc_profile_stats;
/* HIP events around the kernels of every `on`-th step (on = 1: every step; 0: off, the default), and around every
 * look-ahead pass (one per block) whatever the stride.  Event records perturb the launch stream by a few
 * microseconds, so throughput runs should sample (e.g. on = 17). */
int hc_enable_profiling(hc_ctx* ctx, int on);
int hc_get_profile(hc_ctx* ctx, hc_profile_stats* out);
int hc_reset_profile(hc_ctx* ctx);

/* What the INIT half of the path cost this context (seconds of host wall clock unless stated; GPU kernels by HIP events), stage by
 * stage, with the bytes each stage has to move -- so that a host (and bench.py's `init` block) can put every stage beside its
 * bound: the HDF5 reads beside the file size, the staging copies beside the PCIe rate, the re-layout / TaperedDirect / generator
 * kernels beside the HBM rate, the free-surface synthesis beside the FP64 rate.  Accumulated since hc_create. */
typedef struct hc_init_stats {
    double h5_read_seconds, h5_read_bytes;             /* hc_load_bemio_h5: HDF5 reads (datasets this context reads) */
    double rirf_h2d_seconds, rirf_h2d_bytes;           /* radiation IRF tensors {6, 6N, S} per owned body, pageable host memory -> HBM staging */
    double rirf_relayout_seconds, rirf_relayout_bytes; /* relayout_rirf_kernel: file order -> panel layout, rho folded in (bytes read + written) */
    double finalize_seconds;                           /* hc_finalize: widths, hydrostatic tables, added mass, buffers, direct-dispatch set-up + self-tests */
    double direct_setup_seconds;                       /* ... of which the direct-dispatch set-up (code object, queues, self-tests) */
    double wave_resample_seconds;                      /* hc_set_wave_irregular: excitation-IRF resample (LinSpaced + cubic B-spline, host) */
    double wave_spectrum_seconds;                      /* ... spectrum, phases, wavenumbers (host) */
    double wave_eta_seconds;                           /* ... free-surface table on the GPU: eta_kernel (direct FP64 sum) or the rocFFT chirp-z form */
    long long wave_eta_samples, wave_eta_components;   /* nt, nf of that table */
    int wave_eta_mode, pad_;                           /* 0 direct sum, 1 rocFFT */
    double wave_upload_seconds, wave_upload_bytes;     /* ... Kex re-layout, table uploads, read-back of eta */
    double wave_total_seconds;                         /* ... the whole call */
    double taper_seconds, taper_bytes;                 /* TaperedDirect preprocessing (taper_kernel; bytes read + written) */
    double synth_seconds, synth_bytes;                 /* hc_synth_fill: the generator kernel (bytes written) */
} hc_init_stats;
int hc_get_init_stats(const hc_ctx* ctx, hc_init_stats* out);

/* Sizes: S radiation samples, L resampled excitation samples, nf wave components, nt eta samples,
 * H current history length, Hcap ring capacity. Any pointer may be NULL. */
int hc_get_sizes(hc_ctx* ctx, int* N, int* n_local, int* S, int* L, int* nf, int* nt, int* H, int* Hcap);
/* rirf_width_vector (src/hydro_forces.cpp:181-190) */
int hc_get_rirf_width(hc_ctx* ctx, double* w_S);
/* The kernel actually convolved, mapped back to reference indexing: value = GetRIRFval(row, col, s)
 * (src/hydro_forces.cpp:693-711), i.e. rho-scaled and, in TaperedDirect mode, processed (:385-535).
 * rows are local; out is [D_local][D][S].  Meant for small cases. */
int hc_get_rirf_effective(hc_ctx* ctx, double* out_DlocalxDxS);
/* One value of the same: TestHydro::GetRIRFval(row, col, st) (src/hydro_forces.cpp:693-711) for a LOCAL row; indices outside
 * [0, D_local) x [0, D) x [0, S) give HC_ERR_OUT_OF_RANGE (the reference throws std::out_of_range, :694-697). */
int hc_get_rirf_value(hc_ctx* ctx, int row_local, int col, int st, double* out);
/* ex_irf_time_sampled_, ex_irf_width_sampled_, ex_irf_sampled_ (6 x L) of a local body (src/wave_types.cpp:572-628) */
int hc_get_excitation_irf_resampled(hc_ctx* ctx, int body, double* t_L, double* width_L, double* vals_6xL);
/* Bodies may carry different excitation-IRF time grids (the reference keeps one per body, src/wave_types.cpp:432-459): L of this
 * body's resampled grid (hc_get_sizes reports the sum over the distinct grids = the columns of the excitation matrix). */
int hc_get_excitation_irf_size(hc_ctx* ctx, int body, int* L);
/* spectrum_frequencies_, spectral_densities_, spectral_widths_, wave_phases_, wavenumbers_ (:643-676) */
int hc_get_spectrum(hc_ctx* ctx, double* f, double* S, double* df, double* phase, double* k);
/* free_surface_time_sampled_ / free_surface_elevation_sampled_ (:717-774; exporter: runner:668-679) */
int hc_get_eta_table(hc_ctx* ctx, double* t_nt, double* eta_nt);
/* SimulationExporter::WriteIrregularInputs (src/simulation_exporter.cpp:365-393): writes frequencies_hz, spectral_densities,
 * free_surface_time, free_surface_eta (+ the reference's attributes) under /inputs/simulation/waves/irregular of an HDF5
 * result file (created if absent).  Needs libhdf5 (HC_ERR_UNSUPPORTED otherwise). */
int hc_export_irregular_inputs_h5(hc_ctx* ctx, const char* path);
/* RegularWave::excitation_force_mag_/phase_ and wavenumber_ (:278-299) */
int hc_get_regular_coeffs(hc_ctx* ctx, double* mag_D, double* phase_D, double* wavenumber);

/* ------------------------------------------------------------------------------------------------
 * Synthetic many-body inputs generated directly in HBM (benchmark configurations C3/C4 of SURVEY 8d;
 * not part of the reference).  Fills K, K_hs, A_inf, excitation IRF for all local bodies from a
 * counter-based generator so that a 77 GB kernel never exists on the host.  hc_finalize still applies.
 * ---------------------------------------------------------------------------------------------- */
int hc_synth_fill(hc_ctx* ctx, unsigned long long seed, int S, double dt_rirf, int n_exc, double dt_exc);

#ifdef __cplusplus
}
#endif
#endif /* HYDROCHRONO_AMD_H */
