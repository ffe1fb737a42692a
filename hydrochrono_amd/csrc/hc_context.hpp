// hc_context.hpp -- host-side state behind an hc_ctx (see include/hydrochrono_amd.h).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hydrochrono_amd.h"
#include "hc_direct.hpp"
#include "hc_kernels.hpp"
#include "hc_plan.hpp"

namespace hc {

// Error carrying the C status it maps to.
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

#define HC_HIP(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            throw ::hc::Error(HC_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));       \
    } while (0)

template <class T>
struct DeviceBuffer {
    T* p     = nullptr;
    size_t n = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&)            = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) return;
        HC_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
    }
    void upload(const std::vector<T>& h, hipStream_t s) {
        if (n != h.size()) alloc(h.size());
        if (n) {
            HC_HIP(hipMemcpyAsync(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice, s));
            HC_HIP(hipStreamSynchronize(s));  // source is pageable; keep lifetime simple at init
        }
    }
};

template <class T>
struct PinnedBuffer {
    T* p     = nullptr;  // host address
    T* dp    = nullptr;  // the same memory as seen by the GPU (zero-copy)
    size_t n = 0;
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&)            = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() {
        if (p) (void)hipHostFree(p);
    }
    void alloc(size_t count) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
        if (count == 0) return;
        HC_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent));
        HC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dp), p, 0));
        n = count;
    }
};

// Device memory the host can write directly through the PCIe BAR (fine-grained allocation; host-visible where the whole
// VRAM is BAR-mapped, as on MI355X servers).  hc_step puts the body state there: the step kernel then reads it locally
// instead of fetching it from pinned host memory over PCIe (10.1 vs 12.9 us launch-to-result for the same kernel,
// profiles/r02/latency_probe2.txt).  host_ok == false: not host-visible here, the caller falls back to pinned memory.
template <class T>
struct BarBuffer {
    T* p         = nullptr;  // device address == host address when host_ok
    size_t n     = 0;
    bool host_ok = false;
    BarBuffer() = default;
    BarBuffer(const BarBuffer&)            = delete;
    BarBuffer& operator=(const BarBuffer&) = delete;
    ~BarBuffer() {
        if (p) (void)hipFree(p);
    }
    void alloc(size_t count);  // hc_runtime.cpp (probes host visibility without faulting)
};

struct BodyHost {
    bool have_props = false, have_lin = false, have_ainf = false, have_rirf = false, have_rao = false, have_exirf = false;
    double disp_vol = 0.0, cg[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    double lin[36]  = {0};
    std::vector<double> ainf;       // [6][D], rho-scaled (local bodies only)
    std::vector<double> rao_w;      // [nw]
    std::vector<double> rao_mag;    // [6][nw] rho*g-scaled
    std::vector<double> rao_phase;  // [6][nw]
    std::vector<double> exirf_t;    // [n]
    std::vector<double> exirf_f;    // [6][n] rho*g-scaled (local bodies only)
};

// What a step still has to enqueue for LATER steps (scatter of its sample, or the plan + pass of the next block) once its step
// kernel is on its way: hc_step_multi rings the step kernels of all shard contexts first and enqueues these tails afterwards.
struct StepTail {
    bool pending = false, rad = false, waves = false, block = false, direct = false, caller_waits = false;
    int m = 0, H = 0;
    hipStream_t stream = nullptr;
};

// Pass schedule "one block ahead" (hc_step.cpp, hc_plan.hpp: FarPass): the pass of the NEXT block while the current one is stepped.
struct AheadPass {
    bool active = false;                 // its launches are going out / its rows are being completed by the short passes of this block
    unsigned long long plan_serial = 0;  // the plan (block) it is computed under
    int slices = 0, issued = 0;          // launches the radiation chunks are spread over, and how many have gone out
    int per_slice = 0;                   // radiation chunks per launch (whole octets)
    bool reduced = false;                // the reduction into the next block's rows has gone out
    bool concurrent = false;             // it runs on the pass lane of the direct queue, beside the steps (ordered by signals)
    bool has_exc = false;                // it also leaves the excitation force of the next block's predicted times
    int Hcap = 0;                        // ring capacity of the history view (a re-allocated ring voids the view)
    int head0 = 0, Hv = 0;               // newest stored slot when the view was taken, samples the view spans (incl. the virtual one)
    double t_first = 0.0, t_last = 0.0;  // predicted times of the next block's first and last step
    double rad_once = 0.0, exc_once = 0.0;
    BlockArgs args;                      // the whole launch; chunk_first / chunk_last are set per slice
};

// Bodies that share one excitation-IRF time grid: columns [off, off + L) of Kex / ex_tau / ex_width are the group's resampled grid.
struct ExGroup {
    int first_body = 0, off = 0, L = 0;
    double tau_front = 0.0, tau_back = 0.0;
};

enum WaveKind { kWaveNone = 0, kWaveRegular = 1, kWaveIrregular = 2, kWaveSpectral = 3 };

// One timed kernel launch (HIP events before / after it on the stream it was launched on).
enum EventKind { kEvConvPlain = 0, kEvPass = 1, kEvStep = 2, kEvScatter = 3, kEvConvExc = 4, kEvMiniPass = 5 };
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;
    double waves_share = 0.0;  // share of the launch that is excitation work (by algorithmic bytes)
};

}  // namespace hc

struct hc_ctx {
    int N = 0, b0 = 0, b1 = 0, nloc = 0, D = 0, Dloc = 0, device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_am = nullptr;  // added-mass products (independent of the step kernels)
    // hc_step_device on a caller's stream: only the step kernel goes to that stream; what later steps need (scatter, pass)
    // runs on the context's own stream behind ev_fin, and the next step kernel waits for ev_bg
    hipEvent_t ev_fin = nullptr, ev_bg = nullptr;
    bool bg_pending = false;
    // stream the last step's kernels went to: a step on another stream is ordered behind it with an event
    hipStream_t last_stream = nullptr;
    bool have_last_stream   = false;
    // Direct AQL dispatch of the synchronous step path (hc_direct.hpp): the step kernel, scatter, pass and its reduction go to a
    // queue of our own when hc_step runs a step that needs no plain convolution launch.  path: where the kernels of the last
    // step went (0 nothing yet, 1 HIP stream(s), 2 direct queue); the other side is drained at every switch.
    hc::DirectQueue* dq = nullptr;
    bool direct_ready   = false;
    std::string direct_why;  // why the direct path is not in use
    std::string direct_how = "direct AQL dispatch";  // ... or how it is (where the runtime put the packet ring, hc_dispatch_mode_reason)
    int path            = 0;
    hc::DirectKernel dk_finalize_slot;  // finalize_kernel<4, true>: the body state behind the argument block (hc_step.cpp: HostState)
    hc::DirectKernel dk_finalize_pre;   // finalize_pre_kernel<4>: the same with its first loads' addresses preloaded into scalar registers
    bool step_preload = false;          // ... in use for this context (tuning build, HC_STEP_PRELOAD=1: measured and not taken, EXPERIMENTS.md)
    hc::DirectKernel dk_step_hot[2];    // step_hot_kernel<1>, <2>: the common block step with a compact argument block (hc_kernels.hpp: StepHotArgs)
    bool step_hot = false;              // ... in use for this context
    int step_halves = 1;                // 2: two workgroups per row tile in step_hot_kernel (systems of kStepHalvesMinColumns columns or more)
    bool slot_state = false;            // ... in use for this context (HC_SLOT_STATE, systems of up to kSlotStateMaxBodies bodies)
    hc::DirectKernel dk_finalize, dk_scatter, dk_reduce, dk_block16, dk_block32, dk_block64, dk_mini16, dk_mini32, dk_narrow, dk_wide, dk_added_mass, dk_step, dk_near;
    hc::StepTail tail;
    // split step (hc_step_begin / hc_step_end, hc_step_multi): 0 nothing begun, 1 the begun step was a cache hit (totals in
    // last_total), 2 its results arrive as tagged granules with sequence number `seq`
    int pending_step  = 0;
    double pending_t  = 0.0;
    int pending_am    = 0;      // hc_added_mass_mv in flight (hc_added_mass_mv_multi): 1 tagged on the direct lane, 2 on stream_am
    bool lost         = false;  // a dispatch never completed or the queue reported an error: every later step fails with HC_ERR_DEVICE
    // Arming of the direct queue (DirectQueue::arm): after a step whose caller had been away for a while before it (the gap between
    // the end of the previous call and this one: a Chrono loop integrates in between), the queue is left parked on a barrier packet
    // so that the next step's kernel starts without the idle penalty.  arm_mode: 0 never, 1 adaptive (default), 2 always.
    int arm_mode = 1;
    bool arm_after_step = false, arm_after_am = false;
    std::chrono::steady_clock::time_point t_step_end{}, t_am_end{};
    bool have_t_step_end = false, have_t_am_end = false;
    int busy_caller_steps   = 0;  // hc_step_device: steps left before the caller's stream is queried again (see enqueue_step)
    std::string err;

    bool have_sim = false;
    double rho = 0, g = 0, depth = 0;
    double gsys[3] = {0.0, 0.0, -9.81};
    std::vector<hc::BodyHost> bodies;

    // radiation IRF
    int S = 0;
    std::vector<double> tau, width;
    hc::DeviceBuffer<double> dK, dKproc, d_tau, d_width, d_stage;
    int ntiles = 0, Dpad = 0, ngp = 0, mt = 1, ngroups = 0;  // panel geometry (hc_kernels.hpp)
    int conv_mode = 0;
    bool proc_ready = false;
    hc_tapered_direct_options taper{};
    std::string diagnostics_dir;  // SetDiagnosticsOutputDirectory ("" = current directory)
    bool finalized = false;

    // history (host mirror of times, newest first) + ring in HBM
    std::deque<double> times;
    std::deque<double> retired;  // samples the prune rule has dropped, newest first; still in the ring slots behind the oldest kept one (history_push)
    int head = -1, Hcap = 0, HcapT = 0;
    hc::DeviceBuffer<double> d_ring_t, d_ring_v, d_ring_vT;  // ring_vT[D][HcapT = Hcap + 2]: per-DoF copy for the look-ahead pass
    long long rewinds = 0;  // steps back in time handled so far (history_push)
    bool have_prev = false, have_prev_device = false;  // per-time cache of hc_step (host totals) / hc_step_device (d_total)
    double prev_time = -1.0, prev_time_device = -1.0;

    // hydrostatics / added mass
    hc::DeviceBuffer<double> d_lin, d_cg, d_cbmcg, d_vol, d_ainf, d_vec_w, d_vec_R;
    std::vector<double> ainf_host;  // [Dloc][D]

    // waves
    int wave_kind = hc::kWaveNone;
    int wave_nb_arg = 0;
    double reg_amp = 0, reg_omega = 0, reg_wavenumber = 0;
    std::vector<double> reg_mag, reg_phase;  // [D] each (all bodies)
    hc::DeviceBuffer<double> d_reg_mag;      // [Dloc]
    hc_irregular_wave_params irr{};
    int eta_mode = 0;  // 0: direct FP64 sum (eta_kernel), 1: rocFFT chirp-z (hc_eta_fft.cpp)
    int L = 0, Lpad = 0, nf = 0, nt = 0;
    std::vector<double> ex_tau, ex_width, ex_vals;  // [L] = the groups' grids one after the other; ex_vals [Dloc][L]
    std::vector<hc::ExGroup> ex_groups;             // per-body excitation-IRF grids (src/wave_types.cpp:432-459), see hc_set_wave_irregular
    std::vector<int> ex_group_of;                   // [N]
    std::vector<double> spec_f, spec_S, spec_df, spec_phase, spec_k;
    std::vector<double> eta_t, eta;
    hc::DeviceBuffer<double> d_kex, d_ex_tau, d_ex_width, d_eta_t, d_eta;
    hc::DeviceBuffer<double> d_spec_mag, d_spec_phase, d_spec_amp, d_spec_omega, d_spec_phi;  // spectral (component-sum) mode

    // GEMV configuration + scratch
    int chunk_gp = 0, nchunks_rad = 0, chunk_gp_ex = 0, nchunks_ex = 0, ngp_ex = 0;
    int chunk_gp_block = 0, nchunks_block = 0;
    hc::DeviceBuffer<double> d_partials, d_partials_block, d_P, d_E;
    hc::DeviceBuffer<double> d_near_partials;  // [16][Dpad] slice partials of near_split_kernel (wide systems)
    hc::DeviceBuffer<int> d_tile_counter;      // [ntiles] arrival counters of wide_step_kernel (zero between launches)
    bool tile_counter_suspect = false;         // a step failed after it may have dispatched a wide_step_kernel: the counters are cleared before the next one
    hc::DeviceBuffer<double> d_Y;           // weighted scatter results per consumer step [kLookahead + 1][kTermMax][Dpad]
    hc::DeviceBuffer<double> d_zero_state;  // 12N zeros: the not-yet-known sample of the look-ahead pass
    int chunk_gp_ex_block = 32, nchunks_ex_block = 0;  // excitation chunks of the look-ahead launch
    int mt_mini = 2;                                    // row tiles per workgroup of the short passes (two-level form)
    int mt_narrow = 2;                                  // ... of their narrow form (16 step columns)
    int mt_block64 = 3;                                 // ... of the experimental depth-64 pass (HC_BLOCK64_MT: 3, 4 or 6)
    int mt_block = 4, mt_block_design = 6;              // row tiles per workgroup of the look-ahead launch (1, 2, 4, 6 or 12)
    int num_cus  = 256;                                 // compute units of the device (grid rounds of the look-ahead launch)
    int lookahead = 0;  // 0: off, else 16, 32 (kDepthDefault) or 64 (experimental, hc_set_lookahead)
    hc::Plan plan;
    unsigned long long plan_serial = 0;  // counts the plans made
    // pass schedule (hc_set_pass_schedule): 0 = the pass of a block when the block starts, 1 = one block ahead, in `pass_slices`
    // launches between the steps of the block before, 2 = adaptive (the default): per block, from the caller's gaps between the
    // synchronous steps of the block before (hc_pass.cpp: schedule_ahead_for_next_block).  d_P / d_E hold two blocks of rows;
    // pe_cur = the half of the current block.
    int pass_ahead = 0, pass_slices = 8, pe_cur = 0;
    bool ahead_now = false;        // adaptive: the rule's last answer (starts as the static choice by size: wide systems run ahead)
    int gap_seen = 0, gap_long = 0, gap_long_lo = 0;  // gaps measured since the last decision; how many of them were longer than
                                   // gap_threshold (it takes a majority of these to GO ahead) / than 0.6 of it (... to STAY ahead)
    double gap_threshold = 4e-6;   // seconds (HC_PASS_AHEAD_GAP_US; wide systems: 0 up to 12 GB of K in the context, then a tenth of the pass's cost per step, hc_setup.cpp)
    double gap_hint = -1.0;        // hc_step_multi: the gap its calling thread measured for the whole group (< 0: none, measure here)
    std::chrono::steady_clock::time_point t_multi_end{};  // (kept on the group's first context) end of the last hc_step_multi
    bool have_t_multi_end = false;
    hc::AheadPass ahead;
    hc::DeviceBuffer<double> d_partials_far;  // partial sums of the pass in the making (the short passes keep d_partials_block)
    hc::DeviceBuffer<double> d_partials_next; // ... and of the short passes towards the next block when they run on the pass lane
    // The pass lane (lane 2 of the direct queue): passes in the making run there BESIDE the steps of lane 0, on a queue that leaves
    // pass_free_cus compute units of every XCD to the step kernels.  0 not created yet, 2 in use, -1 unusable.
    bool pass_concurrent = true;
    int pass_lane = 0, pass_free_cus = 4;

    // step I/O
    hc::DeviceBuffer<double> d_state, d_hs, d_rad, d_waves, d_total;
    hc::DeviceBuffer<double> d_scratch;  // [4][Dloc] outputs of the term-only entry points (they must not clobber the last step)
    hc::DeviceBuffer<int> d_err;
    hc::PinnedBuffer<double> h_state, h_out, h_am;
    hc::BarBuffer<double> bar_state;  // [2][12N + 1] body state (+ the step's canary word) written by the host through the BAR (hc_step)
    hc::BarBuffer<double> bar_am;     // [D + Dloc] w and incoming R of hc_added_mass_mv
    hc::PinnedBuffer<unsigned long long> h_tag_am;  // [Dloc][2] its tagged result
    int am_lane = 0;  // second lane of the direct queue (added-mass products): 0 not created yet, 1 in use, -1 unusable (HIP launches)
    hc::BarBuffer<double> bar_selftest;                    // [2] inputs of the direct-dispatch self-test
    hc::PinnedBuffer<unsigned long long> h_tag_selftest;   // [2] its tagged result
    hc::DeviceBuffer<double> d_selftest;                   // [1]
    unsigned long long seq_am = 0;
    hc::PinnedBuffer<unsigned long long> h_canary;  // 2 x [2] {canary, sequence number}: the word hc_step stored behind the state, handed back by the step kernel
    const double* step_canary_in = nullptr;         // set by step_begin for the enqueue_step it makes (device-visible address of that word)
    unsigned long long* step_canary_out = nullptr;
    long long fault_stale_state_at = -1;            // HC_FAULT_STALE_STATE_AT (tests): the step with this sequence number carries the previous step's canary
    hc::PinnedBuffer<unsigned long long> h_tag;  // 2 x [Dloc][2] {total, sequence number} granules written by finalize_kernel (halves by sequence parity)
    unsigned long long seq = 0;
    unsigned long long *ext_tag_host = nullptr, *ext_tag_dev = nullptr;  // the caller's result buffer (hc_set_result_buffer), else h_tag
    std::vector<double> last_total;               // totals of the last evaluated step (duplicate-time cache of hc_step)
    int zero_copy_max_bodies = 64;                // hc_step: kernels read the state from mapped pinned memory up to this size
    hc::PinnedBuffer<int> h_err;
    bool device_errors_possible = false;  // radiation IRF times < 0 (the only way a per-step query can leave its bracket)

#ifdef HC_TUNING
    // stage clock of the step kernel (HC_STEP_STAMPS=1; hc_tuning_step_stamps): device rows [kStampSteps][kStampWGs][kStampStages] and
    // the host's own stamps of the same steps {entry of hc_step, doorbell of the step kernel, totals seen}, HSA system ticks
    bool stamps_on = false;
    hc::DeviceBuffer<unsigned long long> d_stamps;
    unsigned long long host_stamps[hc::kStampSteps][4] = {};
#endif

    hc_init_stats init{};  // what the init half cost, stage by stage (hc_get_init_stats)

    // profiling
    bool profiling = false;
    int profile_stride = 1;
    long long profile_counter = 0;
    std::vector<hc::EventPair> events;
    size_t events_used = 0;
    bool sample_this_step = false;
    hc_profile_stats prof{};
};
