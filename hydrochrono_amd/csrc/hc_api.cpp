// hc_api.cpp -- implementation of the C ABI declared in include/hydrochrono_amd.h.
//
// Host bookkeeping only: ingest/scaling, the time history mirror (push / prune exactly like
// src/hydro_forces.cpp:327-340,549-584), error rules of the reference, and the three kernel launches per step.
// All arithmetic of the per-step force path runs in hc_kernels.hip on the GPU; there is no CPU fallback.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <xmmintrin.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>

#include "hc_context.hpp"
#include "hc_history.hpp"
#include "hc_host_math.hpp"

using hc::Error;

// Fine-grained device allocation + a fault-free test of whether the CPU can address it: read(2) from /dev/zero INTO the
// buffer and write(2) FROM it fail with EFAULT instead of raising SIGSEGV when the range is not mapped for the host.
template <class T>
void hc::BarBuffer<T>::alloc(size_t count) {
    if (p) (void)hipFree(p);
    p       = nullptr;
    n       = 0;
    host_ok = false;
    if (count == 0) return;
    static const bool disabled = [] { const char* e = std::getenv("HC_NO_BAR_STATE"); return e && std::atoi(e) != 0; }();
    if (disabled) return;
    void* q = nullptr;
    if (hipExtMallocWithFlags(&q, count * sizeof(T), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    p = static_cast<T*>(q);
    n = count;
    const int fz = open("/dev/zero", O_RDONLY), fn = open("/dev/null", O_WRONLY);
    if (fz >= 0 && fn >= 0) {
        const size_t bytes = count * sizeof(T);
        host_ok = read(fz, p, bytes) == static_cast<ssize_t>(bytes) && write(fn, p, bytes) == static_cast<ssize_t>(bytes);
    }
    if (fz >= 0) close(fz);
    if (fn >= 0) close(fn);
}
template struct hc::BarBuffer<double>;

namespace hc {
void eta_synthesis_fft(const std::vector<double>& t, const std::vector<double>& amp, const std::vector<double>& omega,
                       const std::vector<double>& phase, double ramp, const double* d_t, double* d_eta, hipStream_t stream);
}

namespace {

thread_local std::string g_create_error;

const char* kVersion = "hydrochrono_amd 0.1 (gfx950)";

#define HC_API_BEGIN_HOT(ctx)                                \
    if (!(ctx)) return HC_ERR_INVALID;                       \
    try {                                                    \
        HC_HIP(hipSetDevice((ctx)->device));

// every entry point but the per-step ones first waits for what the direct queue still runs (hc_direct.hpp): their HIP work is
// not ordered against it
#define HC_API_BEGIN(ctx)                                    \
    HC_API_BEGIN_HOT(ctx)                                    \
    quiesce_direct(ctx);

#define HC_API_END(ctx)                                      \
    }                                                        \
    catch (const Error& e) {                                 \
        (ctx)->err = e.what();                               \
        return e.status;                                     \
    }                                                        \
    catch (const std::out_of_range& e) {                     \
        (ctx)->err = e.what();                               \
        return HC_ERR_OUT_OF_RANGE;                          \
    }                                                        \
    catch (const std::exception& e) {                        \
        (ctx)->err = e.what();                               \
        return HC_ERR_RUNTIME;                               \
    }                                                        \
    return HC_OK;

// Bounded: a queue that does not drain within HC_STEP_TIMEOUT_S (or reports an error) is a lost device -> HC_ERR_DEVICE.
void quiesce_direct(hc_ctx* c) {
    if (c->dq && c->dq->busy()) {
        static const double limit = [] { const char* e = std::getenv("HC_STEP_TIMEOUT_S"); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 20.0; }();
        if (!c->dq->drain(limit)) {
            c->lost         = true;
            c->direct_ready = false;
            c->direct_why   = c->dq->failed(0) ? "the HSA queue of the direct dispatch reported an error: " + c->dq->failure_text()
                                               : std::string("the direct queue did not drain (timeout)");
            throw Error(HC_ERR_DEVICE, c->direct_why);
        }
    }
    if (c->path == 2) c->path = 1;
}

void require(bool cond, int status, const char* msg) {
    if (!cond) throw Error(status, msg);
}

void check_body(const hc_ctx* c, int body) {
    if (body < 0 || body >= c->N) throw Error(HC_ERR_OUT_OF_RANGE, "body index out of range");
}
bool is_local(const hc_ctx* c, int body) { return body >= c->b0 && body < c->b1; }

// ---- history ring -----------------------------------------------------------------------------
void ring_alloc(hc_ctx* c, int cap) {
    c->d_ring_t.alloc(cap);
    c->d_ring_v.alloc(static_cast<size_t>(cap) * c->D);
    c->d_ring_vT.alloc(static_cast<size_t>(cap + 2) * c->D);  // rows of Hcap + 2: entry [Hcap] mirrors slot 0 (hc_kernels.hpp)
    HC_HIP(hipMemsetAsync(c->d_ring_t.p, 0, cap * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(c->d_ring_v.p, 0, static_cast<size_t>(cap) * c->D * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(c->d_ring_vT.p, 0, static_cast<size_t>(cap + 2) * c->D * sizeof(double), c->stream));
    c->Hcap  = cap;
    c->HcapT = cap + 2;
    c->head = -1;
}

// Grow the ring so that `need` samples fit, keeping the `have` newest stored samples (k = 0..have-1) in order.
void ring_grow(hc_ctx* c, int need, int have) {
    quiesce_direct(c);               // the scatter / pass of the last step may still be reading the ring on the direct queue
    HC_HIP(hipDeviceSynchronize());  // rare; steps may have been enqueued on a caller's stream (hc_step_device)
    const int cap2 = std::max(2 * c->Hcap, need + 16);
    hc::DeviceBuffer<double> nt, nv;
    nt.alloc(cap2);
    nv.alloc(static_cast<size_t>(cap2) * c->D);
    HC_HIP(hipMemsetAsync(nt.p, 0, cap2 * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(nv.p, 0, static_cast<size_t>(cap2) * c->D * sizeof(double), c->stream));
    // new layout: sample k -> slot (have-1-k); oldest at slot 0, newest at slot have-1
    for (int k = 0; k < have; ++k) {
        const int src = ((c->head - k) % c->Hcap + c->Hcap) % c->Hcap;
        const int dst = have - 1 - k;
        HC_HIP(hipMemcpyAsync(nt.p + dst, c->d_ring_t.p + src, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(nv.p + static_cast<size_t>(dst) * c->D, c->d_ring_v.p + static_cast<size_t>(src) * c->D,
                              c->D * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    c->d_ring_vT.alloc(static_cast<size_t>(cap2 + 2) * c->D);
    HC_HIP(hipMemsetAsync(c->d_ring_vT.p, 0, static_cast<size_t>(cap2 + 2) * c->D * sizeof(double), c->stream));
    hc::launch_ring_transpose(nv.p, cap2, cap2 + 2, c->D, c->d_ring_vT.p, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipStreamSynchronize(c->stream));
    std::swap(c->d_ring_t.p, nt.p);
    std::swap(c->d_ring_t.n, nt.n);
    std::swap(c->d_ring_v.p, nv.p);
    std::swap(c->d_ring_v.n, nv.n);
    c->Hcap  = cap2;
    c->HcapT = cap2 + 2;
    c->head  = have - 1;
}

// Push the time of this step and prune like PruneHistory; returns H (samples incl. the current one).  The bookkeeping -- the
// reference's push / prune rule and what a step back in time does -- is hc_history.hpp (host only, unit-tested on CPU).
int history_push(hc_ctx* c, double t) {
    const hc::HistoryAdvance r = hc::history_advance(c->times, c->retired, c->head, c->Hcap, t, c->tau.empty() ? 0.0 : c->tau.back());
    if (r.status == hc::HistoryAdvance::kDuplicateTime)
        throw Error(HC_ERR_RUNTIME, "Tried to compute the radiation damping convolution twice within the same time step!");
    if (r.rewound) {
        // everything already enqueued (a scatter, a pass of the abandoned plan) runs before this step's kernels on the same queue,
        // so no ring slot is overwritten under a reader; the plan of the abandoned attempt is void
        c->plan.valid  = false;
        c->plan.misses = 0;
        c->rewinds++;
        c->prof.history_rewinds++;
    }
    if (r.grow) ring_grow(c, r.grow_need, r.grow_have);
    c->head = (c->head + 1) % c->Hcap;
    return r.H;
}

// ---- profiling --------------------------------------------------------------------------------
// Timed launches carry their own pair of HIP events, recorded on the stream the launch went to (a caller's stream in
// hc_step_device), so the drain waits on the events themselves, not on a particular stream.
void profile_account(hc_ctx* c, int kind, double sec, double waves_share) {
    switch (kind) {
        case hc::kEvConvPlain:  // radiation (+ irregular-wave excitation chunks) of a plain step
            c->prof.conv_kernel_seconds += sec;
            c->prof.conv_kernel_launches += 1;
            c->prof.radiation_seconds += sec * (1.0 - waves_share);
            c->prof.waves_seconds += sec * waves_share;
            break;
        case hc::kEvPass:  // look-ahead pass: radiation part of a block of steps (+ their excitation force)
            c->prof.block_kernel_seconds += sec;
            c->prof.block_kernel_launches += 1;
            c->prof.radiation_seconds += sec * (1.0 - waves_share);
            c->prof.waves_seconds += sec * waves_share;
            break;
        case hc::kEvStep:  // the step kernel: reduction, own-sample part, hydrostatics, regular / spectral wave term
            c->prof.step_kernel_seconds += sec;
            c->prof.step_kernel_launches += 1;
            c->prof.hydrostatics_seconds += sec;
            break;
        case hc::kEvScatter:
            c->prof.scatter_kernel_seconds += sec;
            c->prof.scatter_kernel_launches += 1;
            c->prof.radiation_seconds += sec;
            break;
        case hc::kEvMiniPass:  // short pass of the two-level form (one per sub-block of a wide system)
            c->prof.mini_pass_seconds += sec;
            c->prof.mini_pass_launches += 1;
            c->prof.radiation_seconds += sec;
            break;
        default:  // excitation-only convolution launch
            c->prof.waves_seconds += sec;
            break;
    }
}

void profile_drain(hc_ctx* c) {
    for (size_t i = 0; i < c->events_used; ++i) {
        hc::EventPair& ev = c->events[i];
        HC_HIP(hipEventSynchronize(ev.b));
        float ms = 0;
        HC_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        profile_account(c, ev.kind, ms * 1e-3, ev.waves_share);
    }
    c->events_used = 0;
    // dispatches of the direct queue carry completion signals instead of events
    if (c->dq && c->dq->timed_pending() > 0)
        c->dq->collect([c](int kind, double sec, double share) { profile_account(c, kind, sec, share); });
}

constexpr size_t kEventPoolMax = 4096;

// Called at the top of a step, before anything is enqueued or the history is touched: decides whether this step's launches
// are timed (every stride-th step) and makes room in the event pool (draining synchronises, so it must not happen between
// the history push and the launches).
void profile_begin_step(hc_ctx* c) {
    c->sample_this_step = false;
    if (!c->profiling) return;
    if (c->events_used + 8 > kEventPoolMax || (c->dq && c->dq->timed_pending() + 8 > kEventPoolMax)) profile_drain(c);
    c->sample_this_step = (c->profile_counter++ % c->profile_stride) == 0;
}

// Event pair around one launch, or null.  The look-ahead pass (one per block) is timed whatever the stride.
hc::EventPair* ev_begin(hc_ctx* c, int kind, hipStream_t stream, double waves_share = 0.0) {
    if (!c->profiling || !(c->sample_this_step || kind == hc::kEvPass)) return nullptr;
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= kEventPoolMax) return nullptr;
        hc::EventPair ev;
        HC_HIP(hipEventCreate(&ev.a));
        HC_HIP(hipEventCreate(&ev.b));
        c->events.push_back(ev);
    }
    hc::EventPair* ev = &c->events[c->events_used++];
    ev->kind        = kind;
    ev->waves_share = waves_share;
    HC_HIP(hipEventRecord(ev->a, stream));
    return ev;
}
// A profiling tool that intercepts the HSA queues (rocprofv3, roctracer: they arrive through these variables) replaces the
// completion signals of our packets with its own, and hsa_amd_profiling_get_dispatch_time on ours then returns nothing useful
// (1.3 us for a 190 us pass).  The tool still sees every dispatch; only the library's own timings need the HIP path then.
bool profiling_tool_attached() {
    static const bool attached = [] {
        if (std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("HSA_TOOLS_LIB")) return true;
        const char* pre = std::getenv("LD_PRELOAD");
        return pre && (std::strstr(pre, "rocprofiler") || std::strstr(pre, "roctracer"));
    }();
    return attached;
}

// direct dispatches: the tag to time a launch with (-1: not timed), same sampling rule as ev_begin
int direct_tag(const hc_ctx* c, int kind) { return (c->profiling && (c->sample_this_step || kind == hc::kEvPass)) ? kind : -1; }

void ev_end(hc::EventPair* ev, hipStream_t stream) {
    if (ev) HC_HIP(hipEventRecord(ev->b, stream));
}

// ---- panel geometry and launch tiling ---------------------------------------------------------
int env_int(const char* name, int fallback) {
    const char* e = std::getenv(name);  // HC_* variables are tuning knobs for experiments only
    return e ? std::atoi(e) : fallback;
}

void setup_panel_geometry(hc_ctx* c) {
    c->ntiles = (c->Dloc + 15) / 16;
    c->Dpad   = c->ntiles * 16;
    c->ngp    = static_cast<int>((static_cast<long long>(c->S) * c->D + 7) / 8);
    c->mt     = (c->ntiles % 4 == 0) ? 4 : ((c->ntiles % 2 == 0) ? 2 : 1);
    const int want = env_int("HC_CONV_MT", 0);
    if ((want == 1 || want == 2 || want == 4) && c->ntiles % want == 0) c->mt = want;
    c->ngroups = c->ntiles / c->mt;
    // Look-ahead pass: as many row tiles per workgroup as the tile count allows -- every workgroup of a chunk forms the same
    // B operands (ring gathers + interpolation), so fewer, taller workgroups repeat less of that work; it is what bounds
    // the pass once K streams at the HBM rate (C3: 415 / 277 us with 4 / 6 tiles at depth 32 before the per-DoF ring).
    // The design value comes from the UNSHARDED tile count (the chunk length below must not depend on the rows owned).
    auto pick = [](int tiles, int limit) {
        for (int m : {12, 6, 4, 2, 1})
            if (m <= limit && tiles % m == 0) return m;
        return 1;
    };
    const int tiles_full  = (c->D + 15) / 16;
    const int limit       = env_int("HC_BLOCK_MT", 6);
    c->mt_block_design    = pick(tiles_full, limit);
    c->mt_block           = pick(c->ntiles, c->mt_block_design);
    // Short passes of the two-level form stream a few tens of IRF samples only: with the pass's tall workgroups (6 tiles x half a
    // sample = 1.2 MB each at D = 3072) there are fewer workgroups than CUs and each streams for ~50 us; fewer tiles per workgroup
    // give a multiple of the workgroups, each done sooner (the B-operand work they repeat is small here).
    // Tiles per workgroup = the largest that still leaves about two workgroups per CU (a short pass has roughly 48 chunks of half a
    // sample): 2 for a C4/8 shard (24 tiles: 76 -> 68 us per short pass), 6 for C4 on one GPU (192 tiles: 377 us; 422 with 2).
    {
        const int forced = env_int("HC_MINI_MT", 0);
        int m_pick = 1;
        for (int m : {1, 2, 4, 6})
            if (m <= c->mt_block && c->ntiles % m == 0 && 48LL * (c->ntiles / m) >= 2LL * c->num_cus) m_pick = m;
        c->mt_mini = forced > 0 ? pick(c->ntiles, std::min(c->mt_block, forced)) : m_pick;
    }
}

hc::Panel rad_panel(const hc_ctx* c) {
    hc::Panel p;
    p.base   = (c->conv_mode == 1) ? c->dKproc.p : c->dK.p;
    p.ntiles = c->ntiles;
    p.ngp    = c->ngp;
    return p;
}

void choose_conv_config(hc_ctx* c) {
    // plain per-step kernel: 8 workgroups per CU (one resident set; 184.6 us against 187.2 us with twice as many, and
    // finalize_kernel has half as many partials to add); a chunk is a whole number of 8-column groups.  Like the pass
    // below, the chunk length is a function of the column count only (row groups of the UNSHARDED system), so row-sharded
    // contexts add their partial sums in the same order as the unsharded one (bitwise equal results).
    const int target_wgs       = std::max(1, env_int("HC_CONV_TARGET_WGS", 8 * c->num_cus));
    const long long rows_full  = std::max<long long>(1, (((c->D + 15) / 16) + 3) / 4);
    long long nch              = std::max<long long>({1, target_wgs / rows_full, c->num_cus / 2});
    long long gps        = (c->ngp + nch - 1) / nch;
    gps                  = std::max<long long>(16, ((gps + 3) / 4) * 4);  // every wave of the workgroup gets work
    gps                  = std::min<long long>(gps, 512);                 // the chunk's right-hand side is staged in LDS (<= 32 KB)
    c->chunk_gp          = static_cast<int>(gps);
    c->nchunks_rad       = static_cast<int>((c->ngp + gps - 1) / gps);
    // Look-ahead pass.  Its workgroups live long (tens of microseconds) and two fit on a CU, so the launch runs in "rounds" of
    // 2 * CUs workgroups and is fastest when row groups x chunks fills whole rounds: C3 has 4 row groups (6 tiles each), so
    // CUs / 2 = 128 chunks make exactly one round (207 us per pass against 221 us at three rounds and 249 us at one and a
    // half).  The chunk length depends on the column count only -- never on how many rows this context owns -- so that
    // row-sharded contexts add their partial sums in the same order as the unsharded one (bitwise equal results).
    long long bgps;
    const int forced = env_int("HC_BLOCK_CHUNK_GP", 0);
    if (forced > 0) {
        bgps = std::max(8, forced);
        bgps = std::max<long long>(bgps, (c->ngp + 255) / 256);
    } else {
        // row groups of the UNSHARDED system (6 tiles each): few of them (small systems) need more chunks to fill a round.
        // Two workgroups fit on a CU at depth 16, one at depth 32 (twice the accumulators).
        const long long groups_full = std::max<long long>(1, ((c->D + 15) / 16 + c->mt_block_design - 1) / c->mt_block_design);
        const long long slots       = ((c->lookahead > 16 || c->mt_block_design > 6) ? 1LL : 2LL) * c->num_cus;
        const long long nch_target  = std::max<long long>(slots / 4, (slots + groups_full - 1) / groups_full);
        bgps                       = (c->ngp + nch_target - 1) / nch_target;
        bgps                       = std::min<long long>(bgps, std::max<long long>(16, (16LL * c->D) / 8));  // <= 16 IRF samples per chunk
    }
    const long long cap = std::max<long long>(8, (64LL * c->D) / 8);   // bracket table [samples][16] must fit in LDS
    bgps                = std::min(bgps, cap);
    bgps                = std::max<long long>(16, ((bgps + 15) / 16) * 16);  // whole 16-group sub-tiles
    c->chunk_gp_block   = static_cast<int>(bgps);
    c->nchunks_block    = static_cast<int>((c->ngp + bgps - 1) / bgps);
}

void choose_exc_config(hc_ctx* c) {
    if (c->wave_kind != hc::kWaveIrregular || c->L == 0) {
        c->nchunks_ex  = 0;
        c->chunk_gp_ex = 64;
        c->nchunks_ex_block = 0;
        return;
    }
    c->chunk_gp_ex = std::max(4, env_int("HC_EXC_CHUNK_GP", 8));  // short chunks: the excitation side is latency-bound
    c->nchunks_ex  = (c->ngp_ex + c->chunk_gp_ex - 1) / c->chunk_gp_ex;
    c->chunk_gp_ex_block = 16;  // one 16-group sub-tile of the look-ahead kernel per work item (the items are dealt one per workgroup)
    c->nchunks_ex_block  = (c->ngp_ex + c->chunk_gp_ex_block - 1) / c->chunk_gp_ex_block;
}

void alloc_partials(hc_ctx* c) {
    const size_t n = static_cast<size_t>(c->nchunks_rad + c->nchunks_ex) * c->Dpad;
    if (c->d_partials.n < n) c->d_partials.alloc(n);
    // (the short passes of the two-level form use the same buffer: IRF samples s < kScatterSamples -- the planner refuses blocks whose
    // in-block brackets reach further -- in chunks of at least half a sample)
    const size_t nb = static_cast<size_t>(std::max(c->nchunks_block + c->nchunks_ex_block, 2 * hc::kScatterSamples + 2)) * hc::kLookahead * c->Dpad;
    if (c->d_partials_block.n < nb) c->d_partials_block.alloc(nb);
    if (c->d_P.n < static_cast<size_t>(hc::kLookahead) * c->Dpad) c->d_P.alloc(static_cast<size_t>(hc::kLookahead) * c->Dpad);
    if (c->d_E.n < static_cast<size_t>(hc::kLookahead) * c->Dpad) c->d_E.alloc(static_cast<size_t>(hc::kLookahead) * c->Dpad);
    const size_t ny = static_cast<size_t>(hc::kLookahead + 1) * hc::kTermMax * c->Dpad;
    if (c->d_Y.n < ny) c->d_Y.alloc(ny);
    if (c->d_near_partials.n < static_cast<size_t>(16) * c->Dpad) c->d_near_partials.alloc(static_cast<size_t>(16) * c->Dpad);
}

// ---- TaperedDirect ----------------------------------------------------------------------------
// The diagnostics files of EnsureProcessedRIRF (src/hydro_forces.cpp:509-531): per body one rirf_body<b>_summary.csv (b 0-based)
// with the representative channel row 0 / column 0 before and after the processing, rows s < effective_steps, numbers in the
// default ostream format like the reference's `ofs << s << "," << t << "," << before << "," << after`.  Errors are ignored (:529).
void export_taper_csv(hc_ctx* c, int effective) {
    try {
        const int n = std::max(0, std::min(effective, c->S));
        hc::DeviceBuffer<double> d_before, d_after;
        d_before.alloc(std::max(1, n));
        d_after.alloc(std::max(1, n));
        std::vector<double> before(n), after(n);
        hc::Panel raw;
        raw.base   = c->dK.p;
        raw.ntiles = c->ntiles;
        raw.ngp    = c->ngp;
        hc::Panel proc = raw;
        proc.base      = c->dKproc.p;
        for (int bl = 0; bl < c->nloc; ++bl) {
            if (n > 0) {
                hc::launch_extract_series(raw, 6 * bl, 0, c->D, n, d_before.p, c->stream);
                hc::launch_extract_series(proc, 6 * bl, 0, c->D, n, d_after.p, c->stream);
                HC_HIP(hipMemcpyAsync(before.data(), d_before.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HC_HIP(hipMemcpyAsync(after.data(), d_after.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HC_HIP(hipStreamSynchronize(c->stream));
            }
            const std::string base = "rirf_body" + std::to_string(c->b0 + bl) + "_summary.csv";
            const std::string path = c->diagnostics_dir.empty() ? base : (c->diagnostics_dir + "/" + base);
            std::ofstream ofs(path);
            ofs << "step,time,k_before,k_after\n";
            for (int s_ = 0; s_ < n; ++s_) ofs << s_ << "," << c->tau[s_] << "," << before[s_] << "," << after[s_] << "\n";
        }
    } catch (...) {
        (void)hipGetLastError();  // "ignore export errors"
    }
}

void ensure_processed(hc_ctx* c) {
    if (c->conv_mode != 1 || c->proc_ready) return;
    const int steps = c->S;
    int effective   = steps;
    if (c->taper.rirf_end_time > 0.0) {
        require(c->S >= 2, HC_ERR_INVALID, "TaperedDirect truncation needs at least two IRF samples");
        const double dt = c->tau[1] - c->tau[0];
        const int end   = static_cast<int>(std::floor(c->taper.rirf_end_time / dt));
        effective       = std::min(end, steps);
    }
    int tc_index = static_cast<int>(std::floor(c->taper.taper_start_percent * static_cast<double>(effective)));
    int tc_end   = static_cast<int>(std::floor(c->taper.taper_end_percent * static_cast<double>(effective)));
    tc_index     = std::max(0, std::min(tc_index, effective));
    tc_end       = std::max(tc_index, std::min(tc_end, effective));
    HC_HIP(hipDeviceSynchronize());  // once per option change; orders against steps on a caller's stream
    if (c->dKproc.n != c->dK.n) {
        c->dKproc.alloc(c->dK.n);
        HC_HIP(hipMemsetAsync(c->dKproc.p, 0, c->dKproc.n * sizeof(double), c->stream));
    }
    hc::TaperArgs a{};
    a.Kraw.base       = c->dK.p;
    a.Kraw.ntiles     = c->ntiles;
    a.Kraw.ngp        = c->ngp;
    a.Kproc           = c->dKproc.p;
    a.Dloc            = c->Dloc;
    a.D               = c->D;
    a.S               = c->S;
    a.effective_steps = effective;
    a.smoothing       = c->taper.smoothing;
    a.window          = std::max(3, c->taper.window_length);
    a.tc_index        = tc_index;
    a.tc_end          = tc_end;
    a.final_amplitude = c->taper.taper_final_amplitude;
    hc::launch_taper(a, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipStreamSynchronize(c->stream));
    c->proc_ready = true;
    c->plan.valid = false;
    if (c->taper.export_plot_csv) export_taper_csv(c, effective);
}

// ---- the step ---------------------------------------------------------------------------------
struct StepFlags {
    bool hs = true, rad = true, waves = true;
    bool scratch_out = false;  // term-only entry points: outputs go to scratch buffers, the last step's components stay
};

// the excitation-window tests of check_wave_ready as a predicate (for predicted step times)
bool wave_window_ok(const hc_ctx* c, double t) {
    if (c->wave_kind != hc::kWaveIrregular || c->eta_t.size() < 2 || c->ex_groups.empty()) return false;
    const double tmin = c->eta_t.front(), tmax = c->eta_t.back();
    for (const auto& g : c->ex_groups) {
        const double q0 = t - g.tau_front, q1 = t - g.tau_back;
        if (!(tmin <= q0 && q0 <= tmax) || !(tmin <= q1 && q1 <= tmax)) return false;
        if (q0 > tmin && q0 < tmax && q0 <= c->eta_t[1]) return false;
    }
    return true;
}

void check_wave_ready(hc_ctx* c, double t) {
    if (c->wave_nb_arg < c->N)
        throw Error(HC_ERR_RUNTIME, "wave model was created for fewer bodies than the hydro system (force vector shorter than 6N)");
    if (c->wave_kind == hc::kWaveIrregular) {
        // ExcitationConvolution bounds (src/wave_types.cpp:784-794,833-840) and get_lower_index (src/helper.cpp:8-22)
        const double tmin = c->eta_t.front(), tmax = c->eta_t.back();
        for (const auto& g : c->ex_groups) {  // every body's grid (bodies with one grid share a group)
            const double q0 = t - g.tau_front, q1 = t - g.tau_back;
            if (!(tmin <= q0 && q0 <= tmax) || !(tmin <= q1 && q1 <= tmax))
                throw Error(HC_ERR_RUNTIME,
                            "Excitation convolution: trying to find free surface elevation at a time out of bounds from the "
                            "precomputed free surface elevation. Excitation force ignored at this time step.");
            if (q0 > tmin && q0 < tmax && q0 <= c->eta_t[1])
                throw Error(HC_ERR_RUNTIME, "Could not find index for value in free-surface time array (get_lower_index)");
        }
    }
}

// Number of leading IRF samples that can contribute at query time t_query: samples whose t_query - tau_s lies before the
// oldest history sample have no older bracket and contribute nothing (src/hydro_forces.cpp:604-606), so while the history
// is shorter than the IRF window the kernels need not stream the tail of K at all.  Conservative by a small margin.
int live_samples(const hc_ctx* c, double t_query) {
    if (c->times.empty()) return 0;
    const double span   = t_query - c->times.back();
    const double margin = 1e-6 * std::max(1.0, std::fabs(span));
    const auto it       = std::upper_bound(c->tau.begin(), c->tau.end(), span + margin);
    return static_cast<int>(it - c->tau.begin());
}

// find_bracket of hc_kernels.hip on the host copy of the history (times[0] = t is the current sample, times[k] = ring slot
// head - k): the same comparisons and the same divisions on the same doubles, so the weights are the kernel's bit for bit.
// Returns false where the kernel would raise its "not bracketed" flag.
bool host_bracket(const hc_ctx* c, double q, int H, hc::Bracket* out) {
    auto time_at = [&](int k) { return c->times[static_cast<size_t>(k)]; };
    int lo = 0, hi = H - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (time_at(mid + 1) <= q) hi = mid; else lo = mid + 1;
    }
    hc::Bracket b{0.0, 0.0, 0, 0};
    if (lo >= H - 1) {
        *out = b;
        return true;
    }
    const double newer = time_at(lo), older = time_at(lo + 1);
    if (q == older) { b.wo = 1.0; b.wn = 0.0; }
    else if (q == newer) { b.wo = 0.0; b.wn = 1.0; }
    else if (q > older && q < newer) {
        const double td = newer - older;
        b.wo = (td != 0.0) ? ((newer - q) / td) : 0.0;
        b.wn = 1.0 - b.wo;
    } else {
        return false;
    }
    b.off_older = ((c->head - (lo + 1) + c->Hcap) % c->Hcap) * c->D;
    b.off_newer = (lo == 0) ? -1 : ((c->head - lo + c->Hcap) % c->Hcap) * c->D;
    *out = b;
    return true;
}

// Look-ahead bookkeeping at the start of a step: 0 = plain (whole K this step), j = 1..16: the step is block step j of the
// current plan (its time is the predicted one).
int plan_step(hc_ctx* c, double t, int H) {
    auto& pl = c->plan;
    if (pl.cooldown > 0) --pl.cooldown;
    if (H < 2 || c->lookahead <= 0 || !pl.valid) return 0;
    if (pl.j_next <= c->lookahead) {
        // accept the caller's time if it is the predicted one up to accumulated rounding (t += dt in the caller vs
        // t0 + j*dt here); the radiation term is evaluated on the predicted grid, whose interpolation weights then differ
        // from the caller's by <= tol/dt relative, far inside the 1e-6 contract
        const double tol = std::max(1e-9 * pl.dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(t));
        if (std::fabs(t - pl.tgrid[pl.j_next]) <= tol) return pl.j_next++;
    }
    // the caller left the predicted time grid (variable step): drop the block
    pl.valid = false;
    if (++pl.misses >= 2) {
        pl.misses   = 0;
        pl.cooldown = 64;  // irregular stepping: plain steps for a while, then try again
    }
    return 0;
}

// Wide systems (the same switch as the split own-sample kernel: a function of D only, so that row shards plan alike) use the
// two-level form: their scatter launches would re-read (L/2) * K/S bytes from HBM every step.
int plan_sub_block(const hc_ctx* c) {
    const int forced = env_int("HC_SUB_BLOCK", -1);  // tests / tuning runs: 0 = single level, 4 / 8 = sub-block size
    if (forced >= 0) return forced;
    return hc::near_slices_for(c->D) > 1 ? hc::kSubBlock : 0;
}

bool make_plan(hc_ctx* c) { return hc::build_plan(c->plan, c->lookahead, c->times, c->tau, c->width, plan_sub_block(c), hc::near_slices_for(c->D)); }

struct StepViews {
    hc::Panel kex;
    hc::EtaTable ex;
};

StepViews make_views(const hc_ctx* c) {
    StepViews v{};
    const bool irregular = c->wave_kind == hc::kWaveIrregular;
    v.kex.base   = c->d_kex.p;
    v.kex.ntiles = c->ntiles;
    v.kex.ngp    = c->ngp_ex;
    v.ex.L        = c->L;
    v.ex.ex_tau   = c->d_ex_tau.p;
    v.ex.ex_width = c->d_ex_width.p;
    v.ex.eta_t    = c->d_eta_t.p;
    v.ex.eta      = c->d_eta.p;
    v.ex.nt       = c->nt;
    v.ex.eta_dt   = irregular ? c->irr.simulation_dt : 1.0;
    v.ex.eta_t0   = (irregular && !c->eta_t.empty()) ? c->eta_t.front() : 0.0;
    return v;
}

// The look-ahead pass of the plan just made: for the 16 predicted steps, what the samples known now contribute.  It runs as
// the plain pass of a (virtual) step at tgrid[1] whose own sample is zero -- that sample's share is added later by the
// step itself and by its scatter.  Enqueued behind the step that has just been evaluated (its ring push included).
void launch_pass(hc_ctx* c, hipStream_t stream, bool with_exc, bool direct = false) {
    auto& pl = c->plan;
    const int L = c->lookahead;
    const int H = static_cast<int>(c->times.size());
    // history length the virtual step would see after its own push + prune (PruneHistory, src/hydro_forces.cpp:327-340)
    const double hmin = pl.tgrid[1] - (c->tau.empty() ? 0.0 : c->tau.back());
    int Hv = H + 1;
    auto vtime = [&](int k) { return k == 0 ? pl.tgrid[1] : c->times[static_cast<size_t>(k - 1)]; };
    while (Hv > 1 && vtime(Hv - 2) < hmin) --Hv;

    hc::HistoryView hv{};
    hv.state   = c->d_zero_state.p;
    hv.N       = c->N;
    hv.D       = c->D;
    hv.t       = pl.tgrid[1];
    hv.ring_t  = c->d_ring_t.p;
    hv.ring_v  = c->d_ring_v.p;
    hv.ring_vT = c->d_ring_vT.p;
    hv.head    = (c->head + 1) % c->Hcap;  // slot of the virtual sample (never read: time and velocity come from t / state)
    hv.H       = Hv;
    hv.Hcap    = c->Hcap;
    hv.HcapT   = c->HcapT;
    hv.dt_hint = pl.dt;

    const StepViews vw = make_views(c);
    hc::BlockArgs b{};
    b.K                   = rad_panel(c);
    b.F                   = std::min(c->S, live_samples(c, pl.tgrid[L])) * c->D;
    b.depth               = L;
    b.chunk_gp            = c->chunk_gp_block;
    b.nchunks             = std::max(1, ((b.F + 7) / 8 + c->chunk_gp_block - 1) / c->chunk_gp_block);
    b.max_steps_per_chunk = (c->chunk_gp_block * 8) / c->D + 2;
    b.hist                = hv;
    for (int j = 0; j < L; ++j) {
        b.tpred[j]   = pl.tgrid[j + 1];
        b.s_cut[j]   = pl.s_cut[j];
        b.s_defer[j] = pl.s_defer[j];
    }
    b.tau   = c->d_tau.p;
    b.width = c->d_width.p;
    // The excitation force depends on time only, so the pass also evaluates it for the 16 predicted times (extra chunks
    // over Kex in the same launch) -- provided every predicted time passes the window tests a real step would have to pass.
    static const bool exc_in_block = env_int("HC_EXC_IN_BLOCK", 1) != 0;
    bool exc_block = exc_in_block && with_exc && c->wave_kind == hc::kWaveIrregular && c->nchunks_ex_block > 0;
    for (int j = 1; j <= L && exc_block; ++j) exc_block = wave_window_ok(c, pl.tgrid[j]);
    pl.has_exc    = exc_block;
    b.Kex         = vw.kex;
    b.ex          = vw.ex;
    b.chunk_gp_ex = c->chunk_gp_ex_block;
    b.nchunks_ex  = exc_block ? c->nchunks_ex_block : 0;
    b.partials    = c->d_partials_block.p;
    b.Dpad        = c->Dpad;
    b.error_flag  = c->d_err.p;
    b.item_counter = c->d_err.p + 1;
    b.ngroups     = c->ntiles / c->mt_block;
    // algorithmic bytes: summed over the steps of the block, step j's share of K and of the velocity vector from s_cut[j]
    // on ...; what the launch has to move once: the live part of K, Kex and the staged vectors
    double samples = 0.0;
    for (int j = 0; j < L; ++j) samples += std::max(0, b.F / c->D - pl.s_cut[j]);
    const double rad_16 = 8.0 * samples * (static_cast<double>(c->Dloc) * c->D + c->D);
    const double exc_16 = exc_block ? 8.0 * L * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
    c->prof.block_kernel_bytes      = rad_16 + exc_16;
    const double rad_once = 8.0 * (static_cast<double>(c->Dloc) * b.F + b.F);
    const double exc_once = exc_block ? 8.0 * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
    c->prof.block_kernel_bytes_once = rad_once + exc_once;
    if (env_int("HC_DEBUG_PLAN", 0) != 0) {
        std::fprintf(stderr, "[hc] pass t0=%.6f dt=%.17g Hv=%d F/D=%d nchunks=%d exc=%d\n     s_cut:", pl.tgrid[0], pl.dt, Hv, b.F / c->D, b.nchunks, (int)exc_block);
        for (int j = 0; j < L; ++j) std::fprintf(stderr, " %d", pl.s_cut[j]);
        std::fprintf(stderr, "\n     s_defer:");
        for (int j = 0; j < L; ++j) std::fprintf(stderr, " %d", pl.s_defer[j]);
        std::fprintf(stderr, "\n     scat:");
        for (int i = 1; i <= L; ++i) std::fprintf(stderr, " [%d,%d]", pl.scat_lo[i], pl.scat_hi[i]);
        std::fprintf(stderr, "\n");
    }
    const double exc_share = exc_once / std::max(1.0, rad_once + exc_once);
    const hc::ReduceArgs r{c->d_partials_block.p, b.nchunks, b.nchunks_ex, c->Dpad, L, c->d_P.p, c->d_E.p, b.item_counter, 0, 0, 0, 0};
    if (direct) {
        hc::BlockArgs b2;
        const hc::BlockLaunch l = hc::block_launch_config(b, c->mt_block, &b2);
        if (l.nblocks <= 0) return;
        c->dq->dispatch(L == 32 ? c->dk_block32 : c->dk_block16, static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &b2, sizeof b2,
                        direct_tag(c, hc::kEvPass), exc_share);
        c->prof.direct_dispatches += 1;
        c->dq->dispatch(c->dk_reduce, static_cast<uint32_t>(hc::reduce_block_grid(r)), 256, 0, &r, sizeof r);
        c->prof.direct_dispatches += 1;
        return;
    }
    hc::EventPair* ev = ev_begin(c, hc::kEvPass, stream, exc_share);
    hc::launch_conv_block(b, c->mt_block, stream);
    ev_end(ev, stream);
    hc::launch_reduce_block(r, stream);
    c->prof.hip_launches += 2;
}

// The short pass of the two-level form after block step i0 (hc_plan.hpp: MiniPass): what the samples of the sub-block that has just
// ended contribute to the block steps still to come, added to their rows of P.  The same kernel as the pass of the block, over
// the first few IRF samples only, with the bracket table restricted to those samples (BlockArgs::mini_kw) and a chunking of its
// own (half an IRF sample per chunk -- a function of D only, like every other chunk length).
void launch_mini_pass(hc_ctx* c, int i0, hipStream_t stream, bool direct) {
    const auto& pl = c->plan;
    const int L    = c->lookahead;
    const hc::MiniPass mp = hc::mini_pass_setup(pl, L, i0, c->tau);
    if (mp.n_samples <= 0 || mp.n_steps <= 0) return;
    hc::HistoryView hv{};
    hv.state   = c->d_zero_state.p;
    hv.N       = c->N;
    hv.D       = c->D;
    hv.t       = mp.time[0];
    hv.ring_t  = c->d_ring_t.p;
    hv.ring_v  = c->d_ring_v.p;
    hv.ring_vT = c->d_ring_vT.p;
    hv.head    = (c->head + 1) % c->Hcap;  // slot of the not-yet-known sample of step i0 + 1
    hv.H       = static_cast<int>(c->times.size()) + 1;
    hv.Hcap    = c->Hcap;
    hv.HcapT   = c->HcapT;
    hv.dt_hint = pl.dt;
    hc::BlockArgs b{};
    b.K        = rad_panel(c);
    b.F        = mp.n_samples * c->D;
    b.depth    = L;
    b.chunk_gp = std::max(16, (((c->D + 7) / 8 / 2 + 15) / 16) * 16);
    b.nchunks  = std::max(1, ((b.F + 7) / 8 + b.chunk_gp - 1) / b.chunk_gp);
    b.max_steps_per_chunk = (b.chunk_gp * 8) / c->D + 2;
    b.hist     = hv;
    for (int j = 0; j < L; ++j) {
        b.tpred[j]   = mp.tpred[j];
        b.s_cut[j]   = mp.s_cut[j];
        b.s_defer[j] = mp.s_defer[j];
    }
    b.tau          = c->d_tau.p;
    b.width        = c->d_width.p;
    b.Kex          = make_views(c).kex;
    b.ex           = make_views(c).ex;
    b.chunk_gp_ex  = c->chunk_gp_ex_block;
    b.nchunks_ex   = 0;
    b.partials     = c->d_partials_block.p;
    b.Dpad         = c->Dpad;
    b.error_flag   = c->d_err.p;
    b.item_counter = c->d_err.p + 1;
    b.ngroups      = c->ntiles / c->mt_mini;
    b.mini_kw      = mp.kw;
    b.mini_steps   = mp.n_steps;
    for (int k = 0; k <= mp.kw + 1; ++k) b.mini_time[k] = mp.time[k];
    require(static_cast<size_t>(b.nchunks) * L * c->Dpad <= c->d_partials_block.n, HC_ERR_RUNTIME, "short pass: partials buffer too small");
    hc::ReduceArgs r{c->d_partials_block.p, b.nchunks, 0, c->Dpad, L, c->d_P.p, c->d_E.p, b.item_counter, 1, i0, mp.n_steps, 0};
    if (direct) {
        hc::BlockArgs b2;
        const hc::BlockLaunch l = hc::block_launch_config(b, c->mt_mini, &b2);
        if (l.nblocks <= 0) return;
        c->dq->dispatch(L == 32 ? c->dk_mini32 : c->dk_mini16, static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &b2, sizeof b2,
                        direct_tag(c, hc::kEvMiniPass));
        c->dq->dispatch(c->dk_reduce, static_cast<uint32_t>(hc::reduce_block_grid(r)), 256, 0, &r, sizeof r);
        c->prof.direct_dispatches += 2;
        return;
    }
    hc::EventPair* ev = ev_begin(c, hc::kEvMiniPass, stream);
    hc::launch_conv_block(b, c->mt_mini, stream);
    ev_end(ev, stream);
    hc::launch_reduce_block(r, stream);
    c->prof.hip_launches += 2;
}

// Second half of a step's enqueue: the work LATER steps need (the scatter of this step's sample inside a look-ahead block, or
// the plan and the pass of the next block).  Off the caller's critical path: it is enqueued behind the step kernel and runs
// while the host is away.  On a caller's stream (hc_step_device) whose owner waits for every step it goes to the context's
// own stream behind an event, so that whatever the caller enqueues next on its stream -- the all-gather of the force rows in
// a multi-GPU run -- follows the step kernel directly.  hc_step_multi calls it after the step kernels of ALL shard contexts
// have been handed to their GPUs.
void enqueue_tail(hc_ctx* c) {
    if (!c->tail.pending) return;
    c->tail.pending        = false;
    const bool block       = c->tail.block, direct = c->tail.direct, caller_waits = c->tail.caller_waits;
    const int m            = c->tail.m, H = c->tail.H;
    const hipStream_t stream = c->tail.stream;
    const bool scatter_now = block && m < c->lookahead && c->plan.scat_hi[m] >= c->plan.scat_lo[m];
    const bool plan_now    = (!block || m == c->lookahead) && H >= 2;
    hipStream_t bs         = stream;
    auto to_background = [&]() {
        if (caller_waits && bs == stream) {
            bs = c->stream;
            HC_HIP(hipEventRecord(c->ev_fin, stream));
            HC_HIP(hipStreamWaitEvent(bs, c->ev_fin, 0));
        }
    };
    if (scatter_now) {
        to_background();
        const auto& pl = c->plan;
        hc::ScatterArgs sa{};
        sa.K     = rad_panel(c);
        sa.D     = c->D;
        sa.Dpad  = c->Dpad;
        sa.s_lo  = pl.scat_lo[m];
        sa.ns    = pl.scat_hi[m] - pl.scat_lo[m] + 1;
        sa.v     = c->d_ring_v.p + static_cast<size_t>(c->head) * c->D;  // this step's sample, pushed by finalize_kernel
        sa.width = c->d_width.p;
        sa.Y     = c->d_Y.p;
        for (int si = 0; si < sa.ns; ++si) {
            const int s_ = sa.s_lo + si;
            sa.n_tgt[si] = pl.n_tgt[m][s_];
            for (int t = 0; t < pl.n_tgt[m][s_]; ++t) {
                sa.tgt_off[si][t]  = (pl.tgt_step[m][s_][t] * hc::kTermMax + pl.tgt_k[m][s_][t]) * c->Dpad;
                sa.tgt_coef[si][t] = pl.tgt_coef[m][s_][t];
            }
        }
        if (direct) {
            const hc::ScatterLaunch l = hc::scatter_launch_config(sa);
            c->dq->dispatch(c->dk_scatter, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &sa, sizeof sa, direct_tag(c, hc::kEvScatter));
            c->prof.direct_dispatches += 1;
        } else {
            hc::EventPair* ev = ev_begin(c, hc::kEvScatter, bs);
            hc::launch_scatter(sa, bs);
            c->prof.hip_launches += 1;
            ev_end(ev, bs);
        }
    } else if (block && c->plan.sub > 0 && m < c->lookahead && m % c->plan.sub == 0 && c->plan.mini_s_hi[m] >= 0) {
        to_background();
        launch_mini_pass(c, m, bs, direct);  // two-level form: the sub-block that ends here -> the block steps still to come
    } else if (plan_now) {
        if (block) c->plan.misses = 0;  // a block was consumed completely
        if (make_plan(c)) {
            to_background();
            launch_pass(c, bs, c->tail.waves, direct);
        }
    }
    if (bs != stream) {
        HC_HIP(hipEventRecord(c->ev_bg, bs));
        c->bg_pending = true;
    }
    HC_HIP(hipGetLastError());
}

// Enqueue the kernels of one evaluation at time t.  d_state: device-visible pointer to the 12N state.  d_user_out
// (device) and host_tagged (mapped pinned granules) may be null.
// defer_tail: the caller enqueues the work later steps need itself (enqueue_tail) -- hc_step_multi, after all shard contexts
// have their step kernels on the way.
void enqueue_step(hc_ctx* c, double t, const double* d_state, double* d_user_out, hipStream_t stream, StepFlags f,
                  unsigned long long* host_tagged = nullptr, unsigned long long seq = 0, bool defer_tail = false) {
    require(!c->tail.pending, HC_ERR_INVALID, "a step begun with hc_step_begin has not been completed (hc_step_end)");
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    const bool irregular = c->wave_kind == hc::kWaveIrregular;
    if (f.waves) check_wave_ready(c, t);
    profile_begin_step(c);
    int H = 0, m = 0;
    if (f.rad) {
        ensure_processed(c);
        H = history_push(c, t);
        m = plan_step(c, t, H);
    }
    const bool run_rad = f.rad && H >= 2;  // "Nothing to convolve with if we don't yet have at least 2 time points" (:580)
    const bool run_exc = f.waves && irregular;
    const StepViews vw = make_views(c);
    const bool block   = run_rad && m > 0;
    // A caller's stream that is idle now belongs to a caller that waits for every step (the force exchange of a row-sharded
    // array): the work later steps need then goes to the context's own stream, see below.  A caller that runs ahead of the GPU
    // (stream still busy) gets everything on its stream in order -- the two event hops per step would only slow it down.
    bool caller_waits = false;
    if (stream != c->stream && f.rad && c->lookahead > 0) {
        if (c->busy_caller_steps > 0) {
            --c->busy_caller_steps;  // found busy a moment ago: do not pay for the query on every step of a caller that runs ahead
        } else {
            const hipError_t q = hipStreamQuery(stream);
            caller_waits       = q == hipSuccess;
            if (q != hipSuccess) {
                (void)hipGetLastError();
                c->busy_caller_steps = 15;
            }
        }
    }
    if (c->have_last_stream && c->last_stream != stream) {
        // The steps of a context normally stay on one stream (the velocity ring is updated in stream order).  When they
        // move -- hc_step after hc_step_device on a caller's stream, or the reverse -- this step is ordered behind the
        // previous one with an event.  A caller's stream may have been destroyed since (after a synchronise): then there is
        // nothing left to wait for.
        if (hipEventRecord(c->ev_fin, c->last_stream) == hipSuccess) HC_HIP(hipStreamWaitEvent(stream, c->ev_fin, 0));
        else (void)hipGetLastError();
    }
    c->last_stream      = stream;
    c->have_last_stream = true;
    if (c->bg_pending) {
        // the scatter / pass of the previous step ran on the context's own stream (see below): this step's kernels need them
        if (stream != c->stream) HC_HIP(hipStreamWaitEvent(stream, c->ev_bg, 0));
        c->bg_pending = false;
    }
    const double* P_row = block ? c->d_P.p + static_cast<size_t>(m - 1) * c->Dpad : nullptr;
    const double* E_row = (block && run_exc && c->plan.has_exc) ? c->d_E.p + static_cast<size_t>(m - 1) * c->Dpad : nullptr;

    // plain step: all live columns of K, plus the excitation chunks unless a pass has left the excitation force
    int nchunks_rad = 0, nchunks_ex = (run_exc && !E_row) ? c->nchunks_ex : 0;
    // Where this step's kernels go: the direct queue (hc_direct.hpp) when the step comes from hc_step and needs no plain
    // convolution launch -- the steady state of a look-ahead run -- else the HIP stream.  Nothing orders the two against each
    // other on the device, so the side that was used last is drained at a switch.
    const bool direct = c->direct_ready && host_tagged && stream == c->stream && !f.scratch_out &&
                        (c->dk_step.ok() || !((run_rad && !block) || nchunks_ex > 0)) &&
                        !(c->profiling && profiling_tool_attached());  // the library's own timings under a tool: HIP events
    if (direct && c->path != 2) {
        // switching from HIP launches to the direct queue: everything the HIP side still runs must have finished.  A preceding
        // step on a caller's stream (hc_step_device) has been ordered in front of c->stream by the ev_fin wait above, and what it
        // left on the context's own stream (ev_bg) runs there too, so draining c->stream covers both.
        HC_HIP(hipStreamSynchronize(c->stream));
        c->bg_pending = false;
        c->path       = 2;
    } else if (!direct) {
        quiesce_direct(c);
        c->path = 1;
    }
    if ((run_rad && !block) || nchunks_ex > 0) {
        hc::HistoryView hv{};
        hv.state   = d_state;
        hv.N       = c->N;
        hv.D       = c->D;
        hv.t       = t;
        hv.ring_t  = c->d_ring_t.p;
        hv.ring_v  = c->d_ring_v.p;
        hv.head    = c->head;
        hv.H       = H;
        hv.Hcap    = c->Hcap;
        hv.HcapT   = c->HcapT;
        hv.dt_hint = (H >= 2 && c->times[0] > c->times[1]) ? (c->times[0] - c->times[1]) : 1.0;
        hc::StepArgs a{};
        a.K       = rad_panel(c);
        a.F_limit = (run_rad && !block) ? std::min(c->S, live_samples(c, t)) * c->D : 0;
        a.chunk_gp = c->chunk_gp;
        nchunks_rad           = ((a.F_limit + 7) / 8 + a.chunk_gp - 1) / a.chunk_gp;
        a.nchunks_rad         = nchunks_rad;
        a.max_steps_per_chunk = (a.chunk_gp * 8) / c->D + 2;
        a.rhs_capacity        = 8 * std::max(a.chunk_gp, c->chunk_gp_ex);
        a.hist                = hv;
        a.tau                 = c->d_tau.p;
        a.width               = c->d_width.p;
        a.Kex                 = vw.kex;
        a.ex                  = vw.ex;
        a.chunk_gp_ex         = c->chunk_gp_ex;
        a.nchunks_ex          = nchunks_ex;
        a.partials            = c->d_partials.p;
        a.Dpad                = c->Dpad;
        a.ngroups             = c->ngroups;
        a.error_flag          = c->d_err.p;
        const double rad_b = 8.0 * (static_cast<double>(c->Dloc) * a.F_limit + a.F_limit);
        const double exc_b = nchunks_ex > 0 ? 8.0 * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
        const int kind     = nchunks_rad > 0 ? hc::kEvConvPlain : hc::kEvConvExc;
        const double share = exc_b / std::max(1.0, rad_b + exc_b);
        if (direct) {
            const hc::StepLaunch l = hc::step_launch_config(a, c->mt);
            if (l.nblocks > 0) {
                c->dq->dispatch(c->dk_step, static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &a, sizeof a, direct_tag(c, kind), share);
                c->prof.direct_dispatches += 1;
            }
        } else {
            hc::EventPair* ev = ev_begin(c, kind, stream, share);
            hc::launch_conv_step(a, c->mt, stream);
            c->prof.hip_launches += 1;
            ev_end(ev, stream);
        }
    }

    hc::FinalizeArgs z{};
    z.partials    = c->d_partials.p;
    z.nchunks_rad = nchunks_rad;
    z.nchunks_ex  = nchunks_ex;
    z.P           = P_row;
    z.E           = E_row;
    if (block) {
        const auto& pl = c->plan;
        z.nearK     = rad_panel(c);
        z.ring_v_ro = c->d_ring_v.p;
        for (int e = 0; e < pl.n_own[m]; ++e) {
            hc::NearEntry& ne = z.near[z.n_near++];
            ne   = hc::NearEntry{};
            ne.s = pl.own_s[m][e];
            ne.a = pl.own_a[m][e];
        }
        const int sd = pl.s_defer[m - 1];
        if (sd >= 0) {
            // the IRF sample the pass left to this step: its whole bracket, with the caller's time and the history as it is
            hc::Bracket br{};
            if (host_bracket(c, t - c->tau[sd], H, &br) && (br.wo != 0.0 || br.wn != 0.0)) {
                hc::NearEntry& ne = z.near[z.n_near++];
                ne       = hc::NearEntry{};
                ne.s     = sd;
                ne.off_b = br.off_older;
                ne.b     = br.wo * c->width[sd];
                if (br.off_newer < 0) ne.a = br.wn * c->width[sd];
                else {
                    ne.off_c = br.off_newer;
                    ne.c     = br.wn * c->width[sd];
                }
            }
        }
        z.n_terms = pl.n_terms[m];
        z.Yc      = c->d_Y.p + static_cast<size_t>(m) * hc::kTermMax * c->Dpad;
    }
    static const bool dbg = env_int("HC_DEBUG_PLAN", 0) != 0;
    if (dbg) {
        std::fprintf(stderr, "[hc] t=%.6f H=%d m=%d n_near=%d n_terms=%d sd=%d nchunks_rad=%d nchunks_ex=%d head=%d\n", t, H, m, z.n_near, z.n_terms,
                     block ? c->plan.s_defer[m - 1] : -2, nchunks_rad, nchunks_ex, c->head);
        for (int e = 0; e < z.n_near; ++e)
            std::fprintf(stderr, "     near s=%d a=%.6g b=%.6g c=%.6g offb=%d offc=%d\n", z.near[e].s, z.near[e].a, z.near[e].b, z.near[e].c, z.near[e].off_b, z.near[e].off_c);
    }
    z.host_tagged = host_tagged;
    z.seq         = seq;
    z.Dloc        = c->Dloc;
    z.Dpad        = c->Dpad;
    z.N           = c->N;
    z.b0          = c->b0;
    z.state       = d_state;
    z.lin         = c->d_lin.p;
    z.cg          = c->d_cg.p;
    z.cb_m_cg     = c->d_cbmcg.p;
    z.disp_vol    = c->d_vol.p;
    z.rho         = c->rho;
    z.gx          = c->gsys[0];
    z.gy          = c->gsys[1];
    z.gz          = c->gsys[2];
    z.wave_mode   = c->wave_kind;
    z.reg_mag     = c->d_reg_mag.p;
    for (int i = 0; i < 6; ++i) z.reg_phase[i] = c->reg_phase.size() >= 6 ? c->reg_phase[i] : 0.0;
    z.reg_amplitude = c->reg_amp;
    z.reg_omega     = c->reg_omega;
    z.spec_nf       = c->nf;
    z.spec_mag      = c->d_spec_mag.p;
    z.spec_phase    = c->d_spec_phase.p;
    z.spec_amp      = c->d_spec_amp.p;
    z.spec_omega    = c->d_spec_omega.p;
    z.spec_phi      = c->d_spec_phi.p;
    z.spec_ramp     = c->irr.ramp_duration;
    z.t             = t;
    z.do_hs         = f.hs;
    z.do_rad        = run_rad;
    z.do_waves      = f.waves;
    double* out4    = f.scratch_out ? c->d_scratch.p : nullptr;
    z.hs            = out4 ? out4 : c->d_hs.p;
    z.rad           = out4 ? out4 + c->Dloc : c->d_rad.p;
    z.waves         = out4 ? out4 + 2 * c->Dloc : c->d_waves.p;
    z.total         = out4 ? out4 + 3 * c->Dloc : c->d_total.p;
    z.user_out      = d_user_out;
    z.do_push       = f.rad ? 1 : 0;
    z.head          = c->head;
    z.D             = c->D;
    z.ring_t        = c->d_ring_t.p;
    z.ring_v        = c->d_ring_v.p;
    z.ring_vT       = c->d_ring_vT.p;
    z.Hcap          = c->Hcap;
    z.HcapT         = c->HcapT;
    if (z.n_near > 0 && hc::near_slices_for(c->D) > 1) {
        // wide system: the own-sample part is split over column slices by a kernel of its own (hundreds of workgroups instead of one
        // per row tile); the step kernel adds the slice partials
        hc::NearArgs na{};
        na.K      = z.nearK;
        na.D      = c->D;
        na.Dpad   = c->Dpad;
        na.N      = c->N;
        na.n_near = z.n_near;
        for (int e = 0; e < z.n_near; ++e) na.near[e] = z.near[e];
        na.state    = d_state;
        na.ring_v   = c->d_ring_v.p;
        na.partials = c->d_near_partials.p;
        if (direct) {
            const hc::NearLaunch l = hc::near_launch_config(na);
            c->dq->dispatch(c->dk_near, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &na, sizeof na, direct_tag(c, hc::kEvStep));
            c->prof.direct_dispatches += 1;
        } else {
            hc::EventPair* ev = ev_begin(c, hc::kEvStep, stream);
            hc::launch_near_split(na, stream);
            ev_end(ev, stream);
            c->prof.hip_launches += 1;
            (void)hc::near_launch_config(na);
        }
        z.near_partials = c->d_near_partials.p;
        z.n_near_slices = na.n_slices;
        z.n_near        = 0;
    }
    if (direct) {
        const hc::FinalizeLaunch l = hc::finalize_launch_config(z);
        c->dq->dispatch(c->dk_finalize, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &z, sizeof z, direct_tag(c, hc::kEvStep));
        c->prof.direct_dispatches += 1;
    } else {
        hc::EventPair* ev = ev_begin(c, hc::kEvStep, stream);
        hc::launch_finalize(z, stream);
        c->prof.hip_launches += 1;
        ev_end(ev, stream);
    }

    // ---- off the caller's critical path: what later steps need from this one (enqueue_tail) ----
    c->tail              = hc::StepTail{};
    c->tail.pending      = f.rad && c->lookahead > 0;
    c->tail.rad          = f.rad;
    c->tail.waves        = f.waves;
    c->tail.block        = block;
    c->tail.direct       = direct;
    c->tail.caller_waits = caller_waits;
    c->tail.m            = m;
    c->tail.H            = H;
    c->tail.stream       = stream;
    if (!defer_tail) enqueue_tail(c);
    HC_HIP(hipGetLastError());

    if (f.hs) c->prof.hydrostatics_calls++;
    if (f.rad) c->prof.radiation_calls++;
    if (f.waves) c->prof.waves_calls++;
}

// CreateSpectrum (src/wave_types.cpp:643-676): frequencies, PM/JONSWAP densities, trapezoid widths, mt19937 phases,
// wavenumbers, plus the component amplitude sqrt(2 S df) and angular frequency of GetEtaIrregular (:39-40).
struct Spectrum {
    int nf = 0;
    std::vector<double> f, S, df, phase, k, amp, omega;
};

Spectrum build_spectrum(const hc_ctx* c, const hc_irregular_wave_params& p) {
    Spectrum sp;
    if (p.nfrequencies == 0) {
        const double df = 1.0 / p.simulation_duration;
        sp.nf           = static_cast<int>(std::ceil((p.frequency_max - p.frequency_min) / df));
    } else {
        sp.nf = static_cast<int>(p.nfrequencies);
    }
    require(sp.nf >= 1, HC_ERR_INVALID, "no wave components");
    sp.f = hc::linspaced(sp.nf, p.frequency_min, p.frequency_max);
    std::sort(sp.f.begin(), sp.f.end());  // PiersonMoskowitzSpectrumHz sorts its argument in place (:681)
    sp.S     = hc::jonswap_spectrum_hz(sp.f, p.wave_height, p.wave_period, p.peak_enhancement_factor, p.is_normalized != 0);
    sp.df    = hc::trapezoid_widths(sp.f);
    sp.phase = hc::random_phases(sp.nf, p.seed);
    sp.k.resize(sp.nf);
    sp.amp.resize(sp.nf);
    sp.omega.resize(sp.nf);
    const double two_pi = 2 * M_PI;
    for (int i = 0; i < sp.nf; ++i) {
        sp.k[i]     = hc::wave_number(two_pi * sp.f[i], c->depth, c->g);
        sp.amp[i]   = std::sqrt(2 * sp.S[i] * sp.df[i]);
        sp.omega[i] = two_pi * sp.f[i];
    }
    return sp;
}

void check_device_flag(hc_ctx* c) {
    HC_HIP(hipMemcpyAsync(c->h_err.p, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    if (*c->h_err.p != 0) {
        const int code = *c->h_err.p;
        HC_HIP(hipMemsetAsync(c->d_err.p, 0, sizeof(int), c->stream));
        if (code == 1) throw Error(HC_ERR_RUNTIME, "Radiation convolution: interpolation error; query_time not bracketed by history.");
        throw Error(HC_ERR_RUNTIME, "Excitation convolution: tau value not bracketed by the free-surface table");
    }
}

void stage_state(hc_ctx* c, const double* pos, const double* rpy, const double* linvel, const double* angvel) {
    HC_HIP(hipStreamSynchronize(c->stream));  // kernels of the last hc_step may still be reading the pinned state (zero-copy)
    const int n3 = 3 * c->N;
    double* h    = c->h_state.p;
    const double* src[4] = {pos, rpy, linvel, angvel};
    for (int k = 0; k < 4; ++k) {
        if (src[k]) std::memcpy(h + k * n3, src[k], n3 * sizeof(double));
        else std::memset(h + k * n3, 0, n3 * sizeof(double));
    }
    HC_HIP(hipMemcpyAsync(c->d_state.p, h, 4 * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
}

std::string library_dir() {
    Dl_info info;
    std::string dir = ".";
    if (dladdr(reinterpret_cast<void*>(&hc_version), &info) && info.dli_fname) {
        std::string full(info.dli_fname);
        const size_t slash = full.find_last_of('/');
        if (slash != std::string::npos) dir = full.substr(0, slash);
    }
    return dir;
}

// What the host RE-writes through the BAR must be what the next dispatch reads.  The dispatches carry agent-scope acquire fences
// only, so this rests on the GPU not keeping stale copies of fine-grained device memory across kernels -- for the state buffer
// (two halves, rewritten every second step) and for the kernarg ring (64 slots per lane, reused every 64 dispatches).  Checked
// on the given lane: 2 x 64 + 3 one-row added-mass products whose inputs (w, R_in: the SAME two BAR words every time) and
// whose argument slot change from dispatch to dispatch; every result must be the one of the values written last.  A stale read
// fails the test (reason in c->direct_why) and the caller keeps using HIP launches.  *abandon: a dispatch never completed.
bool direct_selftest_rewrites(hc_ctx* c, hc::DirectQueue* q, int lane, bool* abandon) {
    *abandon = false;
    const double one = 1.0;
    HC_HIP(hipMemcpy(c->d_selftest.p, &one, sizeof one, hipMemcpyHostToDevice));  // the 1 x 1 "matrix"
    volatile unsigned long long* tag = c->h_tag_selftest.p;
    tag[0] = tag[1] = 0;
    for (int i = 0; i < 2 * 64 + 3; ++i) {
        const double wv = 3.0 + i, rv = 0.25 * (i + 1) + lane, cv = 0.5 + 0.125 * (i % 7);
        c->bar_selftest.p[0] = wv;
        c->bar_selftest.p[1] = rv;
        _mm_sfence();
        const unsigned long long sq = 0xABC000ull + static_cast<unsigned long long>(lane) * 1000 + i;
        hc::AddedMassArgs a{c->d_selftest.p, 1, 1, c->bar_selftest.p, c->bar_selftest.p + 1, cv, c->h_tag_selftest.dp, sq};
        q->dispatch(c->dk_added_mass, 1, 256, 0, &a, sizeof a, -1, 0.0, lane);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived  = false;
        for (unsigned long long spins = 0;; ++spins) {
            if (tag[1] == sq) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((spins & 0xFFF) == 0xFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        }
        if (!arrived) {
            // either the dispatch hangs or it ran with the arguments of the slot's previous use (an old sequence number)
            if (q->drain(2.0, lane)) {
                c->direct_why = "self-test of the direct dispatch: a re-used kernel-argument slot was read stale";
            } else {
                c->direct_why = "self-test of the direct dispatch timed out";
                *abandon      = true;
            }
            return false;
        }
        const unsigned long long bits = tag[0];
        double got;
        std::memcpy(&got, &bits, sizeof got);
        if (got != rv + cv * wv) {
            c->direct_why = "self-test of the direct dispatch: memory re-written through the PCIe BAR was read stale";
            (void)q->drain(2.0, lane);
            return false;
        }
    }
    if (!q->drain(2.0, lane)) {
        c->direct_why = "self-test of the direct dispatch timed out";
        *abandon      = true;
        return false;
    }
    return true;
}

// Direct AQL dispatch for the synchronous step path (hc_direct.hpp).  Optional: when anything it needs is missing -- the code
// object next to the library, a host-addressable BAR, one of the kernels of this configuration -- the HIP launches stay in use
// and hc_last_error-style diagnostics keep the reason (HC_DEBUG_PLAN prints it).  Still the GPU path either way.
void setup_direct(hc_ctx* c) {
    c->direct_ready = false;
    if (env_int("HC_DIRECT", 1) == 0) { c->direct_why = "disabled by HC_DIRECT=0"; return; }
    if (std::getenv("HC_BLOCK_V32")) { c->direct_why = "HC_BLOCK_V32 selects a tuning variant of the pass"; return; }
    if (!c->bar_state.host_ok || !c->bar_am.host_ok || !c->bar_selftest.host_ok) { c->direct_why = "the device's memory is not host-addressable"; return; }
    std::unique_ptr<hc::DirectQueue> q(new hc::DirectQueue);
    std::string why;
    if (!q->init(c->device, library_dir() + "/hc_kernels.co", &why)) { c->direct_why = why; return; }
    c->dk_finalize = q->find("finalize_kernelILi4EEEv");
    c->dk_scatter  = q->find("scatter_kernelE");
    c->dk_near     = q->find("near_split_kernelE");
    c->dk_reduce   = q->find("reduce_block_kernelE");
    c->dk_added_mass = q->find("added_mass_mv_tagged_kernelE");  // optional: hc_added_mass_mv falls back to a HIP launch
    {   // the plain per-step convolution of this context's tiling; optional: without it plain steps go through HIP launches
        hc::StepArgs a{};
        a.ngroups = 1;
        const hc::StepLaunch l = hc::step_launch_config(a, c->mt);
        char frag[64];
        std::snprintf(frag, sizeof frag, "conv_step_kernelILi%dELi%dEEEv", l.MT, l.U);
        c->dk_step = q->find(frag);
        if (c->dk_step.kernarg != sizeof(hc::StepArgs) || c->dk_step.priv != 0) c->dk_step = hc::DirectKernel{};
    }
    for (int depth : {16, 32}) {
        hc::BlockArgs a{}, b{};
        a.depth = depth;
        a.ngroups = 1;
        const hc::BlockLaunch l = hc::block_launch_config(a, c->mt_block, &b);
        char frag[96];
        std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", l.MT, l.R, l.NB, l.WPS);
        (depth == 16 ? c->dk_block16 : c->dk_block32) = q->find(frag);
        const hc::BlockLaunch lm = hc::block_launch_config(a, c->mt_mini, &b);  // the short passes' variant (fewer tiles per workgroup)
        std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", lm.MT, lm.R, lm.NB, lm.WPS);
        (depth == 16 ? c->dk_mini16 : c->dk_mini32) = q->find(frag);
    }
    if (!c->dk_mini16.ok() || !c->dk_mini32.ok() || c->dk_mini16.priv || c->dk_mini32.priv || c->dk_mini16.kernarg != sizeof(hc::BlockArgs) ||
        c->dk_mini32.kernarg != sizeof(hc::BlockArgs)) {
        c->direct_why = "the short-pass variant of the pass kernel is missing from hc_kernels.co";
        return;
    }
    if (!c->dk_finalize.ok() || !c->dk_scatter.ok() || !c->dk_reduce.ok() || !c->dk_block16.ok() || !c->dk_block32.ok() || !c->dk_near.ok()) {
        c->direct_why = "a kernel of this configuration is missing from hc_kernels.co";
        return;
    }
    // the code object must be the one built with this library: its kernels take exactly these argument blocks
    if (c->dk_finalize.kernarg != sizeof(hc::FinalizeArgs) || c->dk_scatter.kernarg != sizeof(hc::ScatterArgs) ||
        c->dk_block16.kernarg != sizeof(hc::BlockArgs) || c->dk_block32.kernarg != sizeof(hc::BlockArgs) || c->dk_reduce.kernarg != sizeof(hc::ReduceArgs) ||
        c->dk_near.kernarg != sizeof(hc::NearArgs)) {
        c->direct_why = "hc_kernels.co was not built from the same sources as this library (argument block sizes differ)";
        return;
    }
    if (c->dk_finalize.priv || c->dk_scatter.priv || c->dk_reduce.priv || c->dk_block16.priv || c->dk_block32.priv || c->dk_near.priv) {
        c->direct_why = "a kernel needs scratch memory";
        return;
    }
    if (!c->dk_added_mass.ok() || c->dk_added_mass.kernarg != sizeof(hc::AddedMassArgs) || c->dk_added_mass.priv != 0) {
        c->direct_why = "added_mass_mv_tagged_kernel is missing from hc_kernels.co";
        return;
    }
    static_assert(sizeof(hc::NearArgs) <= hc::DirectQueue::kSlotBytes, "an argument block does not fit a kernarg slot of the direct queue");
    static_assert(sizeof(hc::ScatterArgs) <= hc::DirectQueue::kSlotBytes && sizeof(hc::FinalizeArgs) <= hc::DirectQueue::kSlotBytes &&
                      sizeof(hc::BlockArgs) <= hc::DirectQueue::kSlotBytes && sizeof(hc::StepArgs) <= hc::DirectQueue::kSlotBytes,
                  "an argument block does not fit a kernarg slot of the direct queue");
    // self-test 1: one dispatch of the reduction kernel with nothing to add must clear a marked word of P
    const double mark = 1.0;
    HC_HIP(hipMemcpy(c->d_P.p, &mark, sizeof mark, hipMemcpyHostToDevice));
    hc::ReduceArgs r{c->d_partials_block.p, 0, 0, c->Dpad, 1, c->d_P.p, c->d_E.p, c->d_err.p + 1, 0, 0, 0, 0};
    q->dispatch(c->dk_reduce, static_cast<uint32_t>((c->Dpad + 15) / 16), 256, 0, &r, sizeof r);
    if (!q->drain(2.0)) {
        c->direct_why = "self-test of the direct dispatch timed out";
        (void)q.release();  // a queue with a dispatch that never completed is left alone
        return;
    }
    double back = -1.0;
    HC_HIP(hipMemcpy(&back, c->d_P.p, sizeof back, hipMemcpyDeviceToHost));
    if (back != 0.0) { c->direct_why = "self-test of the direct dispatch failed"; return; }
    // self-test 2 (direct_selftest_rewrites): what the host RE-writes through the BAR must be what the next dispatch reads; lane 0
    // here, lane 1 when hc_added_mass_mv first uses it
    {
        bool abandon = false;
        if (!direct_selftest_rewrites(c, q.get(), 0, &abandon)) {
            if (abandon) (void)q.release();  // a queue with a dispatch that never completed is left alone
            return;
        }
    }
    c->dq           = q.release();
    c->direct_ready = true;
    c->direct_why.clear();
    if (env_int("HC_DEBUG_PLAN", 0) != 0) std::fprintf(stderr, "[hc] direct AQL dispatch in use for the step path\n");
}

}  // namespace

// =================================================================================================
extern "C" {

const char* hc_version(void) { return kVersion; }

int hc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* hc_last_error(const hc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hc_create_sharded(int num_bodies, int body_begin, int body_end, int device_id, hc_ctx** out) {
    if (!out) return HC_ERR_INVALID;
    *out = nullptr;
    try {
        require(num_bodies > 0, HC_ERR_INVALID, "num_bodies must be positive");
        require(body_begin >= 0 && body_begin < body_end && body_end <= num_bodies, HC_ERR_INVALID, "invalid body shard");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Error(HC_ERR_DEVICE, "no HIP device available: the hydro-force path has no CPU fallback");
        require(device_id >= 0 && device_id < ndev, HC_ERR_DEVICE, "device_id out of range");
        HC_HIP(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HC_HIP(hipGetDeviceProperties(&prop, device_id));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw Error(HC_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        std::unique_ptr<hc_ctx> c(new hc_ctx);
        c->N      = num_bodies;
        c->b0     = body_begin;
        c->b1     = body_end;
        c->nloc   = body_end - body_begin;
        c->D      = 6 * num_bodies;
        c->Dloc   = 6 * c->nloc;
        c->device  = device_id;
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->bodies.resize(num_bodies);
        HC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HC_HIP(hipStreamCreateWithFlags(&c->stream_am, hipStreamNonBlocking));
        HC_HIP(hipEventCreateWithFlags(&c->ev_fin, hipEventDisableTiming));
        HC_HIP(hipEventCreateWithFlags(&c->ev_bg, hipEventDisableTiming));
        hc_tapered_direct_options_default(&c->taper);
        hc_irregular_wave_params_default(&c->irr);
        *out = c.release();
    } catch (const Error& e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

int hc_create(int num_bodies, int device_id, hc_ctx** out) { return hc_create_sharded(num_bodies, 0, num_bodies, device_id, out); }

void hc_destroy(hc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    delete ctx->dq;  // drains its queue
    ctx->dq = nullptr;
    (void)hipDeviceSynchronize();  // steps may still be running on a caller's stream; the buffers go away below
    if (ctx->ext_tag_host) (void)hipHostUnregister(ctx->ext_tag_host);
    for (auto& ev : ctx->events) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream_am) (void)hipStreamDestroy(ctx->stream_am);
    if (ctx->ev_fin) (void)hipEventDestroy(ctx->ev_fin);
    if (ctx->ev_bg) (void)hipEventDestroy(ctx->ev_bg);
    delete ctx;
}

// ---- ingest -----------------------------------------------------------------------------------
int hc_set_simulation_parameters(hc_ctx* c, double rho, double g, double water_depth) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    c->rho = rho;
    c->g = g;
    c->depth = water_depth;
    c->have_sim = true;
    HC_API_END(c)
}

int hc_set_body_properties(hc_ctx* c, int body, double disp_vol, const double cg[3], const double cb[3]) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(cg && cb, HC_ERR_INVALID, "null pointer");
    auto& b = c->bodies[body];
    b.disp_vol = disp_vol;
    std::copy(cg, cg + 3, b.cg);
    std::copy(cb, cb + 3, b.cb);
    b.have_props = true;
    HC_API_END(c)
}

int hc_set_hydrostatic_stiffness(hc_ctx* c, int body, const double lin[36]) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(lin, HC_ERR_INVALID, "null pointer");
    std::copy(lin, lin + 36, c->bodies[body].lin);
    c->bodies[body].have_lin = true;
    HC_API_END(c)
}

int hc_set_added_mass_inf(hc_ctx* c, int body, const double* A) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(A, HC_ERR_INVALID, "null pointer");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho scales the added mass)");
    auto& b = c->bodies[body];
    b.have_ainf = true;
    if (is_local(c, body)) {
        b.ainf.assign(A, A + static_cast<size_t>(6) * c->D);
        for (auto& x : b.ainf) x *= c->rho;  // src/h5fileinfo.cpp:61
    }
    HC_API_END(c)
}

int hc_set_rirf(hc_ctx* c, int body, const double* t, int S, const double* K) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(t && K && S > 0, HC_ERR_INVALID, "null pointer or empty IRF");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho scales the radiation IRF)");
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    if (c->S == 0) {
        c->S = S;
        c->tau.assign(t, t + S);
        setup_panel_geometry(c);
        c->dK.alloc(hc::panel_doubles(c->ntiles, c->ngp));
        HC_HIP(hipMemsetAsync(c->dK.p, 0, c->dK.n * sizeof(double), c->stream));  // padding rows / columns stay zero
        c->d_stage.alloc(static_cast<size_t>(6) * c->D * S);
    } else {
        require(S == c->S, HC_ERR_RUNTIME, "RIRF time vectors have to be exactly the same for all bodies (length differs)");
        for (int j = 0; j < S; ++j)
            if (std::fabs(t[j] - c->tau[j]) > 1e-10)
                throw Error(HC_ERR_RUNTIME, "RIRF time vectors have to be exactly the same for all bodies.");
    }
    c->bodies[body].have_rirf = true;
    for (int j = 0; j < S; ++j)
        if (t[j] < 0.0) c->device_errors_possible = true;  // a query t - tau could then exceed t (reference: throws at :370)
    if (is_local(c, body)) {
        HC_HIP(hipMemcpyAsync(c->d_stage.p, K, static_cast<size_t>(6) * c->D * S * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hc::launch_relayout_rirf(c->d_stage.p, c->dK.p, c->ngp, c->D, S, 6 * (body - c->b0), c->rho, c->stream);
        HC_HIP(hipGetLastError());
        HC_HIP(hipStreamSynchronize(c->stream));
        c->proc_ready = false;
    }
    HC_API_END(c)
}

int hc_set_excitation_rao(hc_ctx* c, int body, const double* w, int nw, const double* mag, const double* phase) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(w && mag && phase && nw > 0, HC_ERR_INVALID, "null pointer or empty RAO");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho*g scales the excitation magnitude)");
    auto& b = c->bodies[body];
    b.rao_w.assign(w, w + nw);
    b.rao_mag.assign(mag, mag + static_cast<size_t>(6) * nw);
    const double rg = c->rho * c->g;
    for (auto& x : b.rao_mag) x = x * rg;  // src/h5fileinfo.cpp:73-75
    b.rao_phase.assign(phase, phase + static_cast<size_t>(6) * nw);
    b.have_rao = true;
    HC_API_END(c)
}

int hc_set_excitation_irf(hc_ctx* c, int body, const double* t, int n, const double* f) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(t && f && n > 0, HC_ERR_INVALID, "null pointer or empty excitation IRF");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho*g scales the excitation IRF)");
    auto& b = c->bodies[body];
    b.exirf_t.assign(t, t + n);
    b.have_exirf = true;
    if (is_local(c, body)) {
        b.exirf_f.assign(f, f + static_cast<size_t>(6) * n);
        const double rg = c->rho * c->g;
        for (auto& x : b.exirf_f) x *= rg;  // src/h5fileinfo.cpp:90
    }
    HC_API_END(c)
}

namespace {
// The HDF5 code lives in libhc_bemio.so (built only where libhdf5 exists) next to this library.
using bemio_fn_t = int (*)(hc_ctx*, const char*, char*, size_t);
bemio_fn_t bemio_symbol(const char* name) {
    Dl_info info;
    std::string dir = ".";
    if (dladdr(reinterpret_cast<void*>(&hc_version), &info) && info.dli_fname) {
        std::string full(info.dli_fname);
        const size_t slash = full.find_last_of('/');
        if (slash != std::string::npos) dir = full.substr(0, slash);
    }
    const std::string lib = dir + "/libhc_bemio.so";
    void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) throw Error(HC_ERR_UNSUPPORTED, std::string("HDF5 support not available: ") + dlerror());
    bemio_fn_t fn = reinterpret_cast<bemio_fn_t>(dlsym(h, name));
    if (!fn) throw Error(HC_ERR_UNSUPPORTED, std::string("libhc_bemio.so lacks ") + name);
    return fn;
}
}  // namespace

int hc_load_bemio_h5(hc_ctx* c, const char* path) {
    HC_API_BEGIN(c)
    require(path, HC_ERR_INVALID, "null path");
    char msg[1024] = {0};
    const int rc = bemio_symbol("hc_bemio_load")(c, path, msg, sizeof msg);
    if (rc != HC_OK) throw Error(rc, msg[0] ? std::string(msg) : c->err);
    HC_API_END(c)
}

int hc_export_irregular_inputs_h5(hc_ctx* c, const char* path) {
    HC_API_BEGIN(c)
    require(path, HC_ERR_INVALID, "null path");
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    char msg[1024] = {0};
    const int rc = bemio_symbol("hc_bemio_export_irregular")(c, path, msg, sizeof msg);
    if (rc != HC_OK) throw Error(rc, msg[0] ? std::string(msg) : c->err);
    HC_API_END(c)
}

int hc_finalize(hc_ctx* c) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    require(c->have_sim, HC_ERR_INVALID, "simulation parameters missing");
    require(c->S > 0, HC_ERR_INVALID, "radiation IRF missing");
    for (int b = c->b0; b < c->b1; ++b) {
        const auto& bd = c->bodies[b];
        require(bd.have_props && bd.have_lin && bd.have_ainf && bd.have_rirf, HC_ERR_INVALID,
                "a local body lacks properties, hydrostatic stiffness, added mass or radiation IRF");
    }
    // trapezoid widths (src/hydro_forces.cpp:181-190)
    c->width = hc::trapezoid_widths(c->tau);
    c->d_tau.upload(c->tau, c->stream);
    c->d_width.upload(c->width, c->stream);
    // hydrostatics tables; equilibrium = cg, cb - cg (:208-216)
    std::vector<double> lin(static_cast<size_t>(c->nloc) * 36), cg(static_cast<size_t>(c->nloc) * 3), cbm(static_cast<size_t>(c->nloc) * 3),
        vol(c->nloc);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        std::copy(bd.lin, bd.lin + 36, lin.begin() + static_cast<size_t>(bl) * 36);
        for (int k = 0; k < 3; ++k) {
            cg[bl * 3 + k]  = bd.cg[k];
            cbm[bl * 3 + k] = bd.cb[k] - bd.cg[k];
        }
        vol[bl] = bd.disp_vol;
    }
    c->d_lin.upload(lin, c->stream);
    c->d_cg.upload(cg, c->stream);
    c->d_cbmcg.upload(cbm, c->stream);
    c->d_vol.upload(vol, c->stream);
    // added mass rows (src/chloadaddedmass.cpp:18-21)
    c->ainf_host.assign(static_cast<size_t>(c->Dloc) * c->D, 0.0);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        std::copy(bd.ainf.begin(), bd.ainf.end(), c->ainf_host.begin() + static_cast<size_t>(bl) * 6 * c->D);
    }
    c->d_ainf.upload(c->ainf_host, c->stream);
    c->d_vec_w.alloc(c->D);
    c->d_vec_R.alloc(c->Dloc);
    c->d_stage.release();
    // history ring
    ring_alloc(c, std::max(64, c->S + 2) + hc::kRewindSlack);
    c->times.clear();
    c->retired.clear();
    c->have_prev = c->have_prev_device = false;
    c->prev_time = c->prev_time_device = -1.0;
    {
        const int want = env_int("HC_LOOKAHEAD", 32);
        c->lookahead   = want <= 0 ? 0 : (want <= 16 ? 16 : hc::kLookahead);
    }
    // GEMV scratch
    choose_conv_config(c);
    // step I/O
    c->d_state.alloc(static_cast<size_t>(12) * c->N);
    c->d_hs.alloc(c->Dloc);
    c->d_rad.alloc(c->Dloc);
    c->d_waves.alloc(c->Dloc);
    c->d_total.alloc(c->Dloc);
    for (auto* b : {&c->d_hs, &c->d_rad, &c->d_waves, &c->d_total}) HC_HIP(hipMemsetAsync(b->p, 0, b->n * sizeof(double), c->stream));
    c->d_err.alloc(2);  // [0] error flag of the convolution kernels, [1] work-item counter of the look-ahead pass
    HC_HIP(hipMemsetAsync(c->d_err.p, 0, 2 * sizeof(int), c->stream));
    c->h_state.alloc(static_cast<size_t>(2) * 12 * c->N);  // two halves used alternately by hc_step, see there
    c->bar_state.alloc(static_cast<size_t>(2) * 12 * c->N);
    if (c->bar_state.host_ok) {
        // trust, but verify: what the host stores through the BAR must be what a device-side copy sees
        const size_t nb = c->bar_state.n;
        for (size_t k = 0; k < nb; ++k) c->bar_state.p[k] = 0.5 + static_cast<double>(k);
        _mm_sfence();
        hc::DeviceBuffer<double> tmp;
        tmp.alloc(nb);
        std::vector<double> back(nb);
        HC_HIP(hipMemcpyAsync(tmp.p, c->bar_state.p, nb * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(back.data(), tmp.p, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HC_HIP(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < nb && c->bar_state.host_ok; ++k) c->bar_state.host_ok = back[k] == 0.5 + static_cast<double>(k);
    }
    c->h_out.alloc(static_cast<size_t>(4) * c->Dloc);
    c->h_err.alloc(1);
    c->h_am.alloc(static_cast<size_t>(c->D) + c->Dloc);
    c->bar_am.alloc(static_cast<size_t>(c->D) + c->Dloc);
    c->h_tag_am.alloc(static_cast<size_t>(2) * c->Dloc);
    c->bar_selftest.alloc(2);
    c->h_tag_selftest.alloc(2);
    c->d_selftest.alloc(1);
    std::memset(c->h_tag_am.p, 0, c->h_tag_am.n * sizeof(unsigned long long));
    c->seq_am = 0;
    c->h_tag.alloc(static_cast<size_t>(4) * c->Dloc);  // two halves of [Dloc][2], used alternately (step sequence parity)
    std::memset(c->h_tag.p, 0, c->h_tag.n * sizeof(unsigned long long));
    c->seq = 0;
    c->last_total.assign(c->Dloc, 0.0);
    c->d_scratch.alloc(static_cast<size_t>(4) * c->Dloc);
    c->d_zero_state.alloc(static_cast<size_t>(12) * c->N);
    HC_HIP(hipMemsetAsync(c->d_zero_state.p, 0, c->d_zero_state.n * sizeof(double), c->stream));
    c->zero_copy_max_bodies = env_int("HC_ZERO_COPY_BODIES", 64);
    // default wave model: NoWave for all bodies (the reference's default NoWave() covers one body only and is
    // read out of bounds for N > 1, src/hydro_forces.cpp:758-760; that overread is deliberately not reproduced)
    c->wave_kind   = hc::kWaveNone;
    c->wave_nb_arg = c->N;
    choose_exc_config(c);
    alloc_partials(c);
    c->prof.conv_kernel_bytes  = 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D);
    c->prof.block_kernel_bytes = hc::kLookahead * c->prof.conv_kernel_bytes;
    c->plan                    = hc::Plan{};
    HC_HIP(hipStreamSynchronize(c->stream));
    setup_direct(c);
    c->finalized = true;
    HC_API_END(c)
}

// ---- configuration ----------------------------------------------------------------------------
int hc_set_gravity(hc_ctx* c, const double g[3]) {
    HC_API_BEGIN(c)
    require(g, HC_ERR_INVALID, "null pointer");
    std::copy(g, g + 3, c->gsys);
    HC_API_END(c)
}

int hc_set_wave_none(hc_ctx* c, int num_bodies_arg) {
    HC_API_BEGIN(c)
    c->plan.has_exc = false;  // excitation rows precomputed by a look-ahead pass belong to the previous wave model
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(num_bodies_arg >= 0, HC_ERR_INVALID, "negative body count");
    c->wave_kind   = hc::kWaveNone;
    c->wave_nb_arg = num_bodies_arg;
    choose_exc_config(c);
    HC_API_END(c)
}

int hc_set_wave_regular(hc_ctx* c, int num_bodies_arg, double amplitude, double omega) {
    HC_API_BEGIN(c)
    c->plan.has_exc = false;  // excitation rows precomputed by a look-ahead pass belong to the previous wave model
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(num_bodies_arg >= 1 && num_bodies_arg <= c->N, HC_ERR_OUT_OF_RANGE, "regular wave created for more bodies than the hydro data holds");
    for (int b = 0; b < num_bodies_arg; ++b) require(c->bodies[b].have_rao, HC_ERR_INVALID, "excitation RAO missing for a body");
    // RegularWave::AddH5Data (src/wave_types.cpp:278-299), GetOmegaDelta (:329-333), Get*Interp (:335-352)
    const auto& w0        = c->bodies[0].rao_w;
    const double nfreq    = static_cast<double>(w0.size());
    const double dw       = w0.back() / nfreq;
    const double idx_des  = (omega / dw) - 1;
    const double frac     = idx_des - std::floor(idx_des);
    const int k0          = static_cast<int>(std::floor(idx_des));
    std::vector<double> mag(c->D, 0.0), ph(c->D, 0.0);
    for (int b = 0; b < num_bodies_arg; ++b) {
        const auto& bd = c->bodies[b];
        const int nw   = static_cast<int>(bd.rao_w.size());
        if (k0 < 0 || k0 + 1 >= nw) throw Error(HC_ERR_OUT_OF_RANGE, "regular wave frequency outside the BEM frequency list");
        for (int r = 0; r < 6; ++r) {
            const double m0 = bd.rao_mag[static_cast<size_t>(r) * nw + k0], m1 = bd.rao_mag[static_cast<size_t>(r) * nw + k0 + 1];
            const double p0 = bd.rao_phase[static_cast<size_t>(r) * nw + k0], p1 = bd.rao_phase[static_cast<size_t>(r) * nw + k0 + 1];
            mag[6 * b + r] = (frac * (m1 - m0)) + m0;
            ph[6 * b + r]  = (frac * (p1 - p0)) + p0;
        }
    }
    const double k = hc::wave_number(omega, c->depth, c->g);  // RegularWave::Initialize (:274-276)
    c->reg_mag = mag;
    c->reg_phase = ph;
    c->reg_amp = amplitude;
    c->reg_omega = omega;
    c->reg_wavenumber = k;
    std::vector<double> local(mag.begin() + 6 * c->b0, mag.begin() + 6 * c->b1);
    c->d_reg_mag.upload(local, c->stream);
    c->wave_kind   = hc::kWaveRegular;
    c->wave_nb_arg = num_bodies_arg;
    choose_exc_config(c);
    HC_API_END(c)
}

void hc_irregular_wave_params_default(hc_irregular_wave_params* p) {
    if (!p) return;
    p->num_bodies = 1;
    p->simulation_dt = 0.0;
    p->simulation_duration = 0.0;
    p->ramp_duration = 0.0;
    p->wave_height = 0.0;
    p->wave_period = 0.0;
    p->frequency_min = 0.001;
    p->frequency_max = 1.0;
    p->nfrequencies = 0;
    p->peak_enhancement_factor = 1.0;
    p->is_normalized = 0;
    p->seed = 1;
}

int hc_set_wave_irregular(hc_ctx* c, const hc_irregular_wave_params* pp) {
    HC_API_BEGIN(c)
    c->plan.has_exc = false;  // excitation rows precomputed by a look-ahead pass belong to the previous wave model
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pp, HC_ERR_INVALID, "null parameters");
    const hc_irregular_wave_params p = *pp;
    require(p.num_bodies == c->N, HC_ERR_INVALID, "IrregularWaveParams.num_bodies_ must equal the number of hydro bodies");
    require(p.simulation_dt > 0.0, HC_ERR_INVALID, "simulation_dt must be positive");
    require(p.wave_height != 0.0 && p.wave_period != 0.0, HC_ERR_INVALID,
            "wave_height and wave_period must be non-zero (the reference leaves the free-surface table empty otherwise)");
    // Excitation-IRF time grids.  The reference keeps one grid per body (ex_irf_time_sampled_[b], src/wave_types.cpp:432-459) and
    // resamples each on its own (:572-606).  BEMIO writes one grid per file, so bodies normally share it; bodies with the same
    // grid (within 1e-10, the tolerance the reference applies to the radiation grids) form a group, and the columns of Kex are
    // the groups' resampled grids one after the other -- a body's rows are non-zero in the columns of its own group only, so
    // one launch still serves all bodies.  Groups are formed over ALL bodies (not only the local ones): the column layout, and
    // with it the summation order, is the same in every row shard of the system.
    std::vector<hc::ExGroup> groups;
    std::vector<int> group_of(c->N, -1);
    for (int b = 0; b < c->N; ++b) {
        if (!c->bodies[b].have_exirf) {
            // a row-sharded context whose caller ingested its own bodies only: the other bodies' grids are unknown here
            require(!is_local(c, b), HC_ERR_INVALID, "excitation IRF missing for a local body");
            continue;
        }
        const auto& tb = c->bodies[b].exirf_t;
        require(tb.size() >= 2, HC_ERR_INVALID, "excitation IRF with fewer than two samples");
        for (size_t g = 0; g < groups.size() && group_of[b] < 0; ++g) {
            const auto& tg = c->bodies[groups[g].first_body].exirf_t;
            bool same = tg.size() == tb.size();
            for (size_t j = 0; j < tb.size() && same; ++j) same = std::fabs(tb[j] - tg[j]) <= 1e-10;
            if (same) group_of[b] = static_cast<int>(g);
        }
        if (group_of[b] < 0) {
            hc::ExGroup g;
            g.first_body = b;
            group_of[b]  = static_cast<int>(groups.size());
            groups.push_back(g);
        }
    }
    // ResampleIRF (src/wave_types.cpp:572-606), per group
    std::vector<double> ex_tau, ex_width;
    int L = 0;
    for (auto& g : groups) {
        const auto& t_old = c->bodies[g.first_body].exirf_t;
        const double t0 = t_old.front(), t1 = t_old.back();
        g.L   = static_cast<int>(std::ceil((t1 - t0) / p.simulation_dt));
        require(g.L >= 2, HC_ERR_INVALID, "excitation IRF resamples to fewer than two points");
        g.off = L;
        const std::vector<double> tg = hc::linspaced(g.L, t0, t1), wg = hc::trapezoid_widths(tg);
        g.tau_front = tg.front();
        g.tau_back  = tg.back();
        ex_tau.insert(ex_tau.end(), tg.begin(), tg.end());
        ex_width.insert(ex_width.end(), wg.begin(), wg.end());
        L += g.L;
    }
    const int Lpad = (L + 7) & ~7;
    std::vector<double> vals(static_cast<size_t>(c->Dloc) * L, 0.0);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        const hc::ExGroup& g = groups[group_of[c->b0 + bl]];
        const auto r = hc::resample_cubic_bspline6(bd.exirf_f, static_cast<int>(bd.exirf_t.size()), g.L);
        for (int d = 0; d < 6; ++d)
            std::copy(r.begin() + static_cast<size_t>(d) * g.L, r.begin() + static_cast<size_t>(d + 1) * g.L,
                      vals.begin() + static_cast<size_t>(6 * bl + d) * L + g.off);
    }
    // CreateSpectrum (:643-676)
    Spectrum sp = build_spectrum(c, p);
    const int nf = sp.nf;
    std::vector<double>&f = sp.f, &Sd = sp.S, &dfv = sp.df, &phase = sp.phase, &kk = sp.k, &amp = sp.amp, &omg = sp.omega;
    // CreateFreeSurfaceElevation (:717-774): the min/max scan over every body's resampled grid = over the groups' ends
    double t_irf_min = 0.0, t_irf_max = 0.0;
    for (const auto& g : groups) {
        if (g.tau_front < t_irf_min) t_irf_min = g.tau_front;
        if (g.tau_front > t_irf_max) t_irf_max = g.tau_front;
        if (g.tau_back > t_irf_max) t_irf_max = g.tau_back;
        if (g.tau_back < t_irf_min) t_irf_min = g.tau_back;
    }
    const double duration = p.simulation_duration + 2 * (t_irf_max - t_irf_min);
    const int nts         = static_cast<int>(std::ceil(duration / p.simulation_dt));
    std::vector<double> eta_t = hc::linspaced(nts + 1, 0, nts * p.simulation_dt);
    for (auto& x : eta_t) x += -t_irf_max;
    const int nt = nts + 1;
    require(nt >= 2, HC_ERR_INVALID, "free-surface table too short");

    hc::DeviceBuffer<double> d_amp, d_omg, d_ph;
    d_amp.upload(amp, c->stream);
    d_omg.upload(omg, c->stream);
    d_ph.upload(phase, c->stream);
    c->d_eta_t.upload(eta_t, c->stream);
    c->d_eta.alloc(nt);
    if (c->eta_mode == 1 && nf >= 2) {
        hc::eta_synthesis_fft(eta_t, amp, omg, phase, p.ramp_duration, c->d_eta_t.p, c->d_eta.p, c->stream);  // rocFFT chirp-z
    } else {
        hc::launch_eta_synthesis(c->d_eta_t.p, nt, d_amp.p, d_omg.p, d_ph.p, nf, p.ramp_duration, c->d_eta.p, c->stream);
    }
    HC_HIP(hipGetLastError());
    std::vector<double> eta(nt);
    HC_HIP(hipMemcpyAsync(eta.data(), c->d_eta.p, nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));

    {   // excitation IRF into the same panel layout as K (row tiles of 16, column groups of 8)
        hc::DeviceBuffer<double> d_rowmajor;
        d_rowmajor.upload(vals, c->stream);
        c->ngp_ex = Lpad / 8;
        c->d_kex.alloc(hc::panel_doubles(c->ntiles, c->ngp_ex));
        HC_HIP(hipMemsetAsync(c->d_kex.p, 0, c->d_kex.n * sizeof(double), c->stream));
        hc::launch_relayout_rowmajor(d_rowmajor.p, c->Dloc, L, c->d_kex.p, c->ngp_ex, 0, c->stream);
        HC_HIP(hipGetLastError());
        HC_HIP(hipStreamSynchronize(c->stream));
    }
    c->d_ex_tau.upload(ex_tau, c->stream);
    c->d_ex_width.upload(ex_width, c->stream);
    c->irr = p;
    c->L = L;
    c->Lpad = Lpad;
    c->nf = nf;
    c->nt = nt;
    c->ex_tau.swap(ex_tau);
    c->ex_width.swap(ex_width);
    c->ex_vals.swap(vals);
    c->ex_groups.swap(groups);
    c->ex_group_of.swap(group_of);
    c->spec_f.swap(f);
    c->spec_S.swap(Sd);
    c->spec_df.swap(dfv);
    c->spec_phase.swap(phase);
    c->spec_k.swap(kk);
    c->eta_t.swap(eta_t);
    c->eta.swap(eta);
    c->wave_kind   = hc::kWaveIrregular;
    c->wave_nb_arg = p.num_bodies;
    choose_exc_config(c);
    alloc_partials(c);
    c->prof.conv_kernel_bytes = 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D +
                                       static_cast<double>(c->Dloc) * L + L);
    c->prof.block_kernel_bytes = hc::kLookahead * 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D);
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

int hc_set_wave_irregular_spectral(hc_ctx* c, const hc_irregular_wave_params* pp) {
    HC_API_BEGIN(c)
    c->plan.has_exc = false;  // excitation rows precomputed by a look-ahead pass belong to the previous wave model
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pp, HC_ERR_INVALID, "null parameters");
    const hc_irregular_wave_params p = *pp;
    require(p.num_bodies == c->N, HC_ERR_INVALID, "IrregularWaveParams.num_bodies_ must equal the number of hydro bodies");
    require(p.wave_height != 0.0 && p.wave_period != 0.0, HC_ERR_INVALID, "wave_height and wave_period must be non-zero");
    for (int b = c->b0; b < c->b1; ++b) require(c->bodies[b].have_rao, HC_ERR_INVALID, "excitation RAO missing for a local body");
    Spectrum sp = build_spectrum(c, p);
    // per-component excitation RAO: RegularWave's interpolator (src/wave_types.cpp:329-352: uniform list starting at
    // d_omega, index = omega/d_omega - 1, linear between neighbours), held constant outside the BEM frequency range
    std::vector<double> mag(static_cast<size_t>(c->Dloc) * sp.nf), ph(static_cast<size_t>(c->Dloc) * sp.nf);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd  = c->bodies[c->b0 + bl];
        const int nw    = static_cast<int>(bd.rao_w.size());
        const double dw = bd.rao_w.back() / static_cast<double>(nw);
        for (int i = 0; i < sp.nf; ++i) {
            double idx = sp.omega[i] / dw - 1;
            idx        = std::min(std::max(idx, 0.0), static_cast<double>(nw - 1));
            const int k0 = std::min(static_cast<int>(std::floor(idx)), nw - 2 >= 0 ? nw - 2 : 0);
            const int k1 = std::min(k0 + 1, nw - 1);
            const double fr = idx - k0;
            for (int r = 0; r < 6; ++r) {
                const double m0 = bd.rao_mag[static_cast<size_t>(r) * nw + k0], m1 = bd.rao_mag[static_cast<size_t>(r) * nw + k1];
                const double p0 = bd.rao_phase[static_cast<size_t>(r) * nw + k0], p1 = bd.rao_phase[static_cast<size_t>(r) * nw + k1];
                mag[static_cast<size_t>(6 * bl + r) * sp.nf + i] = (fr * (m1 - m0)) + m0;
                ph[static_cast<size_t>(6 * bl + r) * sp.nf + i]  = (fr * (p1 - p0)) + p0;
            }
        }
    }
    c->d_spec_mag.upload(mag, c->stream);
    c->d_spec_phase.upload(ph, c->stream);
    c->d_spec_amp.upload(sp.amp, c->stream);
    c->d_spec_omega.upload(sp.omega, c->stream);
    c->d_spec_phi.upload(sp.phase, c->stream);
    c->irr = p;
    c->nf  = sp.nf;
    c->L = c->Lpad = c->nt = 0;
    c->spec_f.swap(sp.f);
    c->spec_S.swap(sp.S);
    c->spec_df.swap(sp.df);
    c->spec_phase.swap(sp.phase);
    c->spec_k.swap(sp.k);
    c->wave_kind   = hc::kWaveSpectral;
    c->wave_nb_arg = p.num_bodies;
    choose_exc_config(c);
    HC_API_END(c)
}

int hc_set_eta_synthesis(hc_ctx* c, int mode) {
    HC_API_BEGIN(c)
    require(mode == 0 || mode == 1, HC_ERR_INVALID, "mode must be 0 (direct FP64 sum) or 1 (rocFFT chirp-z)");
    c->eta_mode = mode;
    HC_API_END(c)
}

int hc_set_convolution_mode(hc_ctx* c, int mode) {
    HC_API_BEGIN(c)
    require(mode == 0 || mode == 1, HC_ERR_INVALID, "mode must be 0 (Baseline) or 1 (TaperedDirect)");
    c->conv_mode  = mode;
    c->plan.valid = false;
    HC_API_END(c)
}

void hc_tapered_direct_options_default(hc_tapered_direct_options* o) {
    if (!o) return;
    o->smoothing = 0;
    o->window_length = 5;
    o->rirf_end_time = -1.0;
    o->taper_start_percent = 0.8;
    o->taper_end_percent = 1.0;
    o->taper_final_amplitude = 0.0;
    o->export_plot_csv = 0;
}

int hc_set_diagnostics_output_directory(hc_ctx* c, const char* dir) {
    HC_API_BEGIN(c)
    c->diagnostics_dir = dir ? dir : "";
    HC_API_END(c)
}

int hc_set_tapered_direct_options(hc_ctx* c, const hc_tapered_direct_options* o) {
    HC_API_BEGIN(c)
    require(o, HC_ERR_INVALID, "null options");
    require(o->smoothing == 0 || o->smoothing == 1, HC_ERR_INVALID, "smoothing must be 0 (sg) or 1 (moving_average)");
    c->taper      = *o;
    c->proc_ready = false;
    c->plan.valid = false;
    HC_API_END(c)
}

// ---- per-step ---------------------------------------------------------------------------------
namespace {
double step_timeout_seconds() {
    static const double s = [] {
        const char* e = std::getenv("HC_STEP_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        return v > 0.0 ? v : 20.0;
    }();
    return s;
}

[[noreturn]] void device_lost(hc_ctx* c, const std::string& what) {
    c->lost         = true;
    c->direct_ready = false;  // nothing more goes to a queue that has stopped answering
    c->direct_why   = what;
    throw Error(HC_ERR_DEVICE, what);
}

// Wait until finalize_kernel's {total, sequence} granules of step `seq` have all arrived in mapped pinned memory, then
// copy the totals out.  Each granule is one 16-byte store, so its value is valid as soon as its sequence number is.  This
// replaces hipStreamSynchronize on the per-step path (14 -> 9 us for an empty launch, profiles/r02/latency_probe_v1.txt).
// The wait is bounded: every 2^16 spins the slow path looks at the clock, at the queue's error flag (direct path) or at the
// stream (HIP path), so a failed launch or a lost device ends the wait with HC_ERR_DEVICE after HC_STEP_TIMEOUT_S (20 s)
// instead of hanging the host.
void wait_tagged(hc_ctx* c, const unsigned long long* granules, unsigned long long seq, hipStream_t stream, double* out, int lane = 0) {
    const volatile unsigned long long* g = granules;
    unsigned long long spins = 0;
    std::chrono::steady_clock::time_point t_begin{};
    for (int r = c->Dloc - 1; r >= 0; --r) {
        while (g[2 * r + 1] != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
                const auto now = std::chrono::steady_clock::now();
                if (spins == 0x10000) t_begin = now;
                const bool timed_out = std::chrono::duration<double>(now - t_begin).count() > step_timeout_seconds();
                if (stream == nullptr || (c->path == 2 && stream == c->stream)) {
                    // the step went to the direct queue: there is no stream to ask
                    if (c->dq && c->dq->failed(lane)) device_lost(c, "hc_step: the HSA queue of the direct dispatch reported an error: " + c->dq->failure_text());
                    if (timed_out) device_lost(c, "hc_step: the step's results did not arrive (direct queue, timeout)");
                    continue;
                }
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) {
                    if (g[2 * r + 1] != seq) device_lost(c, "hc_step: the stream drained but the step's results did not arrive");
                } else if (q != hipErrorNotReady) {
                    device_lost(c, std::string("hc_step: ") + hipGetErrorString(q));
                } else if (timed_out) {
                    device_lost(c, "hc_step: the step's results did not arrive (timeout)");
                }
            }
        }
    }
    for (int r = 0; r < c->Dloc; ++r) {
        const unsigned long long bits = g[2 * r];
        std::memcpy(out + r, &bits, sizeof(double));
    }
}

unsigned long long* result_tags_dev(hc_ctx* c, unsigned long long seq) {
    return (c->ext_tag_dev ? c->ext_tag_dev : c->h_tag.dp) + (seq & 1) * static_cast<size_t>(2) * c->Dloc;
}
const unsigned long long* result_tags_host(hc_ctx* c, unsigned long long seq) {
    return (c->ext_tag_host ? c->ext_tag_host : c->h_tag.p) + (seq & 1) * static_cast<size_t>(2) * c->Dloc;
}

// First half of a synchronous step: cache rules of CoordinateFuncForBody (src/hydro_forces.cpp:742-751), the state stored where
// the kernels read it, the step kernel handed to the GPU.  Leaves c->pending_step = 1 (cache hit, totals in last_total) or 2
// (results arrive as tagged granules of sequence number c->seq).  defer_tail: see enqueue_step.
void step_begin(hc_ctx* c, double t, const double* pos, const double* rpy, const double* linvel, const double* angvel, bool defer_tail) {
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pos && rpy && linvel && angvel, HC_ERR_INVALID, "null pointer");
    require(c->pending_step == 0, HC_ERR_INVALID, "the step begun before has not been completed (hc_step_end)");
    if (c->lost) throw Error(HC_ERR_DEVICE, "the device stopped answering in an earlier step: " + c->direct_why);
    if (c->have_prev && t == c->prev_time) {  // src/hydro_forces.cpp:742-744
        c->pending_step = 1;
        return;
    }
    if (c->have_prev_device && t == c->prev_time_device) {
        // this time was evaluated through hc_step_device (possibly on a caller's stream): fetch its totals, do not re-evaluate
        quiesce_direct(c);
        HC_HIP(hipDeviceSynchronize());
        HC_HIP(hipMemcpy(c->last_total.data(), c->d_total.p, c->Dloc * sizeof(double), hipMemcpyDeviceToHost));
        c->prev_time    = t;
        c->have_prev    = true;
        c->pending_step = 1;
        return;
    }
    c->prev_time = t;  // :747 (set before the terms are computed, so a throwing step is not retried)
    c->have_prev = true;
    c->have_prev_device = false;  // d_total is about to be replaced (or left stale by a step that throws)
    std::fill(c->last_total.begin(), c->last_total.end(), 0.0);  // the reference zero-fills total_force_ before the terms (:749-751)
    // Boundary without copy launches or stream synchronisation: the host stores the state doubles straight into device
    // memory through the PCIe BAR (fallback: mapped pinned memory the kernels read over PCIe), finalize_kernel stores the
    // totals straight into mapped pinned memory, tagged with this step's sequence number; one kernel launch for a step
    // inside a block.
    // The state buffer has two halves used alternately: this call returns as soon as the totals have arrived, while the
    // workgroup that stores the step's sample into the ring may still be reading the state -- the next call must not
    // overwrite it.  (The kernels of step n+1 run after those of step n, and step n+2 starts only after the totals of
    // step n+1 have arrived, so two halves are enough.)
    // A row-sharded context reads positions and angles of its OWN bodies only (hydrostatics), velocities of all: the other
    // bodies' pos / rpy entries are not stored (half the bytes through the BAR for each shard of a wide array).
    const int n3   = 3 * c->N;
    const size_t o = (c->seq & 1) ? static_cast<size_t>(12) * c->N : 0;
    const size_t l0 = static_cast<size_t>(3) * c->b0, ln = static_cast<size_t>(3) * c->nloc;
    auto put = [&](double* h) {
        std::memcpy(h + l0, pos + l0, ln * sizeof(double));
        std::memcpy(h + n3 + l0, rpy + l0, ln * sizeof(double));
        std::memcpy(h + 2 * n3, linvel, n3 * sizeof(double));
        std::memcpy(h + 3 * n3, angvel, n3 * sizeof(double));
    };
    const double* d_state;
    if (c->bar_state.host_ok) {
        // device memory written through the PCIe BAR: the kernels read the state locally (no PCIe read on the critical path)
        double* h = c->bar_state.p + o;
        put(h);
        _mm_sfence();  // write-combined stores are globally visible before the doorbell of the launch
        d_state = h;
    } else {
        double* h = c->h_state.p + o;
        put(h);
        d_state = c->h_state.dp + o;
        if (c->N > c->zero_copy_max_bodies) {  // many workgroups re-read the state: one small H2D copy beats their PCIe reads
            HC_HIP(hipMemcpyAsync(c->d_state.p, h, 4 * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            d_state = c->d_state.p;
        }
    }
    const unsigned long long seq = ++c->seq;
    // The tagged results of consecutive steps go to alternate halves of the result buffer: a reader in ANOTHER process (a caller's
    // buffer in shared memory, hc_set_result_buffer) may still be collecting step n while this process has moved on to step n + 1;
    // it cannot reach step n + 2 before every process has the rows of step n + 1, i.e. has finished with step n.
    enqueue_step(c, t, d_state, nullptr, c->stream, StepFlags{}, result_tags_dev(c, seq), seq, defer_tail);
    c->pending_step = 2;
    c->pending_t    = t;
}

// Second half: wait for the tagged totals of the step begun last and hand them out.
void step_end(hc_ctx* c, double* force_out) {
    require(c->pending_step != 0, HC_ERR_INVALID, "hc_step_end without hc_step_begin");
    const int how   = c->pending_step;
    c->pending_step = 0;
    if (how == 2) {
        if (c->tail.pending) enqueue_tail(c);  // (a caller that deferred the tail and never enqueued it)
        wait_tagged(c, result_tags_host(c, c->seq), c->seq, c->stream, c->last_total.data());
        if (c->device_errors_possible) {
            quiesce_direct(c);
            check_device_flag(c);
        }
        c->prev_time_device = c->pending_t;  // finalize_kernel has left the same totals in d_total: hc_step_device at this time copies them
        c->have_prev_device = true;
    }
    if (force_out) std::memcpy(force_out, c->last_total.data(), c->Dloc * sizeof(double));
}

// A failed begin leaves nothing pending; whatever the step had enqueued before it threw is harmless (results nobody waits for).
void step_abort(hc_ctx* c) {
    c->pending_step = 0;
    c->tail.pending = false;
}
}  // namespace

int hc_step(hc_ctx* c, double t, const double* pos, const double* rpy, const double* linvel, const double* angvel, double* force_out) {
    HC_API_BEGIN_HOT(c)
    require(force_out, HC_ERR_INVALID, "null pointer");
    try {
        step_begin(c, t, pos, rpy, linvel, angvel, false);
        step_end(c, force_out);
    } catch (...) {
        step_abort(c);
        throw;
    }
    HC_API_END(c)
}

int hc_step_begin(hc_ctx* c, double t, const double* pos, const double* rpy, const double* linvel, const double* angvel) {
    HC_API_BEGIN_HOT(c)
    try {
        step_begin(c, t, pos, rpy, linvel, angvel, false);
    } catch (...) {
        step_abort(c);
        throw;
    }
    HC_API_END(c)
}

int hc_step_end(hc_ctx* c, double* force_out) {
    HC_API_BEGIN_HOT(c)
    require(force_out, HC_ERR_INVALID, "null pointer");
    try {
        step_end(c, force_out);
    } catch (...) {
        step_abort(c);
        throw;
    }
    HC_API_END(c)
}

// The result buffer of hc_step in memory the caller provides -- e.g. a POSIX shared-memory segment that the OTHER processes of a
// one-process-per-GPU host map too: every process then collects the force rows of all shards straight from the buffers the GPUs
// write (hc_wait_result_buffer), a host gather without a collective or a copy ("outputs -> host gather").
int hc_set_result_buffer(hc_ctx* c, void* host_buffer, size_t bytes) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(c->pending_step == 0, HC_ERR_INVALID, "a step is pending");
    HC_HIP(hipDeviceSynchronize());
    if (c->ext_tag_host) {
        (void)hipHostUnregister(c->ext_tag_host);
        c->ext_tag_host = c->ext_tag_dev = nullptr;
    }
    if (host_buffer) {
        const size_t need = static_cast<size_t>(4) * c->Dloc * sizeof(unsigned long long);
        require(bytes >= need, HC_ERR_INVALID, "result buffer too small: 2 x 16 bytes per owned row");
        require((reinterpret_cast<uintptr_t>(host_buffer) & 15) == 0, HC_ERR_INVALID, "result buffer must be 16-byte aligned");
        HC_HIP(hipHostRegister(host_buffer, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
        void* dp = nullptr;
        const hipError_t e = hipHostGetDevicePointer(&dp, host_buffer, 0);
        if (e != hipSuccess) {
            (void)hipHostUnregister(host_buffer);
            throw Error(HC_ERR_DEVICE, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
        }
        std::memset(host_buffer, 0, need);
        c->ext_tag_host = static_cast<unsigned long long*>(host_buffer);
        c->ext_tag_dev  = static_cast<unsigned long long*>(dp);
    }
    HC_API_END(c)
}

int hc_step_sequence(const hc_ctx* c, unsigned long long* seq) {
    if (!c || !seq) return HC_ERR_INVALID;
    *seq = c->seq;
    return HC_OK;
}

// Host-only: waits until the `rows` tagged results of step `seq` have arrived in a result buffer (this process's or another's)
// and copies the values out.  No context, no HIP call.
int hc_wait_result_buffer(const void* host_buffer, int rows, unsigned long long seq, double* out, double timeout_seconds) {
    if (!host_buffer || rows <= 0 || !out) return HC_ERR_INVALID;
    const volatile unsigned long long* g = static_cast<const unsigned long long*>(host_buffer) + (seq & 1) * static_cast<size_t>(2) * rows;
    unsigned long long spins = 0;
    std::chrono::steady_clock::time_point t0{};
    const double limit = timeout_seconds > 0.0 ? timeout_seconds : step_timeout_seconds();
    for (int r = rows - 1; r >= 0; --r) {
        while (g[2 * r + 1] != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
                const auto now = std::chrono::steady_clock::now();
                if (spins == 0x10000) t0 = now;
                if (std::chrono::duration<double>(now - t0).count() > limit) return HC_ERR_DEVICE;
            }
        }
    }
    for (int r = 0; r < rows; ++r) {
        const unsigned long long bits = g[2 * r];
        std::memcpy(out + r, &bits, sizeof(double));
    }
    return HC_OK;
}

// One evaluation of a body-row-sharded system held by G contexts of ONE host process (the drop-in variant: the host
// holds all body states -> a state store per GPU -> host-side gather).  Three phases: (1) every context gets the state and its
// step kernel -- all G GPUs are working before the host does anything else; (2) the work later steps need is enqueued on each;
// (3) the host collects the tagged totals of each shard into its rows of the 6N vector.  No collective, no torch: the same
// kernels and the same per-shard arithmetic as hc_step, so the gathered vector is bitwise the unsharded one.
int hc_step_multi(hc_ctx* const* ctxs, int n_ctx, double t, const double* pos, const double* rpy, const double* linvel,
                  const double* angvel, double* force_out) {
    if (!ctxs || n_ctx <= 0 || !force_out) return HC_ERR_INVALID;
    for (int g = 0; g < n_ctx; ++g)
        if (!ctxs[g]) return HC_ERR_INVALID;
    int status = HC_OK;
    std::string message;
    auto guarded = [&](hc_ctx* c, auto&& fn) {
        try {
            HC_HIP(hipSetDevice(c->device));
            fn();
            return true;
        } catch (const Error& e) {
            if (status == HC_OK) { status = e.status; message = e.what(); }
        } catch (const std::out_of_range& e) {
            if (status == HC_OK) { status = HC_ERR_OUT_OF_RANGE; message = e.what(); }
        } catch (const std::exception& e) {
            if (status == HC_OK) { status = HC_ERR_RUNTIME; message = e.what(); }
        }
        step_abort(c);
        return false;
    };
    std::vector<char> begun(static_cast<size_t>(n_ctx), 0);
    for (int g = 0; g < n_ctx; ++g) {
        hc_ctx* c = ctxs[g];
        begun[g]  = guarded(c, [&] {
            require(c->N == ctxs[0]->N, HC_ERR_INVALID, "hc_step_multi: the contexts belong to different systems");
            step_begin(c, t, pos, rpy, linvel, angvel, true);
        });
    }
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) begun[g] = guarded(ctxs[g], [&] { enqueue_tail(ctxs[g]); });
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) guarded(ctxs[g], [&] { step_end(ctxs[g], force_out + static_cast<size_t>(6) * ctxs[g]->b0); });
    if (status != HC_OK)
        for (int g = 0; g < n_ctx; ++g) ctxs[g]->err = message;  // hc_last_error of any context of the group tells why
    return status;
}

int hc_step_device(hc_ctx* c, double t, const double* d_state, double* d_force_out, void* stream) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(d_state && d_force_out, HC_ERR_INVALID, "null pointer");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (c->have_prev_device && t == c->prev_time_device) {
        if (s != c->stream) {  // the totals may have been left by an hc_step, whose kernels ran on the context's stream
            HC_HIP(hipEventRecord(c->ev_fin, c->stream));
            HC_HIP(hipStreamWaitEvent(s, c->ev_fin, 0));
        }
        HC_HIP(hipMemcpyAsync(d_force_out, c->d_total.p, c->Dloc * sizeof(double), hipMemcpyDeviceToDevice, s));
        return HC_OK;
    }
    c->have_prev = false;  // the host-side cache of hc_step does not hold this step
    enqueue_step(c, t, d_state, d_force_out, s, StepFlags{});
    c->prev_time_device = t;
    c->have_prev_device = true;
    HC_API_END(c)
}

int hc_get_force_components(hc_ctx* c, double* hs, double* rad, double* waves) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    HC_HIP(hipDeviceSynchronize());
    const size_t nb = c->Dloc * sizeof(double);
    HC_HIP(hipMemcpyAsync(c->h_out.p, c->d_hs.p, nb, hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipMemcpyAsync(c->h_out.p + c->Dloc, c->d_rad.p, nb, hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipMemcpyAsync(c->h_out.p + 2 * c->Dloc, c->d_waves.p, nb, hipMemcpyDeviceToHost, c->stream));
    check_device_flag(c);  // synchronises
    if (hs) std::memcpy(hs, c->h_out.p, nb);
    if (rad) std::memcpy(rad, c->h_out.p + c->Dloc, nb);
    if (waves) std::memcpy(waves, c->h_out.p + 2 * c->Dloc, nb);
    HC_API_END(c)
}

// The three term-only entry points write to scratch outputs: the components and the cached total of the last full step
// (hc_get_force_components, the duplicate-time cache) stay what that step left.
int hc_compute_radiation(hc_ctx* c, double t, const double* linvel, const double* angvel, double* rad_out) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(linvel && angvel && rad_out, HC_ERR_INVALID, "null pointer");
    stage_state(c, nullptr, nullptr, linvel, angvel);
    StepFlags f;
    f.hs = false;
    f.waves = false;
    f.scratch_out = true;
    enqueue_step(c, t, c->d_state.p, nullptr, c->stream, f);
    HC_HIP(hipMemcpyAsync(c->h_out.p + c->Dloc, c->d_scratch.p + c->Dloc, c->Dloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    check_device_flag(c);
    std::memcpy(rad_out, c->h_out.p + c->Dloc, c->Dloc * sizeof(double));
    HC_API_END(c)
}

int hc_compute_hydrostatics(hc_ctx* c, const double* pos, const double* rpy, double* hs_out) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pos && rpy && hs_out, HC_ERR_INVALID, "null pointer");
    stage_state(c, pos, rpy, nullptr, nullptr);
    StepFlags f;
    f.rad = false;
    f.waves = false;
    f.scratch_out = true;
    enqueue_step(c, 0.0, c->d_state.p, nullptr, c->stream, f);
    HC_HIP(hipMemcpyAsync(c->h_out.p, c->d_scratch.p, c->Dloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(hs_out, c->h_out.p, c->Dloc * sizeof(double));
    HC_API_END(c)
}

int hc_compute_waves(hc_ctx* c, double t, double* waves_out) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(waves_out, HC_ERR_INVALID, "null pointer");
    stage_state(c, nullptr, nullptr, nullptr, nullptr);
    StepFlags f;
    f.hs = false;
    f.rad = false;
    f.scratch_out = true;
    enqueue_step(c, t, c->d_state.p, nullptr, c->stream, f);
    HC_HIP(hipMemcpyAsync(c->h_out.p + 2 * c->Dloc, c->d_scratch.p + 2 * c->Dloc, c->Dloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    check_device_flag(c);
    std::memcpy(waves_out, c->h_out.p + 2 * c->Dloc, c->Dloc * sizeof(double));
    HC_API_END(c)
}

int hc_set_lookahead(hc_ctx* c, int steps) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    HC_HIP(hipDeviceSynchronize());  // a pass of the previous depth may still be running
    c->lookahead = steps <= 0 ? 0 : (steps <= 16 ? 16 : hc::kLookahead);
    choose_conv_config(c);  // the pass chunking depends on the depth
    alloc_partials(c);
    c->plan      = hc::Plan{};
    HC_API_END(c)
}

int hc_direct_dispatch_active(const hc_ctx* c) { return (c && c->direct_ready) ? 1 : 0; }
const char* hc_dispatch_mode_reason(const hc_ctx* c) {
    if (!c) return "no context";
    if (!c->finalized) return "hc_finalize has not been called";
    return c->direct_ready ? "direct AQL dispatch" : c->direct_why.c_str();
}

int hc_reset_history(hc_ctx* c) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    HC_HIP(hipDeviceSynchronize());  // steps may still be running on a caller's stream (hc_step_device)
    c->times.clear();
    c->retired.clear();
    c->head = -1;
    c->have_last_stream = c->bg_pending = false;  // everything has run
    c->have_prev = c->have_prev_device = false;
    c->prev_time = c->prev_time_device = -1.0;
    c->plan = hc::Plan{};
    HC_API_END(c)
}

int hc_set_history(hc_ctx* c, int n, const double* times, const double* vel) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(n >= 0 && (n == 0 || (times && vel)), HC_ERR_INVALID, "bad history arguments");
    for (int k = 1; k < n; ++k) require(times[k] < times[k - 1], HC_ERR_INVALID, "history times must be strictly decreasing (newest first)");
    HC_HIP(hipDeviceSynchronize());  // steps may still be running on a caller's stream (hc_step_device)
    c->have_last_stream = c->bg_pending = false;  // everything has run
    if (n > c->Hcap) ring_alloc(c, n + 16 + hc::kRewindSlack);
    c->times.assign(times, times + n);
    c->retired.clear();
    // sample k -> slot n-1-k, head = n-1
    std::vector<double> tt(n), vv(static_cast<size_t>(n) * c->D);
    for (int k = 0; k < n; ++k) {
        tt[n - 1 - k] = times[k];
        std::copy(vel + static_cast<size_t>(k) * c->D, vel + static_cast<size_t>(k + 1) * c->D, vv.begin() + static_cast<size_t>(n - 1 - k) * c->D);
    }
    if (n) {
        // on the context's stream: ring_alloc's memsets are queued there, and a copy on the null stream is not ordered
        // against a non-blocking stream (it could be overtaken by them)
        HC_HIP(hipMemcpyAsync(c->d_ring_t.p, tt.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(c->d_ring_v.p, vv.data(), vv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hc::launch_ring_transpose(c->d_ring_v.p, c->Hcap, c->HcapT, c->D, c->d_ring_vT.p, c->stream);
        HC_HIP(hipGetLastError());
    }
    HC_HIP(hipStreamSynchronize(c->stream));  // tt / vv are released on return
    c->head      = n - 1;
    c->plan      = hc::Plan{};
    // No step has been evaluated at times[0], so the per-time cache holds nothing (a step at exactly that time is the
    // reference's duplicate-time error, raised by the history push).
    c->have_prev = c->have_prev_device = false;
    c->prev_time = c->prev_time_device = -1.0;
    HC_API_END(c)
}

int hc_get_history(hc_ctx* c, int* n, double* times, double* vel) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    HC_HIP(hipDeviceSynchronize());
    const int H = static_cast<int>(c->times.size());
    if (n) *n = H;
    if (times) std::copy(c->times.begin(), c->times.end(), times);
    if (vel) {
        for (int k = 0; k < H; ++k) {
            const int slot = ((c->head - k) % c->Hcap + c->Hcap) % c->Hcap;
            HC_HIP(hipMemcpy(vel + static_cast<size_t>(k) * c->D, c->d_ring_v.p + static_cast<size_t>(slot) * c->D, c->D * sizeof(double),
                             hipMemcpyDeviceToHost));
        }
    }
    HC_API_END(c)
}

// ---- added mass -------------------------------------------------------------------------------
int hc_added_mass_matrix(hc_ctx* c, double* M) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(M, HC_ERR_INVALID, "null pointer");
    std::copy(c->ainf_host.begin(), c->ainf_host.end(), M);
    HC_API_END(c)
}

namespace {
// Chrono's integrator calls the product between force evaluations, so it is built like hc_step: staging buffers and a stream of
// its own (kernels of the last hc_step may still be reading the state buffer, and the work that step left for later steps
// is still running on the context's stream -- the product does not wait for it); w and the incoming R go to the device
// through the BAR (fallback: mapped pinned memory), one launch, the result comes back as tagged granules.
void added_mass_begin(hc_ctx* c, const double* w, double cc, const double* R, int n_sys) {
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(w && R, HC_ERR_INVALID, "null pointer");
    require(n_sys >= c->D, HC_ERR_INVALID, "system has fewer coordinates than the added-mass block");
    require(c->pending_am == 0, HC_ERR_INVALID, "an added-mass product is still in flight");
    if (c->lost) throw Error(HC_ERR_DEVICE, "the device stopped answering in an earlier step: " + c->direct_why);
    const int row0 = 6 * c->b0;
    const bool bar = c->bar_am.host_ok && c->bar_state.host_ok;  // bar_state.host_ok also carries the coherence check of hc_finalize
    double* hw       = bar ? c->bar_am.p : c->h_am.p;
    double* hr       = hw + c->D;
    std::memcpy(hw, w, c->D * sizeof(double));
    std::memcpy(hr, R + row0, c->Dloc * sizeof(double));
    if (bar) _mm_sfence();
    const double* dw = bar ? c->bar_am.p : c->h_am.dp;
    const unsigned long long seq = ++c->seq_am;
    if (c->direct_ready && bar && c->am_lane == 0) {
        // first product of this context: the second lane (a queue of its own) is created and self-tested now
        std::string why;
        bool abandon = false;
        c->am_lane   = (c->dq->ensure_lane(1, &why) && direct_selftest_rewrites(c, c->dq, 1, &abandon)) ? 1 : -1;
        c->direct_why.clear();  // (the step path's lane stays in use whatever the second lane's test said)
    }
    if (c->direct_ready && bar && c->am_lane == 1) {
        // the second lane of the direct queue: an AQL packet instead of a HIP launch, independent of the step path's lane
        hc::AddedMassArgs a{c->d_ainf.p, c->Dloc, c->D, dw, dw + c->D, cc, c->h_tag_am.dp, seq};
        c->dq->dispatch(c->dk_added_mass, static_cast<uint32_t>((c->Dloc + 3) / 4), 256, 0, &a, sizeof a, -1, 0.0, 1);
        c->prof.direct_dispatches += 1;
        c->pending_am = 1;
    } else {
        hc::launch_added_mass_mv_tagged(c->d_ainf.p, c->Dloc, c->D, dw, dw + c->D, cc, c->h_tag_am.dp, seq, c->stream_am);
        c->prof.hip_launches += 1;
        HC_HIP(hipGetLastError());
        c->pending_am = 2;
    }
}
void added_mass_end(hc_ctx* c, double* R) {
    const int how = c->pending_am;
    c->pending_am = 0;
    if (how == 1) wait_tagged(c, c->h_tag_am.p, c->seq_am, nullptr, R + 6 * c->b0, 1);
    else if (how == 2) wait_tagged(c, c->h_tag_am.p, c->seq_am, c->stream_am, R + 6 * c->b0);
}
}  // namespace

int hc_added_mass_mv(hc_ctx* c, const double* w, double cc, double* R, int n_sys) {
    HC_API_BEGIN_HOT(c)  // own stream, own buffers: independent of whatever the step queues still run
    try {
        added_mass_begin(c, w, cc, R, n_sys);
        added_mass_end(c, R);
    } catch (...) {
        c->pending_am = 0;
        throw;
    }
    HC_API_END(c)
}

// LoadIntLoadResidual_Mv of a row-sharded system held by G contexts of one process: every shard's product is handed to its GPU
// first, then the rows are collected (each shard owns rows [6*b0, 6*b1) of R; w is the full vector).
int hc_added_mass_mv_multi(hc_ctx* const* ctxs, int n_ctx, const double* w, double cc, double* R, int n_sys) {
    if (!ctxs || n_ctx <= 0) return HC_ERR_INVALID;
    for (int g = 0; g < n_ctx; ++g)
        if (!ctxs[g]) return HC_ERR_INVALID;
    int status = HC_OK;
    std::string message;
    auto guarded = [&](hc_ctx* c, auto&& fn) {
        try {
            HC_HIP(hipSetDevice(c->device));
            fn();
            return true;
        } catch (const Error& e) {
            if (status == HC_OK) { status = e.status; message = e.what(); }
        } catch (const std::exception& e) {
            if (status == HC_OK) { status = HC_ERR_RUNTIME; message = e.what(); }
        }
        c->pending_am = 0;
        return false;
    };
    std::vector<char> begun(static_cast<size_t>(n_ctx), 0);
    for (int g = 0; g < n_ctx; ++g) begun[g] = guarded(ctxs[g], [&] { added_mass_begin(ctxs[g], w, cc, R, n_sys); });
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) guarded(ctxs[g], [&] { added_mass_end(ctxs[g], R); });
    if (status != HC_OK)
        for (int g = 0; g < n_ctx; ++g) ctxs[g]->err = message;
    return status;
}

// ---- introspection ----------------------------------------------------------------------------
int hc_enable_profiling(hc_ctx* c, int on) {
    HC_API_BEGIN(c)
    if (!on) profile_drain(c);
    c->profiling       = on != 0;
    c->profile_stride  = on > 1 ? on : 1;
    c->profile_counter = 0;
    HC_API_END(c)
}

int hc_get_profile(hc_ctx* c, hc_profile_stats* out) {
    HC_API_BEGIN(c)
    require(out, HC_ERR_INVALID, "null pointer");
    profile_drain(c);
    *out = c->prof;
    HC_API_END(c)
}

int hc_reset_profile(hc_ctx* c) {
    HC_API_BEGIN(c)
    profile_drain(c);
    const double bytes = c->prof.conv_kernel_bytes, bbytes = c->prof.block_kernel_bytes, obytes = c->prof.block_kernel_bytes_once;
    c->prof = hc_profile_stats{};
    c->prof.conv_kernel_bytes       = bytes;
    c->prof.block_kernel_bytes      = bbytes;
    c->prof.block_kernel_bytes_once = obytes;
    HC_API_END(c)
}

int hc_get_sizes(hc_ctx* c, int* N, int* n_local, int* S, int* L, int* nf, int* nt, int* H, int* Hcap) {
    HC_API_BEGIN(c)
    if (N) *N = c->N;
    if (n_local) *n_local = c->nloc;
    if (S) *S = c->S;
    if (L) *L = c->L;
    if (nf) *nf = c->nf;
    if (nt) *nt = c->nt;
    if (H) *H = static_cast<int>(c->times.size());
    if (Hcap) *Hcap = c->Hcap;
    HC_API_END(c)
}

int hc_get_rirf_width(hc_ctx* c, double* w) {
    HC_API_BEGIN(c)
    require(c->finalized && w, HC_ERR_INVALID, "not finalized or null pointer");
    std::copy(c->width.begin(), c->width.end(), w);
    HC_API_END(c)
}

int hc_get_rirf_effective(hc_ctx* c, double* out) {
    HC_API_BEGIN(c)
    require(c->finalized && out, HC_ERR_INVALID, "not finalized or null pointer");
    ensure_processed(c);
    const size_t n = static_cast<size_t>(c->Dloc) * c->D * c->S;
    hc::DeviceBuffer<double> tmp;
    tmp.alloc(n);
    hc::launch_unrelayout(rad_panel(c), c->Dloc, c->D, c->S, tmp.p, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipMemcpyAsync(out, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

int hc_get_excitation_irf_resampled(hc_ctx* c, int body, double* t, double* width, double* vals) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular, HC_ERR_INVALID, "no irregular wave model attached");
    check_body(c, body);
    require(is_local(c, body), HC_ERR_INVALID, "body is not owned by this context");
    const hc::ExGroup& g = c->ex_groups[c->ex_group_of[body]];
    if (t) std::copy(c->ex_tau.begin() + g.off, c->ex_tau.begin() + g.off + g.L, t);
    if (width) std::copy(c->ex_width.begin() + g.off, c->ex_width.begin() + g.off + g.L, width);
    if (vals)
        for (int d = 0; d < 6; ++d) {
            const size_t off = static_cast<size_t>(6 * (body - c->b0) + d) * c->L + g.off;
            std::copy(c->ex_vals.begin() + off, c->ex_vals.begin() + off + g.L, vals + static_cast<size_t>(d) * g.L);
        }
    HC_API_END(c)
}

int hc_get_excitation_irf_size(hc_ctx* c, int body, int* L) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular && L, HC_ERR_INVALID, "no irregular wave model attached, or null pointer");
    check_body(c, body);
    require(c->ex_group_of[body] >= 0, HC_ERR_INVALID, "no excitation IRF was ingested for this body");
    *L = c->ex_groups[c->ex_group_of[body]].L;
    HC_API_END(c)
}

int hc_get_shard(hc_ctx* c, int* body_begin, int* body_end) {
    if (!c) return HC_ERR_INVALID;
    if (body_begin) *body_begin = c->b0;
    if (body_end) *body_end = c->b1;
    return HC_OK;
}

int hc_get_spectrum(hc_ctx* c, double* f, double* S, double* df, double* phase, double* k) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    if (f) std::copy(c->spec_f.begin(), c->spec_f.end(), f);
    if (S) std::copy(c->spec_S.begin(), c->spec_S.end(), S);
    if (df) std::copy(c->spec_df.begin(), c->spec_df.end(), df);
    if (phase) std::copy(c->spec_phase.begin(), c->spec_phase.end(), phase);
    if (k) std::copy(c->spec_k.begin(), c->spec_k.end(), k);
    HC_API_END(c)
}

int hc_get_eta_table(hc_ctx* c, double* t, double* eta) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    if (t) std::copy(c->eta_t.begin(), c->eta_t.end(), t);
    if (eta) std::copy(c->eta.begin(), c->eta.end(), eta);
    HC_API_END(c)
}

int hc_get_regular_coeffs(hc_ctx* c, double* mag, double* phase, double* wavenumber) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveRegular, HC_ERR_INVALID, "no regular wave model attached");
    if (mag) std::copy(c->reg_mag.begin(), c->reg_mag.end(), mag);
    if (phase) std::copy(c->reg_phase.begin(), c->reg_phase.end(), phase);
    if (wavenumber) *wavenumber = c->reg_wavenumber;
    HC_API_END(c)
}

// ---- synthetic inputs generated in HBM --------------------------------------------------------
int hc_synth_fill(hc_ctx* c, unsigned long long seed, int S, double dt_rirf, int n_exc, double dt_exc) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    require(S > 0 && dt_rirf > 0, HC_ERR_INVALID, "bad synthetic sizes");
    if (!c->have_sim) {
        c->rho = 1000.0;
        c->g = 9.81;
        c->depth = std::numeric_limits<double>::infinity();
        c->have_sim = true;
    }
    c->S = S;
    c->tau.resize(S);
    for (int s = 0; s < S; ++s) c->tau[s] = s * dt_rirf;
    setup_panel_geometry(c);
    c->dK.alloc(hc::panel_doubles(c->ntiles, c->ngp));
    hc::launch_synth_rirf(c->dK.p, c->ntiles, c->ngp, c->Dloc, c->D, S, 6 * c->b0, dt_rirf, seed, c->rho, c->stream);
    HC_HIP(hipGetLastError());
    // small per-body tables from the same counter-based stream, on the host
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    auto u01 = [](uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); };
    for (int b = 0; b < c->N; ++b) {
        auto& bd = c->bodies[b];
        const uint64_t hb = mix(seed ^ (0xB0D1ull << 40) ^ static_cast<uint64_t>(b));
        bd.disp_vol = 200.0 + 100.0 * u01(mix(hb + 1));
        for (int k = 0; k < 3; ++k) {
            bd.cg[k] = (k == 2 ? -2.0 : 20.0 * (b % 8) * (k == 0) + 20.0 * (b / 8) * (k == 1));
            bd.cb[k] = bd.cg[k] + (k == 2 ? 0.1 : 0.0);
        }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                const double r = u01(mix(hb + 100 + 6 * std::min(i, j) + std::max(i, j)));
                bd.lin[6 * i + j] = (i == j ? 50.0 + 50.0 * r : 2.0 * (r - 0.5));
            }
        bd.have_props = bd.have_lin = bd.have_ainf = bd.have_rirf = true;
        if (is_local(c, b)) {
            bd.ainf.resize(static_cast<size_t>(6) * c->D);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < c->D; ++j) {
                    const int gi = 6 * b + i;
                    const uint64_t h = mix(seed ^ (0xA1ull << 48) ^ (static_cast<uint64_t>(std::min(gi, j)) << 24) ^ static_cast<uint64_t>(std::max(gi, j)));
                    const double r = u01(h);
                    bd.ainf[static_cast<size_t>(i) * c->D + j] = c->rho * (gi == j ? 100.0 + 50.0 * r : (r - 0.5) * (j / 6 == b ? 5.0 : 0.5));
                }
            if (n_exc > 0) {
                bd.exirf_f.resize(static_cast<size_t>(6) * n_exc);
                for (int i = 0; i < 6; ++i) {
                    const uint64_t h = mix(seed ^ (0xE7ull << 48) ^ static_cast<uint64_t>(6 * b + i));
                    const double a = 1.0 + u01(mix(h + 1)), wd = 1.0 + 2.0 * u01(mix(h + 2)), om = 0.5 + 1.5 * u01(mix(h + 3));
                    for (int j = 0; j < n_exc; ++j) {
                        const double tt = (j - (n_exc - 1) * 0.5) * dt_exc;
                        bd.exirf_f[static_cast<size_t>(i) * n_exc + j] = c->rho * c->g * a * std::exp(-(tt * tt) / (wd * wd)) * std::cos(om * tt);
                    }
                }
            }
        }
        if (n_exc > 0) {
            bd.exirf_t.resize(n_exc);
            for (int j = 0; j < n_exc; ++j) bd.exirf_t[j] = (j - (n_exc - 1) * 0.5) * dt_exc;
            bd.have_exirf = true;
        }
    }
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

// Diagnostics (not part of the public header): copies an internal device buffer to the host.  which: 0 = P [16][Dpad],
// 1 = E [16][Dpad], 2 = Y [16][kScatterSamples][Dpad].
int hc_debug_read(hc_ctx* c, int which, double* out, long long n) {
    HC_API_BEGIN(c)
    HC_HIP(hipDeviceSynchronize());
    const hc::DeviceBuffer<double>& b = which == 0 ? c->d_P : (which == 1 ? c->d_E : c->d_Y);
    require(n >= 0 && static_cast<size_t>(n) <= b.n, HC_ERR_INVALID, "bad size");
    HC_HIP(hipMemcpy(out, b.p, static_cast<size_t>(n) * sizeof(double), hipMemcpyDeviceToHost));
    HC_API_END(c)
}

}  // extern "C"

// =================================================================================================
// include/hydrochrono_amd_host.h
// =================================================================================================
#include "../../include/hydrochrono_amd_host.h"

extern "C" {

void hc_host_linspaced(int n, double lo, double hi, double* out) {
    const auto v = hc::linspaced(n, lo, hi);
    std::copy(v.begin(), v.end(), out);
}
void hc_host_trapezoid_widths(const double* grid, int n, double* out) {
    const auto v = hc::trapezoid_widths(std::vector<double>(grid, grid + n));
    std::copy(v.begin(), v.end(), out);
}
void hc_host_jonswap_spectrum_hz(const double* f, int n, double Hs, double Tp, double gamma, int is_normalized, double* out) {
    const auto v = hc::jonswap_spectrum_hz(std::vector<double>(f, f + n), Hs, Tp, gamma, is_normalized != 0);
    std::copy(v.begin(), v.end(), out);
}
void hc_host_random_phases(int n, int seed, double* out) {
    const auto v = hc::random_phases(n, seed);
    std::copy(v.begin(), v.end(), out);
}
double hc_host_wave_number(double omega, double water_depth, double g) {
    try {
        return hc::wave_number(omega, water_depth, g);
    } catch (...) {
        return std::numeric_limits<double>::quiet_NaN();
    }
}
int hc_host_resample_irf(const double* vals, int n_old, int n_new, double* out) {
    try {
        const auto v = hc::resample_cubic_bspline6(std::vector<double>(vals, vals + static_cast<size_t>(6) * n_old), n_old, n_new);
        std::copy(v.begin(), v.end(), out);
    } catch (...) {
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

}  // extern "C"
