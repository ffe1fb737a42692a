// hc_step.cpp -- the per-step path behind the C ABI: look-ahead bookkeeping, the kernel launches / AQL dispatches of one evaluation
// (enqueue_step / enqueue_tail), the synchronous step and its halves, the multi-context step, the device-resident step, the
// term-only entry points, history access and the added-mass product.  All arithmetic runs in hc_kernels.hip on the GPU.
#include "hc_internal.hpp"
#include "hc_fanout.hpp"

using namespace hc::detail;

namespace hc {
namespace detail {

// (the per-step canary, see check_canary below)
bool canary_enabled() {
    static const bool on = HC_TUNE_INT("HC_STEP_CANARY", 1) != 0;  // (0: for A/B timing only -- 0.1 us of 8.7 / 11.9 us per hc_step at one / 64 bodies)
    return on;
}

// ---- the step ---------------------------------------------------------------------------------
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
    const int forced = HC_TUNE_INT("HC_SUB_BLOCK", -1);  // tests / tuning runs: 0 = single level, 4 / 8 = sub-block size
    if (forced >= 0) return forced;
    return hc::near_slices_for(c->D) > 1 ? hc::kSubBlock : 0;
}

bool make_plan(hc_ctx* c, bool own_zero = false) {
    const bool ok = hc::build_plan(c->plan, c->lookahead, c->times, c->tau, c->width, plan_sub_block(c), hc::near_slices_for(c->D), own_zero);
    if (ok) ++c->plan_serial;
    return ok;
}

// rows of the current block / of the block after it in d_P and d_E (two blocks of kLookahead rows each)
double* rows_P(hc_ctx* c, bool next) { return c->d_P.p + static_cast<size_t>(next ? 1 - c->pe_cur : c->pe_cur) * hc::kLookahead * c->Dpad; }
double* rows_E(hc_ctx* c, bool next) { return c->d_E.p + static_cast<size_t>(next ? 1 - c->pe_cur : c->pe_cur) * hc::kLookahead * c->Dpad; }

// The scatter of grid index m of the current plan (a block step's sample, pushed by its finalize_kernel; m = 0: the planning step's
// own sample of an own_zero plan): its results into the term slots of the steps they go to.
void launch_scatter_of(hc_ctx* c, int m, hipStream_t bs, bool direct) {
    const auto& pl = c->plan;
    hc::ScatterArgs sa{};
    sa.K     = rad_panel(c);
    sa.D     = c->D;
    sa.Dpad  = c->Dpad;
    sa.s_lo  = pl.scat_lo[m];
    sa.ns    = pl.scat_hi[m] - pl.scat_lo[m] + 1;
    sa.v     = c->d_ring_v.p + static_cast<size_t>(c->head) * c->D;  // the newest sample in the ring
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
    // pass schedule "one block ahead": the pass of the next block is in the making under this block's plan
    if (c->ahead.active && (!block || c->ahead.plan_serial != c->plan_serial)) ahead_drop(c);  // the block it belongs to was abandoned
    // (its windows of block samples end at the sub-block boundaries below L - 1 and at L - 1: the block's last sample is the next
    // block's own grid index 0, hc_plan.hpp: next_window_end)
    const bool window_end   = block && c->ahead.active && hc::next_window_end(c->plan, c->lookahead, m);
    const int window_kw     = window_end ? hc::next_window_length(c->plan, c->lookahead, m) : 0;
    bool window_done        = false;
    if (window_end && c->ahead.concurrent) {
        // a window of this block's samples ends here: what they contribute to the steps of the next block goes to the pass lane, behind
        // the step kernel that has just pushed the window's last sample (and before this step's own scatter / short pass on lane 0)
        const uint64_t h = direct ? c->dq->signal_after(0) : 0;
        if (h != 0) {
            c->dq->wait_for(2, h);
            launch_mini_pass(c, m, bs, true, window_kw, 2);
            window_done = true;
        } else {
            // (no direct dispatch for this step, or no signal: the pass lane is emptied and the short pass follows on the step path)
            pass_lane_drain(c);
        }
    }
    static const bool skip_scatter = HC_TUNE_INT("HC_SKIP_SCATTER", 0) != 0;  // (tuning build: timing bound only -- the forces are wrong)
    if (scatter_now && skip_scatter) {
    } else if (scatter_now) {
        to_background();
        launch_scatter_of(c, m, bs, direct);
    } else if (block && c->plan.sub > 0 && m < c->lookahead && m % c->plan.sub == 0 && c->plan.mini_s_hi[m] >= 0) {
        to_background();
        launch_mini_pass(c, m, bs, direct);  // two-level form: the sub-block that ends here -> the block steps still to come
    }
    if (block && c->ahead.active) {
        to_background();
        if (window_end && !window_done) {
            if (c->ahead.reduced) launch_mini_pass(c, m, bs, direct, window_kw);
            else ahead_drop(c);  // (cannot happen: the slices end within the first window)
        }
        if (m < c->lookahead) ahead_issue_slice(c, bs, direct);
    }
    if (plan_now) {
        const unsigned long long ended = c->plan_serial;
        const bool clean_end           = block && m == c->lookahead;
        if (block) c->plan.misses = 0;  // a block was consumed completely
        // Rows made ahead exist for the block that starts now: it is planned with this step's sample as its own grid index 0 (the
        // short passes that completed the rows stopped one sample earlier), and takes the rows if the plan comes out as predicted.
        const bool rows_ahead = clean_end && ahead_expected(c, ended);
        bool planned = make_plan(c, rows_ahead);
        // (not as predicted after all, or no room for the extra scatter terms: an ordinary block with a pass of its own)
        if (rows_ahead && (!planned || !ahead_adoptable(c, ended))) planned = make_plan(c, false);
        if (planned) {
            to_background();
            if (c->plan.own_zero) {
                // the rows of this block are there already: no pass now.  Made on the pass lane: the steps of this block wait for it.
                if (c->ahead.concurrent && c->dq && c->dq->busy(2)) {
                    const uint64_t h2 = direct ? c->dq->signal_after(2) : 0;
                    if (h2 != 0) c->dq->wait_for(0, h2);
                    else pass_lane_drain(c);
                }
                c->pe_cur ^= 1;
                c->plan.has_exc = c->ahead.has_exc;
                c->prof.ahead_blocks += 1;
                if (c->plan.scat_hi[0] >= c->plan.scat_lo[0]) launch_scatter_of(c, 0, bs, direct);  // this step's sample -> the block's steps
            } else {
                launch_pass(c, bs, c->tail.waves, direct);
            }
            ahead_begin(c, bs, c->tail.waves, direct);
        } else {
            ahead_drop(c);
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
// The classic home of a synchronous step's state: device memory written through the PCIe BAR (fallback: mapped pinned memory the
// kernels read over PCIe), two halves used alternately -- hc_step returns as soon as the totals have arrived, while the workgroup that
// stores the step's sample into the ring may still be reading the state, so the next call must not overwrite it.  (The kernels of step
// n + 1 run after those of step n, and step n + 2 starts only after the totals of step n + 1 have arrived: two halves are enough.)
// A row-sharded context reads positions and angles of its OWN bodies only (hydrostatics), velocities of all: the other bodies' pos / rpy
// entries are not stored (half the bytes through the BAR for each shard of a wide array).  `seq`: the step's sequence number.
const double* stage_host_state(hc_ctx* c, const HostState& hs, unsigned long long seq) {
    const int n3   = 3 * c->N;
    const size_t o = (seq & 1) ? 0 : static_cast<size_t>(12) * c->N + 1;
    const size_t l0 = static_cast<size_t>(3) * c->b0, ln = static_cast<size_t>(3) * c->nloc;
    auto put = [&](double* h) {
        std::memcpy(h + l0, hs.pos + l0, ln * sizeof(double));
        std::memcpy(h + n3 + l0, hs.rpy + l0, ln * sizeof(double));
        std::memcpy(h + 2 * n3, hs.linvel, n3 * sizeof(double));
        std::memcpy(h + 3 * n3, hs.angvel, n3 * sizeof(double));
    };
    const double* d_state;
    if (c->bar_state.host_ok) {
        // device memory written through the PCIe BAR: the kernels read the state locally (no PCIe read on the critical path)
        double* h = c->bar_state.p + o;
        put(h);
        h[4 * n3] = hs.canary;
        _mm_sfence();  // write-combined stores are globally visible before the doorbell of the launch
        d_state = h;
        c->step_canary_in = h + 4 * n3;
    } else {
        double* h = c->h_state.p + o;
        put(h);
        h[4 * n3] = hs.canary;
        d_state = c->h_state.dp + o;
        c->step_canary_in = c->h_state.dp + o + 4 * n3;
        if (c->N > c->zero_copy_max_bodies) {  // many workgroups re-read the state: one small H2D copy beats their PCIe reads
            HC_HIP(hipMemcpyAsync(c->d_state.p, h, 4 * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            d_state = c->d_state.p;
        }
    }
    if (!canary_enabled()) c->step_canary_in = nullptr;
    return d_state;
}

// ... and its place behind the argument block of the step kernel (DirectQueue::FillExtra; layout of hc_limits.hpp: velocities by DoF
// column, positions, angles, canary word).  Every dispatch has a slot of its own, so nothing is overwritten under a reader.
struct SlotStateFill {
    const hc_ctx* c;
    const HostState* hs;
};
void fill_slot_state(char* extra, void* user) {
    const SlotStateFill& w = *static_cast<const SlotStateFill*>(user);
    const int N = w.c->N, n3 = 3 * N;
    double* h = reinterpret_cast<double*>(extra);
    for (int b = 0; b < N; ++b) {
        std::memcpy(h + 6 * b, w.hs->linvel + 3 * b, 3 * sizeof(double));
        std::memcpy(h + 6 * b + 3, w.hs->angvel + 3 * b, 3 * sizeof(double));
    }
    const size_t l0 = static_cast<size_t>(3) * w.c->b0, ln = static_cast<size_t>(3) * w.c->nloc;
    std::memcpy(h + 2 * n3 + l0, w.hs->pos + l0, ln * sizeof(double));
    std::memcpy(h + 3 * n3 + l0, w.hs->rpy + l0, ln * sizeof(double));
    h[4 * n3] = w.hs->canary;
}

// ---- one evaluation = StepJob: what enqueue_step decides once, and the pieces that act on it -----------------------------------
// The four shapes of a step (round-5 review: one 330-line function held all of them):
//   plain          no look-ahead block (first steps, a time off the predicted grid, look-ahead off): conv_step_kernel over all live
//                  columns of K (+ the excitation chunks), then the step kernel adds the chunk partials        launch_plain_convolution
//   block step     inside a block: ONE step kernel -- own IRF samples, look-ahead row, scatter results          add_block_part,
//                  (step_hot_kernel for the common shape, finalize_kernel otherwise)                            dispatch_step_kernel
//   wide           6N >= 1024: the own-sample part split over column slices, fused with the step kernel
//                  (wide_step_kernel) or as near_split_kernel + finalize_kernel                                 dispatch_wide_step
//   device path    hc_step_device (state and result in HBM, a caller's stream): the same launches through HIP  (direct == false)
// host_state (hc_step): the caller's state has NOT been stored anywhere yet (d_state is null).  A step whose one kernel is the step
// kernel of the direct path takes it behind that kernel's argument block (fill_slot_state); every other step stores it the classic way
// first (stage_host_state) -- decided where the step's shape is known.
struct StepJob {
    hc_ctx* c;
    double t;
    const double* d_state;
    double* d_user_out;
    hipStream_t stream;
    StepFlags f;
    unsigned long long* host_tagged;
    unsigned long long seq;
    const HostState* host_state;
    int H = 0, m = 0;
    bool run_rad = false, run_exc = false, block = false, direct = false, caller_waits = false;
    const double *P_row = nullptr, *E_row = nullptr;
    int nchunks_rad = 0, nchunks_ex = 0;
    StepViews vw{};
    // (a state still in the caller's hands goes to its classic place as soon as a kernel other than the direct step kernel wants it)
    const double* staged_state() {
        if (!d_state && host_state) d_state = stage_host_state(c, *host_state, seq);
        return d_state;
    }
};

// A caller's stream that is idle now belongs to a caller that waits for every step (the force exchange of a row-sharded array): the
// work later steps need then goes to the context's own stream (enqueue_tail).  A caller that runs ahead of the GPU (stream still busy)
// gets everything on its stream in order -- the two event hops per step would only slow it down.  And: the steps of a context normally
// stay on one stream; when they move, this step is ordered behind the previous one with an event.
void order_streams(StepJob& j) {
    hc_ctx* c = j.c;
    const hipStream_t stream = j.stream;
    if (stream != c->stream && j.f.rad && c->lookahead > 0) {
        if (c->busy_caller_steps > 0) {
            --c->busy_caller_steps;  // found busy a moment ago: do not pay for the query on every step of a caller that runs ahead
        } else {
            const hipError_t q = hipStreamQuery(stream);
            j.caller_waits     = q == hipSuccess;
            if (q != hipSuccess) {
                (void)hipGetLastError();
                c->busy_caller_steps = 15;
            }
        }
    }
    if (c->have_last_stream && c->last_stream != stream) {
        // hc_step after hc_step_device on a caller's stream, or the reverse.  A caller's stream may have been destroyed since (after a
        // synchronise): then there is nothing left to wait for.
        if (hipEventRecord(c->ev_fin, c->last_stream) == hipSuccess) HC_HIP(hipStreamWaitEvent(stream, c->ev_fin, 0));
        else (void)hipGetLastError();
    }
    c->last_stream      = stream;
    c->have_last_stream = true;
    if (c->bg_pending) {
        // the scatter / pass of the previous step ran on the context's own stream: this step's kernels need them
        if (stream != c->stream) HC_HIP(hipStreamWaitEvent(stream, c->ev_bg, 0));
        c->bg_pending = false;
    }
}

// Where this step's kernels go: the direct queue (hc_direct.hpp) when the step comes from hc_step and needs no plain convolution
// launch the direct queue cannot take -- the steady state of a look-ahead run -- else the HIP stream.  Nothing orders the two against
// each other on the device, so the side that was used last is drained at a switch.
void route_step(StepJob& j) {
    hc_ctx* c = j.c;
    if (c->tile_counter_suspect) {
        // A step failed after its kernels may have gone out (step_abort): a wide_step_kernel that was cut short leaves its tiles'
        // arrival counters non-zero, and no workgroup of a later launch would find itself last.  Everything the context has in
        // flight is waited for, then the counters start from zero again.  BEFORE the routing decision below: quiesce_direct leaves
        // the context on the HIP side (path 1), and a step that then goes out on the direct queue must find path == 2 when it waits
        // (wait_tagged asks the stream otherwise, finds it idle and declares the device lost).
        quiesce_direct(c);
        HC_HIP(hipStreamSynchronize(c->stream));
        HC_HIP(hipMemset(c->d_tile_counter.p, 0, c->d_tile_counter.n * sizeof(int)));
        c->tile_counter_suspect = false;
    }
    j.direct = c->direct_ready && j.host_tagged && j.stream == c->stream && !j.f.scratch_out &&
               (c->dk_step.ok() || !((j.run_rad && !j.block) || j.nchunks_ex > 0)) &&
               !(c->profiling && profiling_tool_attached());  // the library's own timings under a tool: HIP events
    if (j.direct && c->path != 2) {
        // switching from HIP launches to the direct queue: everything the HIP side still runs must have finished.  A preceding
        // step on a caller's stream (hc_step_device) has been ordered in front of c->stream by the ev_fin wait of order_streams, and
        // what it left on the context's own stream (ev_bg) runs there too, so draining c->stream covers both.
        HC_HIP(hipStreamSynchronize(c->stream));
        c->bg_pending = false;
        c->path       = 2;
    } else if (!j.direct) {
        quiesce_direct(c);
        c->path = 1;
    }
}

// plain step: all live columns of K, plus the excitation chunks unless a pass has left the excitation force
void launch_plain_convolution(StepJob& j) {
    hc_ctx* c = j.c;
    hc::HistoryView hv{};
    hv.state   = j.staged_state();
    hv.N       = c->N;
    hv.D       = c->D;
    hv.t       = j.t;
    hv.ring_t  = c->d_ring_t.p;
    hv.ring_v  = c->d_ring_v.p;
    hv.head    = c->head;
    hv.H       = j.H;
    hv.Hcap    = c->Hcap;
    hv.HcapT   = c->HcapT;
    hv.dt_hint = (j.H >= 2 && c->times[0] > c->times[1]) ? (c->times[0] - c->times[1]) : 1.0;
    hc::StepArgs a{};
    a.K       = rad_panel(c);
    a.F_limit = (j.run_rad && !j.block) ? std::min(c->S, live_samples(c, j.t)) * c->D : 0;
    a.chunk_gp = c->chunk_gp;
    j.nchunks_rad         = ((a.F_limit + 7) / 8 + a.chunk_gp - 1) / a.chunk_gp;
    a.nchunks_rad         = j.nchunks_rad;
    a.max_steps_per_chunk = (a.chunk_gp * 8) / c->D + 2;
    a.rhs_capacity        = 8 * std::max(a.chunk_gp, c->chunk_gp_ex);
    a.hist                = hv;
    a.tau                 = c->d_tau.p;
    a.width               = c->d_width.p;
    a.Kex                 = j.vw.kex;
    a.ex                  = j.vw.ex;
    a.chunk_gp_ex         = c->chunk_gp_ex;
    a.nchunks_ex          = j.nchunks_ex;
    a.partials            = c->d_partials.p;
    a.Dpad                = c->Dpad;
    a.ngroups             = c->ngroups;
    a.error_flag          = c->d_err.p;
    const double rad_b = 8.0 * (static_cast<double>(c->Dloc) * a.F_limit + a.F_limit);
    const double exc_b = j.nchunks_ex > 0 ? 8.0 * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
    const int kind     = j.nchunks_rad > 0 ? hc::kEvConvPlain : hc::kEvConvExc;
    const double share = exc_b / std::max(1.0, rad_b + exc_b);
    if (j.direct) {
        const hc::StepLaunch l = hc::step_launch_config(a, c->mt);
        if (l.nblocks > 0) {
            c->dq->dispatch(c->dk_step, static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &a, sizeof a, direct_tag(c, kind), share);
            c->prof.direct_dispatches += 1;
        }
    } else {
        hc::EventPair* ev = ev_begin(c, kind, j.stream, share);
        hc::launch_conv_step(a, c->mt, j.stream);
        c->prof.hip_launches += 1;
        ev_end(ev, j.stream);
    }
}

// A step inside a look-ahead block: the IRF samples it contracts itself (its own sample's, and the one sample per step the pass could
// not decide ahead of time while the history is shorter than the IRF window), and the scatter results of the block's earlier steps.
void add_block_part(const StepJob& j, hc::FinalizeArgs& z) {
    hc_ctx* c      = j.c;
    const auto& pl = c->plan;
    const int m    = j.m;
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
        if (host_bracket(c, j.t - c->tau[sd], j.H, &br) && (br.wo != 0.0 || br.wn != 0.0)) {
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

// the step kernel's arguments that do not depend on the step's shape
void fill_step_args(StepJob& j, hc::FinalizeArgs& z) {
    hc_ctx* c     = j.c;
    z.partials    = c->d_partials.p;
    z.nchunks_rad = j.nchunks_rad;
    z.nchunks_ex  = j.nchunks_ex;
    z.P           = j.P_row;
    z.E           = j.E_row;
    z.host_tagged = j.host_tagged;
    z.canary_out  = c->step_canary_out;  // (set by step_begin for this call only; canary_in goes with the state)
    c->step_canary_out = nullptr;
    z.seq         = j.seq;
    z.Dloc        = c->Dloc;
    z.Dpad        = c->Dpad;
    z.N           = c->N;
    z.b0          = c->b0;
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
    z.t             = j.t;
    z.do_hs         = j.f.hs;
    z.do_rad        = j.run_rad;
    z.do_waves      = j.f.waves;
    double* out4    = j.f.scratch_out ? c->d_scratch.p : nullptr;
    z.hs            = out4 ? out4 : c->d_hs.p;
    z.rad           = out4 ? out4 + c->Dloc : c->d_rad.p;
    z.waves         = out4 ? out4 + 2 * c->Dloc : c->d_waves.p;
    z.total         = out4 ? out4 + 3 * c->Dloc : c->d_total.p;
    z.user_out      = j.d_user_out;
    z.do_push       = j.f.rad ? 1 : 0;
    z.head          = c->head;
    z.D             = c->D;
    z.ring_t        = c->d_ring_t.p;
    z.ring_v        = c->d_ring_v.p;
    z.ring_vT       = c->d_ring_vT.p;
    z.Hcap          = c->Hcap;
    z.HcapT         = c->HcapT;
}

// the classic state pointer (and the canary word behind it) into the step kernel's arguments
void state_into_args(StepJob& j, hc::FinalizeArgs& z) {
    z.state              = j.staged_state();
    z.canary_in          = j.c->step_canary_in;
    j.c->step_canary_in  = nullptr;
}

// Wide system (6N >= 1024): the own-sample part is split over column slices by a kernel of its own (hundreds of workgroups instead of
// one per row tile) and the step kernel adds the slice partials -- in ONE launch where that is one round of workgroups
// (wide_step_kernel: the workgroup that completes a row tile's slices finishes the tile; returns true: the step is out), else as
// near_split_kernel here and the step kernel behind it (returns false: z is set up for it).
bool dispatch_wide_step(StepJob& j, hc::FinalizeArgs& z) {
    hc_ctx* c = j.c;
    state_into_args(j, z);
    hc::NearArgs na{};
    na.K      = z.nearK;
    na.D      = c->D;
    na.Dpad   = c->Dpad;
    na.N      = c->N;
    na.n_near = z.n_near;
    for (int e = 0; e < z.n_near; ++e) na.near[e] = z.near[e];
    na.state    = z.state;
    na.ring_v   = c->d_ring_v.p;
    na.partials = c->d_near_partials.p;
    // HC_WIDE_FUSED=0: the two launches (same arithmetic, bitwise).  Fused only while the launch is ONE round of workgroups (218 VGPRs:
    // two per CU): a C4/8 shard has 24 tiles x 8 slices = 192; the whole 512-body array on one GPU has 1536, runs them in three rounds
    // and is faster with the two launches (23 against 32 us).
    static const bool fused_on = HC_TUNE_INT("HC_WIDE_FUSED", 1) != 0;
    const long long wide_wgs   = static_cast<long long>(c->ntiles) * hc::near_slices_for(c->D);
    if (fused_on && !j.f.scratch_out && wide_wgs <= 2LL * c->num_cus && c->d_tile_counter.n >= static_cast<size_t>(c->ntiles) && (!j.direct || c->dk_wide.ok())) {
        hc::WideStepArgs w{na, z, c->d_tile_counter.p};
        if (j.direct) {
            const hc::WideLaunch l = hc::wide_launch_config(w);
            c->dq->dispatch(c->dk_wide, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &w, sizeof w, direct_tag(c, hc::kEvStep));
            c->prof.direct_dispatches += 1;
        } else {
            hc::EventPair* ev = ev_begin(c, hc::kEvStep, j.stream);
            hc::launch_wide_step(w, j.stream);
            ev_end(ev, j.stream);
            c->prof.hip_launches += 1;
        }
        c->prof.wide_fused_steps += 1;
        return true;
    }
    if (j.direct) {
        const hc::NearLaunch l = hc::near_launch_config(na);
        c->dq->dispatch(c->dk_near, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &na, sizeof na, direct_tag(c, hc::kEvStep));
        c->prof.direct_dispatches += 1;
    } else {
        hc::EventPair* ev = ev_begin(c, hc::kEvStep, j.stream);
        hc::launch_near_split(na, j.stream);
        ev_end(ev, j.stream);
        c->prof.hip_launches += 1;
        (void)hc::near_launch_config(na);
    }
    z.near_partials = c->d_near_partials.p;
    z.n_near_slices = na.n_slices;
    z.n_near        = 0;
    return false;
}

// The common block step -- the step's own IRF samples against its own velocity only, look-ahead row and scatter results there, no
// plain partials -- goes to the step kernel written for exactly that (step_hot_kernel: compact argument block, every load requested
// up front); everything else keeps the general one.  Same arithmetic in the same order (bitwise the same forces).
bool hot_step_eligible(const StepJob& j, const hc::FinalizeArgs& z) {
    bool hot = j.c->step_hot && j.block && z.P && z.Yc && z.nchunks_rad == 0 && z.nchunks_ex == 0 && z.n_near_slices == 0 && (z.n_near == 1 || z.n_near == 2) &&
               z.do_hs && z.do_rad && z.do_waves && !z.user_out && z.wave_mode != hc::kWaveSpectral && (z.wave_mode != hc::kWaveIrregular || z.E) &&
               (z.wave_mode != hc::kWaveRegular || z.reg_mag);
    for (int e = 0; hot && e < z.n_near; ++e) hot = z.near[e].b == 0.0 && z.near[e].c == 0.0;
    return hot;
}
hc::StepHotArgs hot_step_args(const hc_ctx* c, const hc::FinalizeArgs& z) {
    hc::StepHotArgs h{};
    for (int e = 0; e < 2; ++e) {
        const hc::NearEntry& ne = z.near[std::min(e, z.n_near - 1)];
        const int f0 = ne.s * c->D, g0 = f0 >> 3, g1 = (f0 + c->D + 7) >> 3;
        h.kfirst[e] = z.nearK.base + static_cast<size_t>(g0) * 128;
        h.ng[e]     = g1 - g0;
        h.off[e]    = f0 - 8 * g0;
        h.a[e]      = ne.a;
    }
    h.Yc = z.Yc; h.P = z.P; h.E = z.E ? z.E : z.P;
    h.lin = z.lin; h.cg = z.cg; h.cb_m_cg = z.cb_m_cg; h.disp_vol = z.disp_vol;
    h.reg_mag = z.reg_mag ? z.reg_mag : z.P;
    h.ngp = z.nearK.ngp; h.n_terms = z.n_terms; h.Dpad = z.Dpad; h.Dloc = z.Dloc; h.N = z.N; h.b0 = z.b0;
    h.halves = c->step_halves == 2 && c->D >= hc::kStepHalvesMinColumns ? 2 : 1;
    h.ntiles = c->ntiles * h.halves;
    h.D = c->D; h.wave_mode = z.wave_mode; h.has_E = z.E ? 1 : 0;
    h.rho = z.rho; h.gx = z.gx; h.gy = z.gy; h.gz = z.gz; h.t = z.t; h.reg_amplitude = z.reg_amplitude; h.reg_omega = z.reg_omega;
    for (int i = 0; i < 6; ++i) h.reg_phase[i] = z.reg_phase[i];
    h.seq = z.seq; h.hs = z.hs; h.rad = z.rad; h.waves = z.waves; h.total = z.total; h.host_tagged = z.host_tagged; h.canary_out = z.canary_out;
    h.ring_t = z.ring_t; h.ring_v = z.ring_v; h.ring_vT = z.ring_vT; h.head = z.head; h.Hcap = z.Hcap; h.HcapT = z.HcapT;
#ifdef HC_TUNING
    h.stamps = z.stamps;
#endif
    return h;
}

// The step kernel itself.  On the direct path, when nobody has needed the state so far (the step's ONE kernel): it travels behind the
// kernel's arguments, where the kernel can ask for it before it has read a single argument (step_hot_kernel / finalize_kernel<4, true>).
void dispatch_step_kernel(StepJob& j, hc::FinalizeArgs& z) {
    hc_ctx* c = j.c;
    if (j.direct && !j.d_state && j.host_state && c->slot_state) {
        SlotStateFill fill{c, j.host_state};
        hc::FinalizeLaunch l = hc::finalize_launch_config(z);
        const bool hot = hot_step_eligible(j, z);
        if (hot && c->step_halves == 2 && c->D >= hc::kStepHalvesMinColumns) l.grid = 2 * c->ntiles + 1;
#ifdef HC_TUNING
        if (c->stamps_on && l.grid <= hc::kStampWGs) {
            if (c->d_stamps.n == 0) {
                c->d_stamps.alloc(static_cast<size_t>(hc::kStampSteps) * hc::kStampWGs * hc::kStampStages);
                HC_HIP(hipMemset(c->d_stamps.p, 0, c->d_stamps.n * sizeof(unsigned long long)));
            }
            z.stamps = c->d_stamps.p + (j.seq % hc::kStampSteps) * hc::kStampWGs * hc::kStampStages;
        }
#endif
        if (hot) {
            const hc::StepHotArgs h = hot_step_args(c, z);
            const size_t lds = static_cast<size_t>(z.n_near) * (c->D + 16) * sizeof(double);  // (own samples' right-hand sides between zero pads)
            static const bool no_acquire = HC_TUNE_INT("HC_STEP_NO_ACQUIRE", 0) != 0;  // (tuning experiment: timing only, EXPERIMENTS.md round 6)
            c->dq->dispatch(c->dk_step_hot[z.n_near - 1], static_cast<uint32_t>(h.ntiles + 1), 256, static_cast<uint32_t>(lds), &h, sizeof h, direct_tag(c, hc::kEvStep), 0.0,
                            0, fill_slot_state, &fill, no_acquire);
            c->prof.hot_steps += 1;
        } else if (c->step_preload) {
            // (tuning experiment) the addresses of the step kernel's first loads in front of its argument block, where the packet processor preloads them
            hc::FinalizePreArgs pz{};
            const bool near_on = z.do_rad && z.n_near > 0;
            const int f0 = near_on ? z.near[0].s * c->D : 0, g0 = f0 >> 3, g1 = (f0 + c->D + 7) >> 3;
            pz.kfirst  = z.nearK.base + static_cast<size_t>(near_on ? g0 : 0) * 128;
            pz.ngroups = near_on ? g1 - g0 : 0;
            pz.ngp     = z.nearK.ngp;
            pz.yc      = z.Yc;
            pz.n_terms = (z.do_rad && z.Yc) ? z.n_terms : 0;
            pz.dpad    = c->Dpad;
            pz.ntiles  = c->ntiles;
            pz.a       = z;
            c->dq->dispatch(c->dk_finalize_pre, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &pz, sizeof pz, direct_tag(c, hc::kEvStep), 0.0, 0,
                            fill_slot_state, &fill);
        } else {
            c->dq->dispatch(c->dk_finalize_slot, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &z, sizeof z, direct_tag(c, hc::kEvStep), 0.0, 0,
                            fill_slot_state, &fill);
        }
        c->prof.direct_dispatches += 1;
        c->prof.slot_state_steps += 1;
#ifdef HC_TUNING
        if (z.stamps) {
            c->host_stamps[j.seq % hc::kStampSteps][1] = c->dq->system_ticks();  // doorbell of the step kernel
            c->host_stamps[j.seq % hc::kStampSteps][3] = static_cast<unsigned long long>(l.grid);
        } else if (c->stamps_on) {
            c->host_stamps[j.seq % hc::kStampSteps][3] = 0;
        }
#endif
    } else if (j.direct) {
        if (!z.state) state_into_args(j, z);
        const hc::FinalizeLaunch l = hc::finalize_launch_config(z);
        c->dq->dispatch(c->dk_finalize, static_cast<uint32_t>(l.grid), 256, static_cast<uint32_t>(l.smem), &z, sizeof z, direct_tag(c, hc::kEvStep));
        c->prof.direct_dispatches += 1;
    } else {
        if (!z.state) state_into_args(j, z);
        hc::EventPair* ev = ev_begin(c, hc::kEvStep, j.stream);
        hc::launch_finalize(z, j.stream);
        c->prof.hip_launches += 1;
        ev_end(ev, j.stream);
    }
}

// Enqueue the kernels of one evaluation at time t.  d_state: device-visible pointer to the 12N state (null with host_state: hc_step).
// d_user_out (device) and host_tagged (mapped pinned granules) may be null.  defer_tail: the caller enqueues the work later steps need
// itself (enqueue_tail) -- hc_step_multi, after all shard contexts have their step kernels on the way.
void enqueue_step(hc_ctx* c, double t, const double* d_state, double* d_user_out, hipStream_t stream, StepFlags f,
                  unsigned long long* host_tagged, unsigned long long seq, bool defer_tail, const HostState* host_state) {
    require(!c->tail.pending, HC_ERR_INVALID, "a step begun with hc_step_begin has not been completed (hc_step_end)");
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    const bool irregular = c->wave_kind == hc::kWaveIrregular;
    if (f.waves) check_wave_ready(c, t);
    profile_begin_step(c);
    StepJob j{c, t, d_state, d_user_out, stream, f, host_tagged, seq, host_state};
    if (f.rad) {
        ensure_processed(c);
        j.H = history_push(c, t);
        j.m = plan_step(c, t, j.H);
    }
    j.run_rad = f.rad && j.H >= 2;  // "Nothing to convolve with if we don't yet have at least 2 time points" (:580)
    j.run_exc = f.waves && irregular;
    j.vw      = make_views(c);
    j.block   = j.run_rad && j.m > 0;
    order_streams(j);
    j.P_row = j.block ? rows_P(c, false) + static_cast<size_t>(j.m - 1) * c->Dpad : nullptr;
    j.E_row = (j.block && j.run_exc && c->plan.has_exc) ? rows_E(c, false) + static_cast<size_t>(j.m - 1) * c->Dpad : nullptr;
    j.nchunks_ex = (j.run_exc && !j.E_row) ? c->nchunks_ex : 0;
    route_step(j);
    if (!host_state || !j.direct || !c->slot_state) j.staged_state();
    if ((j.run_rad && !j.block) || j.nchunks_ex > 0) launch_plain_convolution(j);

    hc::FinalizeArgs z{};
    fill_step_args(j, z);
    if (j.block) add_block_part(j, z);
    static const bool dbg = HC_TUNE_INT("HC_DEBUG_PLAN", 0) != 0;
    if (dbg) {
        std::fprintf(stderr, "[hc] t=%.6f H=%d m=%d n_near=%d n_terms=%d sd=%d nchunks_rad=%d nchunks_ex=%d head=%d\n", t, j.H, j.m, z.n_near, z.n_terms,
                     j.block ? c->plan.s_defer[j.m - 1] : -2, j.nchunks_rad, j.nchunks_ex, c->head);
        for (int e = 0; e < z.n_near; ++e)
            std::fprintf(stderr, "     near s=%d a=%.6g b=%.6g c=%.6g offb=%d offc=%d\n", z.near[e].s, z.near[e].a, z.near[e].b, z.near[e].c, z.near[e].off_b, z.near[e].off_c);
    }
    const bool wide = z.n_near > 0 && hc::near_slices_for(c->D) > 1;
    if (!(wide && dispatch_wide_step(j, z))) dispatch_step_kernel(j, z);

    // ---- off the caller's critical path: what later steps need from this one (enqueue_tail) ----
    c->tail              = hc::StepTail{};
    c->tail.pending      = f.rad && c->lookahead > 0;
    c->tail.rad          = f.rad;
    c->tail.waves        = f.waves;
    c->tail.block        = j.block;
    c->tail.direct       = j.direct;
    c->tail.caller_waits = j.caller_waits;
    c->tail.m            = j.m;
    c->tail.H            = j.H;
    c->tail.stream       = stream;
    if (!defer_tail) enqueue_tail(c);
    HC_HIP(hipGetLastError());

    if (f.hs) c->prof.hydrostatics_calls++;
    if (f.rad) c->prof.radiation_calls++;
    if (f.waves) c->prof.waves_calls++;
}

}  // namespace detail
}  // namespace hc

// =================================================================================================
extern "C" {

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

// The step kernel hands back the word the host stored behind this step's state (FinalizeArgs::canary_*): it must be this step's
// sequence number.  Anything else means the GPU read a copy of the state buffer that is older than what the host wrote through the
// BAR -- the one assumption the agent-scope fences of the direct dispatch rest on (hc_direct.hpp), tested at start-up by
// direct_selftest_rewrites and here at EVERY step: the step fails loudly (HC_ERR_DEVICE) instead of returning forces of an old state.
void check_canary(hc_ctx* c, unsigned long long seq) {
    const volatile unsigned long long* g = c->h_canary.p + (seq & 1) * 2;
    unsigned long long spins = 0;
    std::chrono::steady_clock::time_point t0{};
    while (g[1] != seq) {  // (the granule is stored by the workgroup that pushes the sample: it lands with the rows, give or take)
        __builtin_ia32_pause();
        if ((++spins & 0xFFFF) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (spins == 0x10000) t0 = now;
            if (std::chrono::duration<double>(now - t0).count() > step_timeout_seconds()) device_lost(c, "hc_step: the state canary of the step did not arrive");
        }
    }
    const unsigned long long bits = g[0];
    double got;
    std::memcpy(&got, &bits, sizeof got);
    if (got != static_cast<double>(seq))
        device_lost(c, "hc_step: the step kernel read a stale body state (canary " + std::to_string(got) + ", step " + std::to_string(seq) +
                           "): memory the host re-writes through the PCIe BAR was not re-read by the GPU");
}

constexpr double kArmGapSeconds = 25e-6;  // a caller that stays away longer than this finds the queue parked (see step_end)

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
    const double gap_hint = c->gap_hint;  // (for this call only, whatever becomes of it)
    c->gap_hint           = -1.0;
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
    // how long was the caller away?  Decides whether the queue is parked on a barrier after this step (step_end) and feeds the
    // adaptive pass schedule (hc_pass.cpp: schedule_ahead_for_next_block).  hc_step_multi hands every context of its group the gap
    // ITS caller saw (gap_hint), so that the shards of one array count the same gaps and decide alike.
    {
        const auto now = std::chrono::steady_clock::now();
        double gap     = -1.0;
        if (gap_hint >= 0.0) gap = gap_hint;
        else if (c->have_t_step_end) gap = std::chrono::duration<double>(now - c->t_step_end).count();
        if (gap >= 0.0) {
            c->gap_seen += 1;
            c->gap_long += gap > c->gap_threshold ? 1 : 0;
            c->gap_long_lo += gap > 0.6 * c->gap_threshold ? 1 : 0;
        }
        c->arm_after_step = c->arm_mode == 2 || (c->arm_mode == 1 && gap > kArmGapSeconds);
    }
    c->prev_time = t;  // :747 (set before the terms are computed, so a throwing step is not retried)
    c->have_prev = true;
    c->have_prev_device = false;  // d_total is about to be replaced (or left stale by a step that throws)
    std::fill(c->last_total.begin(), c->last_total.end(), 0.0);  // the reference zero-fills total_force_ before the terms (:749-751)
    // Boundary without copy launches or stream synchronisation: the host stores the state doubles straight into device memory through
    // the PCIe BAR -- behind the step kernel's arguments, or into the context's state buffer (enqueue_step decides: stage_host_state,
    // fill_slot_state) --, finalize_kernel stores the totals straight into mapped pinned memory, tagged with this step's sequence
    // number; one kernel launch for a step inside a block.
    // the canary: this step's sequence number as a double, stored behind the state through the same path (checked in step_end)
    const unsigned long long seq_next = c->seq + 1;
    HostState hs{pos, rpy, linvel, angvel,
                 static_cast<double>((c->fault_stale_state_at >= 0 && static_cast<long long>(seq_next) == c->fault_stale_state_at) ? seq_next - 1 : seq_next)};
    c->step_canary_in = nullptr;
    const unsigned long long seq = ++c->seq;
#ifdef HC_TUNING
    if (c->stamps_on && c->dq) {
        c->host_stamps[seq % hc::kStampSteps][0] = c->dq->system_ticks();  // (a few hundred ns after the call's entry: the cache rules are behind us)
        c->host_stamps[seq % hc::kStampSteps][3] = 0;
    }
#endif
    c->step_canary_out = canary_enabled() ? c->h_canary.dp + (seq & 1) * 2 : nullptr;
    // The tagged results of consecutive steps go to alternate halves of the result buffer: a reader in ANOTHER process (a caller's
    // buffer in shared memory, hc_set_result_buffer) may still be collecting step n while this process has moved on to step n + 1;
    // it cannot reach step n + 2 before every process has the rows of step n + 1, i.e. has finished with step n.
    enqueue_step(c, t, nullptr, nullptr, c->stream, StepFlags{}, result_tags_dev(c, seq), seq, defer_tail, &hs);
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
#ifdef HC_TUNING
        if (c->stamps_on && c->dq) c->host_stamps[c->seq % hc::kStampSteps][2] = c->dq->system_ticks();  // every row's granule has arrived
#endif
        if (canary_enabled()) check_canary(c, c->seq);
        if (c->device_errors_possible) {
            quiesce_direct(c);
            check_device_flag(c);
        }
        c->prev_time_device = c->pending_t;  // finalize_kernel has left the same totals in d_total: hc_step_device at this time copies them
        c->have_prev_device = true;
        // Everything this step had to enqueue is in the queue.  A caller that was away for a while before this step (a Chrono loop
        // integrating between force evaluations) will be away again: leave the packet processor parked on a barrier, so that the
        // next step's kernel starts at once instead of after the ~6 us an idle queue needs (DirectQueue::arm).
        // (Not when the process holds several contexts on this device -- row shards sharing a GPU: a parked queue keeps a hardware
        // queue slot busy and slows the other contexts' queues down, 19.8 -> 32-70 us for two contexts on one GPU.)
        if (c->path == 2 && c->direct_ready && c->arm_after_step && contexts_on_device(c->device) == 1) {
            c->dq->arm(0);
            c->prof.queue_parkings++;
        }
        c->t_step_end      = std::chrono::steady_clock::now();
        c->have_t_step_end = true;
    }
    if (force_out) std::memcpy(force_out, c->last_total.data(), c->Dloc * sizeof(double));
}

// A failed begin leaves nothing pending; whatever the step had enqueued before it threw is harmless (results nobody waits for).
void step_abort(hc_ctx* c) {
    c->pending_step = 0;
    c->tail.pending = false;
    c->tile_counter_suspect = c->d_tile_counter.n > 0;  // (see enqueue_step: the fused wide step's arrival counters)
    c->step_canary_in  = nullptr;  // (a begin that threw before its enqueue_step consumed them)
    c->step_canary_out = nullptr;
}

extern "C++" {
// ---- several contexts in one call (hc_step_multi, hc_added_mass_mv_multi) ----------------------------------------
// Worker threads of the fan-out (hc_fanout.hpp): HC_MULTI_THREADS = 0 keeps everything on the calling thread; HC_MULTI_SPIN_US is
// how long an idle worker spins before it sleeps.  While it spins it holds a host core at 100 % -- n_ctx - 1 cores for the whole run of
// a host that calls more often than the spin time, beside Chrono's own OpenMP threads -- and once it sleeps the next call pays a futex
// wake-up: 200 us by default (round 4: 1 ms), measured in profiles/r05/multi_spin.txt.
int multi_threads() {
    static const int n = env_int("HC_MULTI_THREADS", 63);
    return n;
}
hc::FanOut& fanout() {
    static hc::FanOut pool(multi_threads(), static_cast<double>(env_int("HC_MULTI_SPIN_US", 200)));
    return pool;
}
// A worker thread of the fan-out serves the same context call after call and nobody else changes its current device, so it sets the
// device when it changes only.  Every other thread sets it every time: the calling thread shares its current device with every other
// entry point, and it runs more than item 0 -- the items beyond HC_MULTI_THREADS, and all of them when another thread holds the pool
// or no worker thread could be created (hc_fanout.hpp) -- so the cache is keyed on the KIND OF THREAD, not on the item.
void bind_device(const hc_ctx* c) {
    static thread_local int worker_device = -1;
    if (!hc::FanOut::on_worker_thread()) {
        HC_HIP(hipSetDevice(c->device));
    } else if (worker_device != c->device) {
        HC_HIP(hipSetDevice(c->device));
        worker_device = c->device;
        // a worker serves one context: it moves next to that context's GPU (its doorbell, its BAR, the memory its results arrive in)
        static const bool pin = env_int("HC_MULTI_PIN", 1) != 0;
        if (pin) (void)bind_calling_thread_to_device(c->device);
    }
}
// status and message per item: items run by the fan-out record their failure in their own slot, the one-thread form records in slot 0
// (it runs the contexts in order, so the first failure it sees is the first in context order); the call reports the first slot that
// failed, and every context of the group gets the message
struct MultiStatus {
    std::vector<int> status;
    std::vector<std::string> message;
    explicit MultiStatus(int n) : status(static_cast<size_t>(n), HC_OK), message(static_cast<size_t>(n)) {}
    template <class F, class A>
    bool guarded(hc_ctx* c, int item, F&& fn, A&& on_failure) {
        auto fail = [&](int code, const char* what) {
            const size_t k = static_cast<size_t>(item);
            if (status[k] == HC_OK) { status[k] = code; message[k] = what; }
            on_failure();
            return false;
        };
        try {
            bind_device(c);
            fn();
            return true;
        } catch (const Error& e) {
            return fail(e.status, e.what());
        } catch (const std::out_of_range& e) {
            return fail(HC_ERR_OUT_OF_RANGE, e.what());
        } catch (const std::exception& e) {
            return fail(HC_ERR_RUNTIME, e.what());
        }
    }
    int finish(hc_ctx* const* ctxs, int n) const {
        for (int g = 0; g < n; ++g)
            if (status[static_cast<size_t>(g)] != HC_OK) {
                for (int k = 0; k < n; ++k) {
                    ctxs[k]->err = message[static_cast<size_t>(g)];  // hc_last_error of any context of the group tells why
                    // the shards that did step have counted a gap the failed one has not: the adaptive pass schedule of the group
                    // starts its count afresh on ALL of them, so that they go on deciding alike
                    reset_schedule_state(ctxs[k]);
                }
                return status[static_cast<size_t>(g)];
            }
        return HC_OK;
    }
};
}  // extern "C++"
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

int hc_step_many(hc_ctx* c, int n, const double* t_n, const double* states, double* forces, double* seconds_n, int* done) {
    if (done) *done = 0;
    if (!c) return HC_ERR_INVALID;
    if (n <= 0) return HC_OK;
    if (!t_n || !states || !forces) {
        c->err = "null pointer";
        return HC_ERR_INVALID;
    }
    const size_t n3 = static_cast<size_t>(3) * c->N, dl = static_cast<size_t>(c->Dloc);
    for (int k = 0; k < n; ++k) {
        const double* st = states + static_cast<size_t>(k) * 4 * n3;
        const auto a     = std::chrono::steady_clock::now();
        const int rc     = hc_step(c, t_n[k], st, st + n3, st + 2 * n3, st + 3 * n3, forces + static_cast<size_t>(k) * dl);
        if (seconds_n) seconds_n[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        if (rc != HC_OK) return rc;
        if (done) *done = k + 1;
    }
    return HC_OK;
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
    MultiStatus st(n_ctx);
    const auto t_entry = std::chrono::steady_clock::now();
    // the gap the caller left since its last call, measured ONCE for the group (kept on its first context): every shard counts the
    // same gaps, so the adaptive pass schedule decides alike on all of them and their rows stay those of one schedule
    // (a call that only re-reads the per-time cache -- Chrono's other 6N - 1 callbacks of a time -- is no step: it neither counts nor
    // moves the mark)
    const bool evaluates = !(ctxs[0]->have_prev && t == ctxs[0]->prev_time);
    if (evaluates && ctxs[0]->have_t_multi_end) {
        const double gap = std::chrono::duration<double>(t_entry - ctxs[0]->t_multi_end).count();
        for (int g = 0; g < n_ctx; ++g) ctxs[g]->gap_hint = gap;
    }
    struct MarkEnd {
        hc_ctx* c;
        bool on;
        ~MarkEnd() {
            if (!on) return;
            c->t_multi_end      = std::chrono::steady_clock::now();
            c->have_t_multi_end = true;
        }
    } mark_end{ctxs[0], evaluates};
    auto doorbell_rung = [t_entry](hc_ctx* c) {  // (the step kernel's packet is in its queue: hc_profile_stats::multi_doorbell_offset_*)
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count();
        c->prof.multi_doorbell_offset_last = s;
        c->prof.multi_doorbell_offset_sum += s;
        c->prof.multi_calls += 1;
    };
    if (n_ctx > 1 && multi_threads() > 0) {
        // one thread per context (hc_fanout.hpp): every GPU gets its state and its step kernel at once, each thread then enqueues what
        // later steps need on ITS context and collects ITS rows -- the last doorbell rings one context's cost after the call
        auto item = [&](int g) {
            hc_ctx* c = ctxs[g];
            st.guarded(c, g, [&] {
                require(c->N == ctxs[0]->N, HC_ERR_INVALID, "hc_step_multi: the contexts belong to different systems");
                step_begin(c, t, pos, rpy, linvel, angvel, true);
                doorbell_rung(c);
                enqueue_tail(c);
                step_end(c, force_out + static_cast<size_t>(6) * c->b0);
            }, [&] { step_abort(c); });
        };
        fanout().run(n_ctx, item);
        return st.finish(ctxs, n_ctx);
    }
    // one thread (HC_MULTI_THREADS=0): three phases, so that every GPU is working before the host does anything else
    std::vector<char> begun(static_cast<size_t>(n_ctx), 0);
    for (int g = 0; g < n_ctx; ++g) {
        hc_ctx* c = ctxs[g];
        begun[g]  = st.guarded(c, 0, [&] {
            require(c->N == ctxs[0]->N, HC_ERR_INVALID, "hc_step_multi: the contexts belong to different systems");
            step_begin(c, t, pos, rpy, linvel, angvel, true);
            doorbell_rung(c);
        }, [&] { step_abort(c); });
    }
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) begun[g] = st.guarded(ctxs[g], 0, [&] { enqueue_tail(ctxs[g]); }, [&] { step_abort(ctxs[g]); });
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) st.guarded(ctxs[g], 0, [&] { step_end(ctxs[g], force_out + static_cast<size_t>(6) * ctxs[g]->b0); }, [&] { step_abort(ctxs[g]); });
    return st.finish(ctxs, n_ctx);
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
    c->lookahead = steps <= 0 ? 0 : (steps <= 16 ? 16 : (steps <= 32 ? 32 : 64));
#ifndef HC_TUNING
    if (c->lookahead == 64) c->lookahead = 32;  // (depth 64 exists in the tuning build only: measured in round 5 and not taken, EXPERIMENTS.md)
#endif
    // depth 64 (tuning build, profiles/r05): the single-level form of D % 8 == 0 systems with the pass at block start, direct or not
    if (c->lookahead == 64 && ((c->D & 7) != 0 || c->D < 32 || hc::near_slices_for(c->D) > 1 || c->ntiles % c->mt_block64 != 0 ||
                               (c->direct_ready && !c->dk_block64.ok()))) c->lookahead = 32;
    choose_conv_config(c);  // the pass chunking depends on the depth
    alloc_partials(c);
    c->plan      = hc::Plan{};
    c->ahead.active = false;
    HC_API_END(c)
}

#ifdef HC_TUNING
// Tuning build only (profiles/step_stamps_probe.cpp): the stage clock of the step kernel.  hc_tuning_enable_step_stamps switches it on;
// hc_tuning_step_stamps hands out, for step `seq` (one of the last kStampSteps), the host's stamps {begin of the step, doorbell of the
// step kernel, totals seen} and the [workgroups][kStampStages] stage times of the step kernel's workgroups, all in microseconds after
// the host's begin stamp (GPU clock converted to the HSA system clock by the runtime, hsa_amd_profiling_convert_tick_to_system_domain).
// Returns HC_ERR_INVALID for a step that was not one direct dispatch of finalize_kernel<4, true> (no stamps were taken).
int hc_tuning_enable_step_stamps(hc_ctx* c, int on) {
    HC_API_BEGIN(c)
    c->stamps_on = on != 0;
    HC_API_END(c)
}
int hc_tuning_step_stamps(hc_ctx* c, unsigned long long seq, double* host_us3, double* wg_us, int* n_wg, int* n_stage) {
    HC_API_BEGIN(c)
    require(host_us3 && wg_us && n_wg && n_stage && c->dq, HC_ERR_INVALID, "bad arguments");
    require(seq <= c->seq && seq + hc::kStampSteps > c->seq && c->d_stamps.n > 0, HC_ERR_INVALID, "no stamps of that step are kept");
    const unsigned long long* hsx = c->host_stamps[seq % hc::kStampSteps];
    const int grid = static_cast<int>(hsx[3]);
    require(grid > 0, HC_ERR_INVALID, "that step did not go out as one direct dispatch of the slot-state step kernel");
    HC_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> raw(static_cast<size_t>(hc::kStampWGs) * hc::kStampStages);
    HC_HIP(hipMemcpy(raw.data(), c->d_stamps.p + (seq % hc::kStampSteps) * raw.size(), raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double us_per_tick = 1e6 / static_cast<double>(std::max<uint64_t>(1, c->dq->system_ticks_per_second()));
    const double t0          = static_cast<double>(hsx[0]);
    for (int k = 0; k < 3; ++k) host_us3[k] = (static_cast<double>(hsx[k]) - t0) * us_per_tick;
    for (int w = 0; w < grid; ++w)
        for (int k = 0; k < hc::kStampStages; ++k) {
            const unsigned long long g = raw[static_cast<size_t>(w) * hc::kStampStages + k];
            wg_us[w * hc::kStampStages + k] = g ? (static_cast<double>(c->dq->gpu_to_system(g)) - t0) * us_per_tick : -1.0;
        }
    *n_wg    = grid;
    *n_stage = hc::kStampStages;
    HC_API_END(c)
}

// Tuning build only (profiles/pass_depth_probe.py): the pass kernel of depth 16 / 32 / 64 over this context's K, `reps` launches timed
// with HIP events, for the predicted steps that would follow the newest history sample -- also for contexts whose step path has no
// plan at that depth (a wide system at depth 64), so that the depth-64 pass can be measured at C4-rank size.  The context's
// look-ahead state is dropped; call hc_set_lookahead afterwards.
int hc_tuning_time_pass(hc_ctx* c, int depth, int reps, double* mean_us, double* bytes_once) {
    HC_API_BEGIN(c)
    require(c->finalized && mean_us, HC_ERR_INVALID, "bad arguments");
    require(depth == 16 || depth == 32 || depth == 64, HC_ERR_INVALID, "depth must be 16, 32 or 64");
    require(c->times.size() >= 2 && (c->D & 7) == 0, HC_ERR_INVALID, "needs a history of two samples and D % 8 == 0");
    require(depth < 64 || c->ntiles % c->mt_block64 == 0, HC_ERR_INVALID, "row tiles not divisible by HC_BLOCK64_MT");
    HC_HIP(hipDeviceSynchronize());
    c->lookahead = depth;
    choose_conv_config(c);
    alloc_partials(c);
    c->ahead.active = false;
    auto& pl   = c->plan;
    pl         = hc::Plan{};
    const double t0 = c->times[0], dt = c->times[0] - c->times[1];
    pl.dt      = dt;
    for (int j = 0; j <= 2 * depth + 1; ++j) pl.tgrid[j] = (j == 0) ? t0 : t0 + j * dt;
    for (int m = 1; m <= depth; ++m) {  // (the pass's share of block step m, as build_plan defines it)
        int sc = 0;
        while (sc < c->S && !(pl.tgrid[m] - c->tau[static_cast<size_t>(sc)] <= pl.tgrid[1])) ++sc;
        pl.s_cut[m - 1]   = sc;
        pl.s_defer[m - 1] = -1;
    }
    const PassSetup ps = make_pass(c, false, false);
    hipEvent_t a, b;
    HC_HIP(hipEventCreate(&a));
    HC_HIP(hipEventCreate(&b));
    const int mt = depth == 64 ? c->mt_block64 : c->mt_block;
    hc::launch_conv_block(ps.b, mt, c->stream);  // warm
    HC_HIP(hipStreamSynchronize(c->stream));
    // every launch timed on its own, with `pause` of idle GPU in front of it (HC_TUNING_PASS_PAUSE_US, default 0 = back to back): passes
    // that follow each other without a break run under sustained matrix-pipe + HBM load, which the chip answers with a lower clock;
    // in the product a pass comes once per block
    const int pause_us = env_int("HC_TUNING_PASS_PAUSE_US", 0);
    double total_ms = 0.0;
    for (int r = 0; r < std::max(1, reps); ++r) {
        if (pause_us > 0) {
            const auto t0_ = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0_).count() < pause_us) {
            }
        }
        HC_HIP(hipEventRecord(a, c->stream));
        hc::launch_conv_block(ps.b, mt, c->stream);
        HC_HIP(hipEventRecord(b, c->stream));
        HC_HIP(hipEventSynchronize(b));
        float ms = 0.0f;
        HC_HIP(hipEventElapsedTime(&ms, a, b));
        total_ms += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *mean_us = 1e3 * total_ms / std::max(1, reps);
    if (bytes_once) *bytes_once = ps.rad_once;
    c->plan = hc::Plan{};
    HC_API_END(c)
}
#endif

int hc_set_pass_schedule(hc_ctx* c, int one_block_ahead, int slices) {
    HC_API_BEGIN(c)
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    HC_HIP(hipDeviceSynchronize());  // a pass in the making may still be running
    c->pass_ahead   = one_block_ahead < 0 ? default_pass_ahead(c) : (one_block_ahead ? 1 : 0);  // (< 0: adaptive unless HC_PASS_AHEAD pins it)
    reset_schedule_state(c);
    c->pass_slices  = slices > 0 ? std::min(slices, hc::kDepthDefault - 1) : default_pass_slices(c);
    c->ahead.active = false;
    alloc_partials(c);
    c->plan = hc::Plan{};
    if (pass_ahead_possible(c) && c->lookahead > 0) (void)pass_lane_ready(c);  // (created and self-tested here, off the step path)
    HC_API_END(c)
}

int hc_get_schedule(const hc_ctx* c, int* lookahead, int* pass_schedule, int* ahead_now, int* slices) {
    if (!c) return HC_ERR_INVALID;
    if (lookahead) *lookahead = c->lookahead;
    if (pass_schedule) *pass_schedule = c->pass_ahead == 2 ? -1 : c->pass_ahead;
    if (ahead_now) *ahead_now = c->pass_ahead == 2 ? (c->ahead_now ? 1 : 0) : c->pass_ahead;
    if (slices) *slices = c->pass_slices;
    return HC_OK;
}

int hc_direct_dispatch_active(const hc_ctx* c) { return (c && c->direct_ready) ? 1 : 0; }
const char* hc_dispatch_mode_reason(const hc_ctx* c) {
    if (!c) return "no context";
    if (!c->finalized) return "hc_finalize has not been called";
    return c->direct_ready ? c->direct_how.c_str() : c->direct_why.c_str();
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
    c->ahead.active = false;
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
    c->ahead.active = false;
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
    const unsigned long long seq = ++c->seq_am;
    hr[c->Dloc] = static_cast<double>(seq);  // the canary: handed back by the kernel, compared in added_mass_end (see check_canary)
    if (bar) _mm_sfence();
    const double* dw = bar ? c->bar_am.p : c->h_am.dp;
    const double* canary_in         = canary_enabled() ? dw + c->D + c->Dloc : nullptr;
    unsigned long long* canary_out  = canary_enabled() ? c->h_tag_am.dp + 2 * static_cast<size_t>(c->Dloc) : nullptr;
    if (c->direct_ready && bar && c->am_lane == 0) {
        // first product of this context: the second lane (a queue of its own) is created and self-tested now
        std::string why;
        bool abandon = false;
        c->am_lane   = (c->dq->ensure_lane(1, &why) && direct_selftest_rewrites(c, c->dq, 1, &abandon)) ? 1 : -1;
        c->direct_why.clear();  // (the step path's lane stays in use whatever the second lane's test said)
    }
    if (c->arm_mode == 1) {
        const auto now  = std::chrono::steady_clock::now();
        c->arm_after_am = c->have_t_am_end && std::chrono::duration<double>(now - c->t_am_end).count() > 25e-6;
    } else {
        c->arm_after_am = c->arm_mode == 2;
    }
    if (c->direct_ready && bar && c->am_lane == 1) {
        // the second lane of the direct queue: an AQL packet instead of a HIP launch, independent of the step path's lane
        hc::AddedMassArgs a{c->d_ainf.p, c->Dloc, c->D, dw, dw + c->D, cc, c->h_tag_am.dp, seq, canary_in, canary_out};
        c->dq->dispatch(c->dk_added_mass, static_cast<uint32_t>((c->Dloc + 3) / 4), 256, 0, &a, sizeof a, -1, 0.0, 1);
        c->prof.direct_dispatches += 1;
        c->pending_am = 1;
    } else {
        hc::launch_added_mass_mv_tagged(c->d_ainf.p, c->Dloc, c->D, dw, dw + c->D, cc, c->h_tag_am.dp, seq, canary_in, canary_out, c->stream_am);
        c->prof.hip_launches += 1;
        HC_HIP(hipGetLastError());
        c->pending_am = 2;
    }
}
void check_am_canary(hc_ctx* c) {
    if (!canary_enabled()) return;
    const volatile unsigned long long* g = c->h_tag_am.p + 2 * static_cast<size_t>(c->Dloc);
    unsigned long long spins = 0;
    std::chrono::steady_clock::time_point t0{};
    while (g[1] != c->seq_am) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFF) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (spins == 0x10000) t0 = now;
            if (std::chrono::duration<double>(now - t0).count() > step_timeout_seconds()) device_lost(c, "hc_added_mass_mv: the canary of the product did not arrive");
        }
    }
    const unsigned long long bits = g[0];
    double got;
    std::memcpy(&got, &bits, sizeof got);
    if (got != static_cast<double>(c->seq_am))
        device_lost(c, "hc_added_mass_mv: the kernel read stale inputs (canary " + std::to_string(got) + ", product " + std::to_string(c->seq_am) +
                           "): memory the host re-writes through the PCIe BAR was not re-read by the GPU");
}
void added_mass_end(hc_ctx* c, double* R) {
    const int how = c->pending_am;
    c->pending_am = 0;
    if (how == 1) {
        wait_tagged(c, c->h_tag_am.p, c->seq_am, nullptr, R + 6 * c->b0, 1);
        check_am_canary(c);
        if (c->direct_ready && c->arm_after_am && contexts_on_device(c->device) == 1) {  // the same parking for the added-mass lane
            c->dq->arm(1);
            c->prof.queue_parkings++;
        }
    } else if (how == 2) {
        wait_tagged(c, c->h_tag_am.p, c->seq_am, c->stream_am, R + 6 * c->b0);
        check_am_canary(c);
    }
    c->t_am_end      = std::chrono::steady_clock::now();
    c->have_t_am_end = true;
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
    MultiStatus st(n_ctx);
    if (n_ctx > 1 && multi_threads() > 0) {
        auto item = [&](int g) {
            hc_ctx* c = ctxs[g];
            st.guarded(c, g, [&] {
                added_mass_begin(c, w, cc, R, n_sys);
                added_mass_end(c, R);
            }, [&] { c->pending_am = 0; });
        };
        fanout().run(n_ctx, item);
        return st.finish(ctxs, n_ctx);
    }
    std::vector<char> begun(static_cast<size_t>(n_ctx), 0);
    for (int g = 0; g < n_ctx; ++g) begun[g] = st.guarded(ctxs[g], 0, [&] { added_mass_begin(ctxs[g], w, cc, R, n_sys); }, [&] { ctxs[g]->pending_am = 0; });
    for (int g = 0; g < n_ctx; ++g)
        if (begun[g]) st.guarded(ctxs[g], 0, [&] { added_mass_end(ctxs[g], R); }, [&] { ctxs[g]->pending_am = 0; });
    return st.finish(ctxs, n_ctx);
}

}  // extern "C"
