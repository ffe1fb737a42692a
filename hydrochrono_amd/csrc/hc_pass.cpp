// hc_pass.cpp -- the look-ahead passes behind the step path (hc_step.cpp): the pass of a block, the short passes of the two-level
// form, and the pass schedule "one block ahead" -- the pass of the NEXT block in slices, its short passes towards the next block,
// the pass lane they run on beside the steps, and the adoption of the rows when the next block starts.  Host code only: what runs on
// the GPU is conv_block_kernel / reduce_block_kernel of hc_kernels.hip, what is planned is hc_plan.hpp.
#include "hc_internal.hpp"

using namespace hc::detail;

namespace hc {
namespace detail {

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

// The look-ahead pass of the plan just made: for the 16 / 32 predicted steps, what the samples known now contribute.  It runs as
// the plain pass of a (virtual) step at tgrid[1] whose own sample is zero -- that sample's share is added later by the
// step itself and by its scatter.  Enqueued behind the step that has just been evaluated (its ring push included).
// next_block: the same pass for the predicted steps of the block AFTER this one (hc_plan.hpp: FarPass) -- the view of the history
// is the same, only the query times move on by a block.

PassSetup make_pass(hc_ctx* c, bool with_exc, bool next_block) {
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
    PassSetup ps;
    hc::BlockArgs& b = ps.b;
    b                     = hc::BlockArgs{};
    b.K                   = rad_panel(c);
    b.F                   = std::min(c->S, live_samples(c, pl.tgrid[L])) * c->D;
    b.depth               = L;
    b.chunk_gp            = next_block ? far_chunk_gp(c) : c->chunk_gp_block;  // (a pass issued in slices: shorter chunks, a full round of workgroups per slice)
    b.nchunks             = std::max(1, ((b.F + 7) / 8 + b.chunk_gp - 1) / b.chunk_gp);
    b.max_steps_per_chunk = (b.chunk_gp * 8) / c->D + 2;
    b.hist                = hv;
    if (next_block) {
        const hc::FarPass fp = hc::far_pass_setup(pl, L, c->tau);
        for (int j = 0; j < L; ++j) {
            b.tpred[j]   = fp.tpred[j];
            b.s_cut[j]   = fp.s_cut[j];
            b.s_defer[j] = -1;
        }
    } else {
        for (int j = 0; j < L; ++j) {
            b.tpred[j]   = pl.tgrid[j + 1];
            b.s_cut[j]   = pl.s_cut[j];
            b.s_defer[j] = pl.s_defer[j];
        }
    }
    b.tau   = c->d_tau.p;
    b.width = c->d_width.p;
    // The excitation force depends on time only, so the pass also evaluates it for the 16 predicted times (extra chunks
    // over Kex in the same launch) -- provided every predicted time passes the window tests a real step would have to pass.
    static const bool exc_in_block = HC_TUNE_INT("HC_EXC_IN_BLOCK", 1) != 0;
    bool exc_block = exc_in_block && with_exc && c->wave_kind == hc::kWaveIrregular && c->nchunks_ex_block > 0;
    for (int j = 0; j < L && exc_block; ++j) exc_block = wave_window_ok(c, b.tpred[j]);
    ps.exc_block  = exc_block;
    b.Kex         = vw.kex;
    b.ex          = vw.ex;
    b.chunk_gp_ex = c->chunk_gp_ex_block;
    b.nchunks_ex  = exc_block ? c->nchunks_ex_block : 0;
    b.partials    = next_block ? c->d_partials_far.p : c->d_partials_block.p;
    b.Dpad        = c->Dpad;
    b.error_flag  = c->d_err.p;
    b.item_counter = c->d_err.p + 1;
    b.ngroups     = c->ntiles / (L == 64 ? c->mt_block64 : c->mt_block);
    // algorithmic bytes: summed over the steps of the block, step j's share of K and of the velocity vector from s_cut[j]
    // on ...; what the launch has to move once: the live part of K, Kex and the staged vectors
    double samples = 0.0;
    for (int j = 0; j < L; ++j) samples += std::max(0, b.F / c->D - b.s_cut[j]);
    const double rad_16 = 8.0 * samples * (static_cast<double>(c->Dloc) * c->D + c->D);
    const double exc_16 = exc_block ? 8.0 * L * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
    ps.bytes_steps = rad_16 + exc_16;
    ps.rad_once    = 8.0 * (static_cast<double>(c->Dloc) * b.F + b.F);
    ps.exc_once    = exc_block ? 8.0 * (static_cast<double>(c->Dloc) * c->L + c->L) : 0.0;
    if (HC_TUNE_INT("HC_DEBUG_PLAN", 0) != 0) {
        std::fprintf(stderr, "[hc] pass%s t0=%.6f dt=%.17g Hv=%d F/D=%d nchunks=%d exc=%d\n     s_cut:", next_block ? " (next block)" : "", pl.tgrid[0], pl.dt, Hv,
                     b.F / c->D, b.nchunks, (int)exc_block);
        for (int j = 0; j < L; ++j) std::fprintf(stderr, " %d", b.s_cut[j]);
        std::fprintf(stderr, "\n     s_defer:");
        for (int j = 0; j < L; ++j) std::fprintf(stderr, " %d", b.s_defer[j]);
        std::fprintf(stderr, "\n     scat:");
        for (int i = 1; i <= L; ++i) std::fprintf(stderr, " [%d,%d]", pl.scat_lo[i], pl.scat_hi[i]);
        std::fprintf(stderr, "\n");
    }
    return ps;
}

// One launch of the pass over the radiation chunks [first, last) (+ the excitation work items if with_items).
void issue_pass_chunks(hc_ctx* c, const PassSetup& ps, int first, int last, bool with_items, hipStream_t stream, bool direct, int lane) {
    hc::BlockArgs b = ps.b;
    b.chunk_first   = first;
    b.chunk_last    = last;
    if (!with_items) b.nchunks_ex = 0;
    const double share_rad = static_cast<double>(last - first) / std::max(1, ps.b.nchunks);
    const double rad_once = ps.rad_once * share_rad, exc_once = with_items ? ps.exc_once : 0.0;
    c->prof.block_kernel_bytes      = ps.bytes_steps * (rad_once + exc_once) / std::max(1.0, ps.rad_once + ps.exc_once);
    c->prof.block_kernel_bytes_once = rad_once + exc_once;
    const double exc_share = exc_once / std::max(1.0, rad_once + exc_once);
    const int L  = c->lookahead;
    const int mt = L == 64 ? c->mt_block64 : c->mt_block;
    if (direct) {
        hc::BlockArgs b2;
        const hc::BlockLaunch l = hc::block_launch_config(b, mt, &b2);
        if (l.nblocks <= 0) return;
        c->dq->dispatch(L == 64 ? c->dk_block64 : (L == 32 ? c->dk_block32 : c->dk_block16), static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &b2, sizeof b2,
                        direct_tag(c, hc::kEvPass), exc_share, lane);
        c->prof.direct_dispatches += 1;
        if (lane == 2) c->prof.pass_lane_launches += 1;
        return;
    }
    hc::EventPair* ev = ev_begin(c, hc::kEvPass, stream, exc_share);
    hc::launch_conv_block(b, mt, stream);
    ev_end(ev, stream);
    c->prof.hip_launches += 1;
}

// ... and the reduction of its chunk partials into the rows P / E of a block.
void issue_pass_reduce(hc_ctx* c, const PassSetup& ps, double* P, double* E, hipStream_t stream, bool direct, int lane) {
    const hc::BlockArgs& b = ps.b;
    const hc::ReduceArgs r{b.partials, b.nchunks, b.nchunks_ex, c->Dpad, c->lookahead, P, E, b.item_counter, 0, 0, 0, 0};
    if (direct) {
        c->dq->dispatch(c->dk_reduce, static_cast<uint32_t>(hc::reduce_block_grid(r)), 256, 0, &r, sizeof r, -1, 0.0, lane);
        c->prof.direct_dispatches += 1;
        return;
    }
    hc::launch_reduce_block(r, stream);
    c->prof.hip_launches += 1;
}

void launch_pass(hc_ctx* c, hipStream_t stream, bool with_exc, bool direct) {
    const PassSetup ps = make_pass(c, with_exc, false);
    c->plan.has_exc    = ps.exc_block;
    issue_pass_chunks(c, ps, 0, ps.b.nchunks, true, stream, direct);
    issue_pass_reduce(c, ps, rows_P(c, false), rows_E(c, false), stream, direct);
}

// ---- pass schedule "one block ahead" (hc_set_pass_schedule) ----------------------------------------
// Begun right after the pass-free start of a block (or after the ordinary pass of the first block): the pass of the NEXT block, in
// `pass_slices` launches -- one now, the others behind the scatter launches of the following steps -- so that a caller that
// leaves the GPU idle between two force evaluations never waits for a whole pass.  A slice is a full round of workgroups over chunks
// `pass_slices` times shorter than those of the ordinary pass (far_chunk_gp); the chunk partials of all slices are added by ONE
// reduction after the last slice, in chunk order -- the sums do not depend on how the chunks were spread over launches.
void ahead_drop(hc_ctx* c) { c->ahead.active = false; }

// The adaptive schedule's state as a fresh context has it: no gaps seen, and as the first answer what a caller without gaps gets --
// "ahead" exactly where the threshold is zero (wide systems whose slice of K is small enough for latency-bound step kernels, hc_setup.cpp).
void reset_schedule_state(hc_ctx* c) {
    c->gap_seen = c->gap_long = c->gap_long_lo = 0;
    c->gap_hint  = -1.0;
    c->ahead_now = hc::near_slices_for(c->D) > 1 && c->gap_threshold <= 0.0;
}

// Does the pass of the block AFTER the one that starts now run one block ahead?  Schedule 0 / 1: as selected.  Adaptive: by majority
// over the gaps the caller left between its synchronous steps since the last decision (step_begin counts them: the time from the
// end of one hc_step / hc_step_multi to the begin of the next, whatever the caller did in between -- integrate, call the added-mass
// product, nothing).  More than half of them longer than gap_threshold: the caller is away between steps, the pass goes beside them;
// else it steps back to back and the pass runs at block start, unsliced.  Decided per block, from at least a quarter block of gaps;
// callers that never wait for a step (hc_step_device) leave no gaps and keep the first answer.  A majority, not a mean: one long
// pause (the host printing a line, the OS taking the core) does not flip a tight loop, and a test that changes its regime at a fixed
// step gets the same decisions run after run.  With hysteresis: it takes a majority above the threshold to go ahead and a majority
// below 0.6 of it to come back, so a caller whose gaps sit at the threshold (where the two schedules cost the same,
// profiles/r05/ahead_probe_fine_gaps.txt) does not pay for a change of schedule every other block (the block after a change to
// "ahead" runs two passes: its own and the next one's).
bool schedule_ahead_for_next_block(hc_ctx* c) {
    if (c->pass_ahead == 0 || c->lookahead > 32) return false;  // (the experimental depth 64: pass at block start only)
    if (c->pass_ahead == 1) return true;
    if (!pass_ahead_size_ok(c)) return false;
    if (c->gap_seen >= std::max(4, c->lookahead / 4)) {
        c->ahead_now = 2 * (c->ahead_now ? c->gap_long_lo : c->gap_long) > c->gap_seen;
        c->gap_seen = c->gap_long = c->gap_long_lo = 0;
    }
    (c->ahead_now ? c->prof.schedule_blocks_ahead : c->prof.schedule_blocks_at_start) += 1;
    return c->ahead_now;
}

void ahead_issue_slice(hc_ctx* c, hipStream_t stream, bool direct) {
    auto& ah = c->ahead;
    if (!ah.active || ah.reduced) return;
    if (ah.Hcap != c->Hcap || ah.plan_serial != c->plan_serial) { ahead_drop(c); return; }  // the ring was re-allocated under the view
    // (defensive; ahead_begin has made the room: the samples pushed since the view was taken, and those the rest of the block will
    // push while the slices run, stay clear of the oldest slot the view reads)
    const int pushed = ((c->head - ah.head0) % c->Hcap + c->Hcap) % c->Hcap;
    if (std::max(pushed, c->lookahead) + ah.Hv - 2 >= c->Hcap) { ahead_drop(c); return; }
    PassSetup ps;
    ps.b         = ah.args;
    // (buffers that may have been re-allocated since the view was taken are re-read; the ring's geometry has been checked above)
    ps.b.hist.ring_t  = c->d_ring_t.p;
    ps.b.hist.ring_v  = c->d_ring_v.p;
    ps.b.hist.ring_vT = c->d_ring_vT.p;
    ps.exc_block = ah.has_exc;
    ps.rad_once  = ah.rad_once;
    ps.exc_once  = ah.exc_once;
    ps.bytes_steps = ah.rad_once + ah.exc_once;
    // beside the steps (the pass lane) when the chain began there and this step is dispatched directly; a step that goes through HIP
    // launches has emptied both lanes on its way in (enqueue_step), so its slice may follow on the stream
    const int lane = (ah.concurrent && direct) ? 2 : 0;
    const int n = ps.b.nchunks, k = ah.issued;
    const int first = std::min(n, k * ah.per_slice), last = std::min(n, (k + 1) * ah.per_slice);  // never empty, see ahead_begin
    const bool final_slice = k + 1 >= ah.slices;
    issue_pass_chunks(c, ps, first, last, final_slice, stream, direct, lane);
    ah.issued = k + 1;
    c->prof.ahead_pass_slices += 1;
    if (final_slice) {
        issue_pass_reduce(c, ps, rows_P(c, true), rows_E(c, true), stream, direct, lane);
        ah.reduced = true;
    }
}

// The pass lane: created when the schedule first needs it, with a CU mask that leaves pass_free_cus compute units of every XCD to the
// kernels of the step path (a pass workgroup holds its CU's registers for its whole life, so a step kernel that finds no free CU
// would wait for a pass workgroup to end; profiles/overlap_probe.hip).  Without the mask the lane is not used.
bool pass_lane_ready(hc_ctx* c) {
    // one more queue per context: only where the device is this context's alone.  Several contexts on ONE device -- of this process
    // (contexts_on_device) or of other processes, which the launcher says with HC_DEVICE_SHARED=1: a test bed for the multi-GPU
    // code, not a deployment -- already time-share its hardware queues; their slices stay on the step path's lane.
    if (contexts_on_device(c->device) > 1) return false;
    if (c->pass_lane == 2) return true;
    if (c->pass_lane < 0 || !c->pass_concurrent || !c->dq || !c->direct_ready) return false;
    std::string why;
    const uint32_t ncu = c->dq->compute_units();
    const uint32_t keep = ncu > 8u * static_cast<uint32_t>(c->pass_free_cus) + 8 ? ncu - 8u * static_cast<uint32_t>(c->pass_free_cus) : 0;
    bool abandon = false;
    // (timed like lane 0; the same self-test as the other lanes: argument slots the host re-writes must be re-read, not served stale)
    bool ok = keep > 0 && c->dq->ensure_lane(2, &why) && c->dq->set_cu_mask(2, keep);
    if (ok) c->dq->enable_timing(2);
    const std::string why_before = c->direct_why;
    ok = ok && direct_selftest_rewrites(c, c->dq, 2, &abandon);
    c->direct_why = why_before;  // (the step path's lane stays in use whatever this lane's test said)
    if (abandon) c->dq->abandon_lane(2);  // a dispatch that never completed: the lane's queue is left alone, nothing ever waits for it again
    c->pass_lane = ok ? 2 : -1;
    return c->pass_lane == 2;
}

void ahead_begin(hc_ctx* c, hipStream_t stream, bool with_exc, bool direct) {
    auto& ah = c->ahead;
    ah.active = false;
    const int L = c->lookahead;
    if (L <= 0 || !schedule_ahead_for_next_block(c) || !hc::far_pass_allowed(c->plan, L, c->times, c->tau)) return;
    // the short passes towards the next block must fit the partials buffer (they stream up to twice the reach of the in-block ones;
    // the first window reaches furthest)
    int first_end = 1;
    while (!hc::next_window_end(c->plan, L, first_end)) ++first_end;
    const hc::MiniPass probe = hc::mini_pass_next(c->plan, L, first_end, hc::next_window_length(c->plan, L, first_end), c->tau);
    const int chunk_gp = std::max(16, (((c->D + 7) / 8 / 2 + 15) / 16) * 16);
    const long long chunks = (static_cast<long long>(probe.n_samples) * c->D / 8 + chunk_gp) / chunk_gp + 1;
    const size_t need = static_cast<size_t>(chunks) * L * c->Dpad;
    if (need > c->d_partials_block.n || need > c->d_partials_next.n || c->d_partials_far.n == 0) return;
    // The view of the history this pass takes now (samples k = 0 .. Hv - 2 behind `head`) is read until its reduction, i.e. while
    // up to L more steps push their samples into the slots after `head`: none of those slots may be one of the view's.  The ring
    // is allocated with 64 slots beyond the IRF window, but a step below the IRF spacing fills them with kept samples (H up to
    // Hcap before history_push grows the ring), so the room is made here when it is missing (rare: a re-allocation, like any grow).
    {
        const int H = static_cast<int>(c->times.size());
        const int room_for = (H + 1) + L + 2;  // view (at most H + 1 samples incl. the virtual one) + the block's pushes + slack
        if (c->Hcap < room_for) {
            ring_grow(c, room_for, std::min(c->Hcap, H + static_cast<int>(c->retired.size())));
            c->prof.ring_grows_for_pass += 1;
        }
    }
    const PassSetup ps = make_pass(c, with_exc, true);
    ah.args        = ps.b;
    ah.has_exc     = ps.exc_block;
    ah.rad_once    = ps.rad_once;
    ah.exc_once    = ps.exc_once;
    ah.Hcap        = c->Hcap;
    ah.head0       = c->head;
    ah.Hv          = ps.b.hist.H;
    ah.plan_serial = c->plan_serial;
    ah.t_first     = c->plan.tgrid[L + 1];
    ah.t_last      = c->plan.tgrid[2 * L];
    // a slice = one round of workgroups of the pass lane (far_chunks_per_slice: whole octets of chunks, as the kernel's block mapping
    // deals them); the reduction must be out before the first short pass adds to the rows, i.e. within the first sub-block (wider
    // slices if that takes fewer of them)
    const int limit = first_end;  // (the first window of block samples ends there: the sub-block size, L - 1 for the single-level form)
    int per         = far_chunks_per_slice(c);
    while ((ps.b.nchunks + per - 1) / per > limit) per += 8;
    ah.per_slice    = per;
    ah.slices       = std::max(1, (ps.b.nchunks + per - 1) / per);
    ah.issued      = 0;
    ah.reduced     = false;
    ah.concurrent  = false;
    ah.active      = true;
    if (direct && pass_lane_ready(c)) {
        // beside the steps: the slices (and later the short passes towards the next block) go to the pass lane, behind everything
        // lane 0 holds now -- the step kernel that pushed the newest sample the pass reads, the last reader of the rows it is going to
        // overwrite.  Still in slices: a caller that stays away between steps then finds each slice done when it comes back (the
        // pass would otherwise share the memory system with the next few steps), one that steps back to back just fills the lane.
        const uint64_t h = c->dq->signal_after(0);
        if (h != 0) {
            c->dq->wait_for(2, h);
            ah.args.item_counter = c->d_err.p + 2;
            ah.concurrent        = true;
        }
    }
    ahead_issue_slice(c, stream, direct);
}

// Before the next block is planned: will there be rows for it?  The pass in the making is complete, belongs to the plan that has just
// ended, and the step times the new plan is going to predict (the same expressions as build_plan) are those it was computed for.
bool ahead_expected(const hc_ctx* c, unsigned long long ended_serial) {
    const auto& ah = c->ahead;
    const int L = c->lookahead;
    if (!ah.active || !ah.reduced || ah.plan_serial != ended_serial || c->times.size() < 2) return false;
    const double t0 = c->times[0], dt = c->times[0] - c->times[1];
    if (!(dt > 0.0)) return false;
    const double tol = std::max(1e-9 * dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(ah.t_last));
    return std::fabs((t0 + 1 * dt) - ah.t_first) <= tol && std::fabs((t0 + L * dt) - ah.t_last) <= tol;
}

// The block that has just been planned can take the rows the pass in the making has left: it was computed for this block's step
// times (up to the tolerance a caller's time is accepted with), completely, under the plan that has just ended.
bool ahead_adoptable(const hc_ctx* c, unsigned long long ended_serial) {
    const auto& ah = c->ahead;
    const auto& pl = c->plan;
    const int L = c->lookahead;
    if (!ah.active || !ah.reduced || ah.plan_serial != ended_serial || !pl.valid) return false;
    for (int j = 0; j < L; ++j)
        if (pl.s_defer[j] >= 0) return false;
    const double tol = std::max(1e-9 * pl.dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(ah.t_last));
    return std::fabs(pl.tgrid[1] - ah.t_first) <= tol && std::fabs(pl.tgrid[L] - ah.t_last) <= tol;
}

// The short pass of the two-level form after block step i0 (hc_plan.hpp: MiniPass): what the samples of the sub-block that has just
// ended contribute to the block steps still to come, added to their rows of P.  The same kernel as the pass of the block, over
// the first few IRF samples only, with the bracket table restricted to those samples (BlockArgs::mini_kw) and a chunking of its
// own (half an IRF sample per chunk -- a function of D only, like every other chunk length).
// next_block (pass schedule "one block ahead", hc_plan.hpp: mini_pass_next): the same for the steps of the NEXT block, added to the
// rows the pass in the making has left; it starts at the first IRF sample those steps take.
void launch_mini_pass(hc_ctx* c, int i0, hipStream_t stream, bool direct, int next_kw, int lane) {
    const auto& pl = c->plan;
    const int L    = c->lookahead;
    const bool next_block = next_kw > 0;  // window length of a short pass towards the NEXT block (0: the in-block short pass)
    const hc::MiniPass mp = next_block ? hc::mini_pass_next(pl, L, i0, next_kw, c->tau) : hc::mini_pass_setup(pl, L, i0, c->tau);
    if (mp.n_samples <= 0 || mp.n_steps <= 0 || mp.s_first >= mp.n_samples) return;
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
    hc::DeviceBuffer<double>& scratch = lane == 2 ? c->d_partials_next : c->d_partials_block;  // the pass lane runs beside lane 0's short passes
    b.partials     = scratch.p;
    b.Dpad         = c->Dpad;
    b.error_flag   = c->d_err.p;
    b.item_counter = c->d_err.p + (lane == 2 ? 2 : 1);
    b.ngroups      = c->ntiles / c->mt_mini;
    b.mini_kw      = mp.kw;
    b.mini_steps   = mp.n_steps;
    for (int k = 0; k <= mp.kw + 1; ++k) b.mini_time[k] = mp.time[k];
    b.chunk_first  = next_block ? static_cast<int>((static_cast<long long>(mp.s_first) * c->D / 8) / b.chunk_gp) : 0;
    b.chunk_last   = b.nchunks;
    require(static_cast<size_t>(b.nchunks) * L * c->Dpad <= scratch.n, HC_ERR_RUNTIME, "short pass: partials buffer too small");
    hc::ReduceArgs r{scratch.p, b.nchunks, 0, c->Dpad, L, rows_P(c, next_block), rows_E(c, next_block), b.item_counter, 1, next_block ? 0 : i0, mp.n_steps,
                     b.chunk_first};
    // Narrow form: through ONE IRF sample the window's kw samples reach kw + 2 consecutive steps at most (the query times of
    // consecutive steps are a step apart, like the samples), so a chunk needs 16 step columns if its samples' live steps fit a
    // window of 16 -- always, for the in-block short passes of sub-blocks of 8 (kw <= 9); a window towards the next block spans the
    // whole block and keeps the wide form.  The live range per IRF sample is taken generously (q inside the span of the view's
    // times: the kernel's own bracket test decides, entries outside simply weigh 0).
    const bool narrow_on = HC_TUNE_INT("HC_MINI_NARROW", 1) != 0;  // (read per launch: tests switch it between contexts)
    if (narrow_on && L == 32 && mp.kw + 2 <= 14 && b.nchunks <= hc::kMiniChunks) {
        bool fits = true;
        const double t_new = mp.time[0], t_old = mp.time[mp.kw + 1];
        for (int ch = b.chunk_first; ch < b.nchunks && fits; ++ch) {
            const int gp0 = ch * b.chunk_gp, gp1 = std::min((b.F + 7) >> 3, gp0 + b.chunk_gp);
            const int sa = (gp0 * 8) / c->D, sb = (std::min(b.F, gp1 * 8) - 1) / c->D;
            int lo = hc::kLookahead, hi = -1;
            for (int s_ = sa; s_ <= sb; ++s_)
                for (int j = 0; j < mp.n_steps; ++j) {
                    const double q = mp.tpred[j] - c->tau[static_cast<size_t>(s_)];
                    if (s_ >= mp.s_cut[j] && q >= t_old && q <= t_new) {
                        lo = std::min(lo, j);
                        hi = std::max(hi, j);
                    }
                }
            if (hi < 0) lo = 0;  // nothing live in this chunk: any window will do
            fits = hi - lo < 16;
            b.mini_jbase[ch] = r.jbase[ch] = static_cast<unsigned char>(lo);
        }
        if (fits) {
            b.depth       = 16;
            b.mini_narrow = r.narrow = 1;
        }
    }
    const bool narrow = b.mini_narrow != 0;
    const int mt      = narrow ? c->mt_narrow : c->mt_mini;
    b.ngroups         = c->ntiles / mt;
    if (direct) {
        hc::BlockArgs b2;
        const hc::BlockLaunch l = hc::block_launch_config(b, mt, &b2);
        if (l.nblocks <= 0) return;
        c->dq->dispatch(narrow ? c->dk_narrow : (L == 32 ? c->dk_mini32 : c->dk_mini16), static_cast<uint32_t>(l.nblocks), 256, static_cast<uint32_t>(l.smem), &b2, sizeof b2,
                        direct_tag(c, hc::kEvMiniPass), 0.0, lane);
        c->dq->dispatch(c->dk_reduce, static_cast<uint32_t>(hc::reduce_block_grid(r)), 256, 0, &r, sizeof r, -1, 0.0, lane);
        c->prof.direct_dispatches += 2;
        if (lane == 2) c->prof.pass_lane_launches += 1;
        return;
    }
    hc::EventPair* ev = ev_begin(c, hc::kEvMiniPass, stream);
    hc::launch_conv_block(b, mt, stream);
    ev_end(ev, stream);
    hc::launch_reduce_block(r, stream);
    c->prof.hip_launches += 2;
}

void pass_lane_drain(hc_ctx* c) {
    if (c->dq && c->dq->busy(2) && !c->dq->drain(20.0, 2)) {
        c->lost = true;
        throw Error(HC_ERR_DEVICE, "the pass lane of the direct queue did not drain: " + c->dq->failure_text());
    }
}

}  // namespace detail
}  // namespace hc
