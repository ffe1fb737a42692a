// hc_plan.hpp -- look-ahead plan of the scatter-form evaluation (host only; no HIP dependency, so that the planner is
// unit-tested on CPU: tests/cpp/plan_test.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>
#include <vector>

#include "hc_limits.hpp"

namespace hc {

// Look-ahead plan (scatter form, see hc_kernels.hpp).  Made right after a step at time tgrid[0] has been enqueued: the
// block covers the next 16 or 32 predicted steps tgrid[j] = tgrid[0] + j*dt.  The pass launched with the plan has computed, for
// each block step, what the samples known at planning time contribute; the tables below say what each block step adds.
struct Plan {
    bool valid = false;
    double dt = 0.0;
    int j_next = 1;                       // next block step, 1..L
    double tgrid[2 * kLookahead + 2] = {0};  // predicted times of this block (1..L) and of the one after it (L+1..2L, pass schedule "one block ahead")
    int s_cut[kLookahead]    = {0};       // pass: block step j+1 takes IRF samples s >= s_cut[j]
    int s_defer[kLookahead]  = {0};       // IRF sample whose "is there an older history sample" test is too close to call ahead of time (-1: none)
    // block step m = 1..L (index m): IRF samples involving the step's own sample (weight x width) ...
    int n_own[kLookahead + 1] = {0};
    int own_s[kLookahead + 1][kNearMax];
    double own_a[kLookahead + 1][kNearMax];
    // ... and the number of scatter results of earlier block steps it adds (term slots k = 0 .. n_terms - 1)
    int n_terms[kLookahead + 1] = {0};
    // scatter of block step i, IRF sample s: the (step, term slot, weight) targets of its result
    int n_tgt[kLookahead + 1][kScatterSamples];
    int tgt_step[kLookahead + 1][kScatterSamples][kTargets];
    int tgt_k[kLookahead + 1][kScatterSamples][kTargets];
    double tgt_coef[kLookahead + 1][kScatterSamples][kTargets];
    // scatter launched after block step i covers IRF samples [scat_lo[i], scat_hi[i]] (hi < lo: nothing)
    int scat_lo[kLookahead + 1] = {0}, scat_hi[kLookahead + 1] = {0};
    // Two-level form (wide systems, where the scatter launches re-read K from HBM): the block is cut into sub-blocks of `sub`
    // steps; a scatter result only goes to later steps of its OWN sub-block, and what the samples of a sub-block contribute to the
    // steps of the later sub-blocks is computed once, by a short pass over the head of K right after the sub-block's last step
    // (block step i0 = sub, 2*sub, ... < L): mini_s_hi[i0] = last IRF sample that pass needs (-1: nothing to do).
    int sub = 0;
    int slices = 1;  // term slots one scatter result takes (column slices of the scatter launch; wide systems)
    int mini_s_hi[kLookahead + 1];
    int misses = 0, cooldown = 0;
    bool has_exc = false;  // the pass also left the excitation force of the predicted times (E rows)
    // own_zero (pass schedule "one block ahead"): the sample of the planning step itself -- grid index 0 -- is treated like a block
    // sample: what it contributes to the block's steps comes from a scatter launched with the plan (n_tgt[0] / scat_lo[0] / scat_hi[0])
    // and, in the two-level form, from the short pass after the first sub-block (whose window then starts at index 0), not from the
    // rows the pass left.  The rows of such a block were made BEFORE that sample existed, see hc_pass.cpp.
    bool own_zero = false;
    double t_before_zero = 0.0;  // time of the history sample before grid index 0 (the short pass of the first sub-block needs it)
};

// Plans the block that follows the step just pushed (times[0], newest first).  On the predicted time grid it classifies, for
// every block step m and IRF sample s, the interpolation bracket of the query time tgrid[m] - tau_s by who owns its two
// samples: samples known now (the pass), earlier block steps (scatter targets), step m itself (own entries).  The
// comparisons and the weight arithmetic are those of find_bracket / InterpolateVelocity6D (src/hydro_forces.cpp:343-381).
// Returns false (plan invalid) when a block cannot be planned; pl.cooldown then says for how many steps not to retry.
inline bool build_plan(Plan& pl, int lookahead, const std::deque<double>& times, const std::vector<double>& tau,
                       const std::vector<double>& width, int sub = 0, int slices = 1, bool own_zero = false) {
    const int keep_misses = pl.misses, keep_cool = pl.cooldown;
    pl             = Plan{};
    pl.misses      = keep_misses;
    pl.cooldown    = keep_cool;
    const int H    = static_cast<int>(times.size());
    const int S    = static_cast<int>(tau.size());
    if (lookahead <= 0 || H < 2 || pl.cooldown > 0 || S < 2 || tau.front() < 0.0) return false;
    const double t0 = times[0], dt = times[0] - times[1];
    if (!(dt > 0.0)) return false;
    const int L = lookahead;  // steps per block: 16 or 32
    if (dt * L > 0.5 * (tau.back() - tau.front())) {
        pl.cooldown = 64;  // a block would span most of the IRF window: the scatter launches would re-read most of K every step
        return false;
    }
    pl.dt = dt;
    for (int j = 0; j <= 2 * L + 1; ++j) pl.tgrid[j] = (j == 0) ? t0 : t0 + j * dt;
    auto G = [&](int idx) { return idx >= 1 ? pl.tgrid[idx] : times[static_cast<size_t>(-idx)]; };  // idx > -H
    pl.sub    = (sub > 0 && sub < L) ? sub : 0;
    pl.slices = slices;
    pl.own_zero      = own_zero;
    pl.t_before_zero = times[1];
    const int first_own = own_zero ? 0 : 1;  // smallest grid index whose sample the block's own machinery (not the pass) accounts for
    for (int i = 0; i <= L; ++i) {
        pl.scat_lo[i] = S;
        pl.scat_hi[i] = -1;
        pl.mini_s_hi[i] = -1;
        for (int s = 0; s < kScatterSamples; ++s) pl.n_tgt[i][s] = 0;
    }
    const double oldest = times.back();
    for (int m = 1; m <= L; ++m) {
        const int j = m - 1;
        // pass: samples s >= s_cut[j] of block step m need only history known now and the (zero) not-yet-known sample at
        // tgrid[1]: tgrid[m] - tau_s <= tgrid[1] (same expression as the kernel)
        int sc = 0;
        while (sc < S && !(pl.tgrid[m] - tau[sc] <= pl.tgrid[1])) ++sc;
        pl.s_cut[j] = sc;
        // While the history is shorter than the IRF window the reference's "no older sample -> the IRF step contributes
        // nothing" rule (src/hydro_forces.cpp:604-606) makes the sum discontinuous in t where t - tau_s crosses the oldest
        // sample time; a predicted time that differs from the caller's by an ulp could flip that decision.  The one sample
        // per step for which the test is too close to call is left out of the pass and evaluated by the step itself.
        pl.s_defer[j]       = -1;
        const double margin = 8.0 * std::max(1e-9 * dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(pl.tgrid[m]));
        const double target = pl.tgrid[m] - oldest;  // tau of the sample that lands on the oldest history time
        if (target <= tau.back() + margin) {
            const auto it = std::lower_bound(tau.begin(), tau.end(), target - margin);
            if (it != tau.end() && std::fabs(*it - target) <= margin) pl.s_defer[j] = static_cast<int>(it - tau.begin());
        }
        // brackets that touch a block sample (grid index >= 1)
        for (int s = 0; s < S; ++s) {
            const double q = pl.tgrid[m] - tau[s];
            int lo = 0;  // smallest lo with G(m - lo - 1) <= q
            while (m - lo - 1 > -H && G(m - lo - 1) > q) ++lo;
            const int nm = m - lo, om = nm - 1;
            if (nm < first_own) break;  // both samples are the pass's: it has this bracket, and so it has every later s
            if (om <= -H) break;        // (no older sample; cannot happen for nm >= 1)
            if (s >= kScatterSamples || s == pl.s_defer[j]) return false;
            const double newer = G(nm), older = G(om);
            double wo = 0.0, wn = 0.0;
            if (q == older) { wo = 1.0; wn = 0.0; }
            else if (q == newer) { wo = 0.0; wn = 1.0; }
            else if (q > older && q < newer) {
                const double td = newer - older;
                wo = (td != 0.0) ? ((newer - q) / td) : 0.0;
                wn = 1.0 - wo;
            } else {
                return false;
            }
            const int idx[2]     = {nm, om};
            const double wgt[2]  = {wn, wo};
            for (int e = 0; e < 2; ++e) {
                if (wgt[e] == 0.0 || idx[e] < first_own) continue;
                if (idx[e] == m) {
                    if (pl.n_own[m] >= kNearMax - 1) return false;  // one entry stays free for the deferred sample
                    pl.own_s[m][pl.n_own[m]] = s;
                    pl.own_a[m][pl.n_own[m]] = wgt[e] * width[s];
                    pl.n_own[m]++;
                } else if (pl.sub > 0 && std::max(idx[e] - 1, 0) / pl.sub != (m - 1) / pl.sub) {
                    // another sub-block: the short pass after the last step of the sample's sub-block has this pair
                    // (grid index 0 -- own_zero -- counts to the first sub-block)
                    const int i0     = (std::max(idx[e] - 1, 0) / pl.sub + 1) * pl.sub;
                    pl.mini_s_hi[i0] = std::max(pl.mini_s_hi[i0], s);
                } else {
                    const int i = idx[e];
                    // (wide systems: a scatter result arrives as `slices` column-slice partials, one term slot each)
                    if (pl.n_terms[m] + slices > kTermMax || pl.n_tgt[i][s] >= kTargets) return false;
                    const int k = pl.n_terms[m];
                    pl.n_terms[m] += slices;
                    const int t = pl.n_tgt[i][s]++;
                    pl.tgt_step[i][s][t] = m;
                    pl.tgt_k[i][s][t]    = k;
                    pl.tgt_coef[i][s][t] = wgt[e];
                    pl.scat_lo[idx[e]] = std::min(pl.scat_lo[idx[e]], s);
                    pl.scat_hi[idx[e]] = std::max(pl.scat_hi[idx[e]], s);
                }
            }
        }
    }
    pl.j_next = 1;
    pl.valid  = true;
    return true;
}
// The short pass of the two-level form that follows block step i0 (a multiple of pl.sub, < L): it covers the block steps
// i0 + 1 + j, j < n_steps, and the IRF samples s < n_samples, restricted to the brackets that touch the samples of the sub-block
// that has just ended (history indices 1..kw of the view whose sample 0 is the not-yet-known step i0 + 1).
struct MiniPass {
    int kw = 0, n_steps = 0, n_samples = 0;
    int s_first = 0;                       // first IRF sample any of its steps takes
    double time[kLookahead + 2] = {0};     // predicted-grid times of history indices 0 .. kw + 1
    double tpred[kLookahead]   = {0};
    int s_cut[kLookahead]      = {0};
    int s_defer[kLookahead]    = {0};
};

inline MiniPass mini_pass_setup(const Plan& pl, int lookahead, int i0, const std::vector<double>& tau) {
    MiniPass mp;
    const int S = static_cast<int>(tau.size());
    mp.kw        = pl.sub + ((pl.own_zero && i0 == pl.sub) ? 1 : 0);  // own_zero: the first sub-block's window starts at grid index 0
    mp.n_steps   = lookahead - i0;
    mp.n_samples = std::min(S, pl.mini_s_hi[i0] + 1);
    for (int k = 0; k <= mp.kw + 1; ++k) mp.time[k] = (i0 + 1 - k >= 0) ? pl.tgrid[i0 + 1 - k] : pl.t_before_zero;  // (index -1 only under own_zero)
    for (int j = 0; j < kLookahead; ++j) {
        const int m = i0 + 1 + j;
        mp.tpred[j]   = (j < mp.n_steps) ? pl.tgrid[m] : pl.tgrid[lookahead];
        mp.s_defer[j] = (j < mp.n_steps) ? pl.s_defer[m - 1] : -1;
        int sc = S;  // steps beyond the block: nothing
        if (j < mp.n_steps) {
            sc = 0;
            while (sc < S && !(pl.tgrid[m] - tau[sc] <= pl.tgrid[i0 + 1])) ++sc;  // same expression as the pass of the block
        }
        mp.s_cut[j] = sc;
    }
    mp.s_first = mp.s_cut[0];
    return mp;
}

// Pass schedule "one block ahead" (hc_step.cpp).  The pass of block b + 1 is computed while block b is being stepped, from the
// history known when block b was planned -- the plain pass of the same virtual step at tgrid[1], for the predicted times
// tgrid[L + 1 .. 2L]: FarPass.  What the samples of block b itself contribute to the steps of block b + 1 is added by short passes
// like those of the two-level form, one per window of block-b samples (a sub-block, or the whole block for the single-level form),
// right after the window's last step i0: mini_pass_next.  Both are only used while the history covers the whole IRF window
// (far_pass_allowed), so no "is there an older sample" decision (s_defer) is ever involved.
struct FarPass {
    double tpred[kLookahead] = {0};
    int s_cut[kLookahead]    = {0};
};

inline bool far_pass_allowed(const Plan& pl, int lookahead, const std::deque<double>& times, const std::vector<double>& tau) {
    if (!pl.valid || times.size() < 2 || tau.empty()) return false;
    const double span   = pl.dt * (2 * lookahead + 1);
    const double margin = 8.0 * std::max(1e-9 * pl.dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(pl.tgrid[2 * lookahead]));
    if (span > 0.5 * (tau.back() - tau.front())) return false;          // the short passes towards the next block would stream most of K
    for (int j = 0; j < lookahead; ++j)
        if (pl.s_defer[j] >= 0) return false;
    return pl.tgrid[1] - times.back() > tau.back() + margin;            // every IRF sample of every step has an older history sample
}

inline FarPass far_pass_setup(const Plan& pl, int lookahead, const std::vector<double>& tau) {
    FarPass fp;
    const int S = static_cast<int>(tau.size());
    for (int j = 0; j < kLookahead; ++j) {
        const int m  = std::min(lookahead + 1 + j, 2 * lookahead);
        fp.tpred[j]  = pl.tgrid[m];
        int sc = S;
        if (j < lookahead) {
            sc = 0;
            while (sc < S && !(pl.tgrid[m] - tau[sc] <= pl.tgrid[1])) ++sc;  // same expression as the pass of the block
        }
        fp.s_cut[j] = sc;
    }
    return fp;
}

// The short pass towards the next block after block step i0: window = the kw block samples that end at i0, steps = the L steps of the
// next block.  A block has ONE such window, the samples 1 .. L - 1, launched behind step L - 1 (next_window_end / next_window_length;
// the functions would allow several): the block's LAST sample is not its -- the next block, planned right after it, takes it as its
// own grid index 0 (Plan::own_zero), so that no short pass stands between the last step of a block and the first step of the next.
// (Measured for a C4/8 rank, back to back: windows of 8 / 16 / 31 samples 81.2 / 72.4 / 69.7 us per step -- every window streams the
// head of K up to the block length again -- against 74.6 us with the pass at block start; with host work between the calls all the same.)
inline int next_window_size(const Plan& pl, int lookahead) {
    (void)pl;
    return lookahead;
}
inline bool next_window_end(const Plan& pl, int lookahead, int m) {
    const int kwin = next_window_size(pl, lookahead);
    return m >= 1 && m <= lookahead - 1 && (m == lookahead - 1 || m % kwin == 0);
}
inline int next_window_length(const Plan& pl, int lookahead, int m) {  // m: a window end
    const int kwin = next_window_size(pl, lookahead);
    const int prev = (m % kwin == 0) ? m - kwin : (m / kwin) * kwin;
    return m - prev;
}
inline MiniPass mini_pass_next(const Plan& pl, int lookahead, int i0, int kw, const std::vector<double>& tau) {
    MiniPass mp;
    const int S = static_cast<int>(tau.size());
    mp.kw       = kw;
    mp.n_steps  = lookahead;
    for (int k = 0; k <= mp.kw + 1; ++k) mp.time[k] = pl.tgrid[i0 + 1 - k];  // i0 >= kw
    for (int j = 0; j < kLookahead; ++j) {
        const int m   = std::min(lookahead + 1 + j, 2 * lookahead);
        mp.tpred[j]   = pl.tgrid[m];
        mp.s_defer[j] = -1;
        int sc = S;
        if (j < mp.n_steps) {
            sc = 0;
            while (sc < S && !(pl.tgrid[m] - tau[sc] <= pl.tgrid[i0 + 1])) ++sc;
        }
        mp.s_cut[j] = sc;
    }
    mp.s_first = mp.s_cut[0];
    // last IRF sample whose bracket can touch the window: the query of the LAST step must not be older than the sample before the window
    int hi = mp.s_first - 1;
    while (hi + 1 < S && pl.tgrid[2 * lookahead] - tau[hi + 1] >= pl.tgrid[i0 - mp.kw]) ++hi;
    mp.n_samples = hi + 1;
    return mp;
}

}  // namespace hc
