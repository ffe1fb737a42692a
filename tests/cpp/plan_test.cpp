// CPU unit test of the look-ahead planner (hydrochrono_amd/csrc/hc_plan.hpp: build_plan), no GPU and no HIP needed.
//
// The scatter-form evaluation splits the radiation sum of a block step m by history sample:
//     rad(m) = [samples known at planning time: the pass] + [earlier block steps: scatter targets] + [step m: own entries].
// This program emulates the three parts on the host with scalar velocities and a scalar kernel K[s] (one row, one column)
// and compares their sum with the direct evaluation -- bracket search and interpolation weights as in
// TestHydro::ComputeForceRadiationDampingConv (src/hydro_forces.cpp:343-381, 589-647) over the complete time list -- for
// uniform step sizes equal to, below and above the IRF spacing, both block lengths, short and long histories.
//   usage: plan_test          (exit code 0 = all scenarios agree to 1e-12 relative)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <random>
#include <vector>

#include "../../hydrochrono_amd/csrc/hc_plan.hpp"

namespace {

// interpolation weights of query q in a newest-first time list (times[0] newest); returns false when no older sample exists
bool bracket(const std::vector<double>& times, double q, int* older, int* newer, double* wo, double* wn) {
    const int H = static_cast<int>(times.size());
    int lo = 0;
    while (lo < H - 1 && times[lo + 1] > q) ++lo;  // smallest lo with times[lo + 1] <= q
    if (lo >= H - 1) return false;
    *newer = lo;
    *older = lo + 1;
    const double tn = times[lo], to = times[lo + 1];
    if (q == to) { *wo = 1.0; *wn = 0.0; }
    else if (q == tn) { *wo = 0.0; *wn = 1.0; }
    else if (q > to && q < tn) { *wo = (tn - q) / (tn - to); *wn = 1.0 - *wo; }
    else return false;
    return true;
}

struct Scenario {
    double dt_rirf, dt_step;
    int S, H0, L;
    int sub = 0;  // > 0: the two-level form (sub-blocks of `sub` steps + a short pass after each)
    bool own_zero = false;  // Plan::own_zero: the newest known sample is left out of the pass and treated like a block sample
};

int run(const Scenario& sc, unsigned seed, double* worst_out) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> tau(sc.S), width(sc.S), K(sc.S);
    for (int s = 0; s < sc.S; ++s) {
        tau[s] = s * sc.dt_rirf;
        K[s]   = U(rng) * std::exp(-tau[s]);
    }
    for (int s = 0; s < sc.S; ++s) width[s] = (s == 0 || s == sc.S - 1) ? 0.5 * sc.dt_rirf : sc.dt_rirf;
    // history known at planning time: H0 samples ending at t0 (newest first), then the L block steps on the same grid
    const double t0 = 3.0;
    std::deque<double> known;
    for (int k = 0; k < sc.H0; ++k) known.push_back(t0 - k * sc.dt_step);
    std::vector<double> v_known(sc.H0), v_block(sc.L + 1);
    for (auto& v : v_known) v = U(rng);
    for (auto& v : v_block) v = U(rng);

    hc::Plan pl;
    if (!hc::build_plan(pl, sc.L, known, tau, width, sc.sub, 1, sc.own_zero)) {
        *worst_out = -1.0;  // not planned (allowed: e.g. block longer than half the window)
        return 0;
    }
    double worst = 0.0;
    std::vector<std::vector<double>> slots(sc.L + 1, std::vector<double>(hc::kTermMax, 0.0));
    std::vector<double> mini(sc.L + 1, 0.0);  // what the short passes of the two-level form have added to the steps' pass rows
    v_block[0] = v_known[0];  // grid index 0 = the newest known sample
    if (sc.own_zero)          // ... whose scatter is launched with the plan
        for (int s = pl.scat_lo[0]; s <= pl.scat_hi[0]; ++s)
            for (int t = 0; t < pl.n_tgt[0][s]; ++t) {
                if (pl.tgt_step[0][s][t] < 1 || pl.tgt_step[0][s][t] > sc.L) return 2;
                if (sc.sub > 0 && (pl.tgt_step[0][s][t] - 1) / sc.sub != 0) return 3;  // two-level: index 0 counts to the first sub-block
                slots[pl.tgt_step[0][s][t]][pl.tgt_k[0][s][t]] = pl.tgt_coef[0][s][t] * (width[s] * K[s] * v_block[0]);
            }
    for (int m = 1; m <= sc.L; ++m) {
        // ---- direct evaluation at the predicted time with everything known up to step m ----
        std::vector<double> times;   // newest first: block steps m..1, then the known samples
        std::vector<double> vel;
        for (int i = m; i >= 1; --i) { times.push_back(pl.tgrid[i]); vel.push_back(v_block[i]); }
        for (int k = 0; k < sc.H0; ++k) { times.push_back(known[k]); vel.push_back(v_known[k]); }
        double ref = 0.0, scale = 0.0;
        for (int s = 0; s < sc.S; ++s) {
            int o, n;
            double wo, wn;
            if (!bracket(times, pl.tgrid[m] - tau[s], &o, &n, &wo, &wn)) continue;  // no older sample: contributes nothing
            const double term = K[s] * (wo * vel[o] + wn * vel[n]) * width[s];
            ref += term;
            scale += std::fabs(term);
        }
        // ---- the three parts ----
        // pass: the plain sum of a virtual step at tgrid[1] with zero velocity over samples s >= s_cut (deferred sample left out)
        std::vector<double> ptimes, pvel;
        ptimes.push_back(pl.tgrid[1]);
        pvel.push_back(0.0);
        for (int k = 0; k < sc.H0; ++k) { ptimes.push_back(known[k]); pvel.push_back((sc.own_zero && k == 0) ? 0.0 : v_known[k]); }
        double pass = 0.0;
        for (int s = pl.s_cut[m - 1]; s < sc.S; ++s) {
            if (s == pl.s_defer[m - 1]) continue;
            int o, n;
            double wo, wn;
            if (!bracket(ptimes, pl.tgrid[m] - tau[s], &o, &n, &wo, &wn)) continue;
            pass += K[s] * (wo * pvel[o] + wn * pvel[n]) * width[s];
        }
        // deferred sample: evaluated by the step itself with the complete history
        double defer = 0.0;
        if (pl.s_defer[m - 1] >= 0) {
            const int s = pl.s_defer[m - 1];
            int o, n;
            double wo, wn;
            if (bracket(times, pl.tgrid[m] - tau[s], &o, &n, &wo, &wn)) defer = K[s] * (wo * vel[o] + wn * vel[n]) * width[s];
        }
        // scatter terms left in this step's slots by the earlier block steps
        double terms = 0.0;
        for (int k = 0; k < pl.n_terms[m]; ++k) terms += slots[m][k];
        // own entries
        double own = 0.0;
        for (int e = 0; e < pl.n_own[m]; ++e) own += pl.own_a[m][e] * K[pl.own_s[m][e]] * v_block[m];
        const double got = pass + defer + terms + own + mini[m];
        worst = std::fmax(worst, std::fabs(got - ref) / std::fmax(scale, 1e-300));
        // ---- this step's scatter: y_s = width_s * K_s * v_m into the slots of its targets ----
        if (m < sc.L)
            for (int s = pl.scat_lo[m]; s <= pl.scat_hi[m]; ++s)
                for (int t = 0; t < pl.n_tgt[m][s]; ++t) {
                    if (pl.tgt_step[m][s][t] <= m || pl.tgt_step[m][s][t] > sc.L) return 2;  // targets are later block steps
                    if (sc.sub > 0 && (pl.tgt_step[m][s][t] - 1) / sc.sub != (m - 1) / sc.sub) return 3;  // two-level: own sub-block only
                    slots[pl.tgt_step[m][s][t]][pl.tgt_k[m][s][t]] = pl.tgt_coef[m][s][t] * (width[s] * K[s] * v_block[m]);
                }
        // ---- two-level form: the short pass after the last step of a sub-block (the kernel's table arithmetic: mini_bracket) ----
        if (pl.sub > 0 && m < sc.L && m % pl.sub == 0 && pl.mini_s_hi[m] >= 0) {
            const hc::MiniPass mp = hc::mini_pass_setup(pl, sc.L, m, tau);
            for (int j = 0; j < mp.n_steps; ++j)
                for (int s = mp.s_cut[j]; s < mp.n_samples; ++s) {
                    if (s == mp.s_defer[j]) continue;
                    double wo, wn;
                    int lo;
                    if (!hc::mini_bracket(mp.time, mp.kw, mp.tpred[j] - tau[s], &wo, &wn, &lo)) return 4;
                    // history index k of the view = grid index m + 1 - k (k = 0: the unknown step m + 1, weight masked; own_zero:
                    // the window of the first sub-block reaches grid index 0)
                    const double vn = (lo >= 1 && lo <= mp.kw) ? v_block[m + 1 - lo] : 0.0;
                    const double vo = (lo + 1 <= mp.kw) ? v_block[m - lo] : 0.0;
                    mini[m + 1 + j] += K[s] * (wo * width[s] * vo + wn * width[s] * vn);
                }
        }
    }
    *worst_out = worst;
    return worst <= 1e-12 ? 0 : 1;
}


// Pass schedule "one block ahead": block A is planned and stepped as above; the pass of block B is computed from the history known
// at A's start (far_pass_setup), A's samples reach B's steps through the short passes of mini_pass_next, and B is then planned from
// the complete history and evaluated WITHOUT a pass of its own: far + short passes + B's scatter / own / in-block short passes must
// equal the direct evaluation.
int run_ahead(const Scenario& sc, unsigned seed, double* worst_out) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> tau(sc.S), width(sc.S), K(sc.S);
    for (int s = 0; s < sc.S; ++s) {
        tau[s] = s * sc.dt_rirf;
        K[s]   = U(rng) * std::exp(-tau[s]);
    }
    for (int s = 0; s < sc.S; ++s) width[s] = (s == 0 || s == sc.S - 1) ? 0.5 * sc.dt_rirf : sc.dt_rirf;
    const double t0 = 3.0;
    std::deque<double> known;
    for (int k = 0; k < sc.H0; ++k) known.push_back(t0 - k * sc.dt_step);
    std::vector<double> v_known(sc.H0), vA(sc.L + 1), vB(sc.L + 1);
    for (auto& v : v_known) v = U(rng);
    for (auto& v : vA) v = U(rng);
    for (auto& v : vB) v = U(rng);
    const int L = sc.L;
    hc::Plan plA;
    *worst_out = -1.0;
    if (!hc::build_plan(plA, L, known, tau, width, sc.sub)) return 0;
    if (!hc::far_pass_allowed(plA, L, known, tau)) return 0;
    // ---- far pass: the plain sum of the virtual step at tgrid[1] for the times of block B ----
    const hc::FarPass fp = hc::far_pass_setup(plA, L, tau);
    std::vector<double> ptimes, pvel;
    ptimes.push_back(plA.tgrid[1]);
    pvel.push_back(0.0);
    for (int k = 0; k < sc.H0; ++k) { ptimes.push_back(known[k]); pvel.push_back(v_known[k]); }
    std::vector<double> P_next(L + 1, 0.0);
    for (int j = 0; j < L; ++j)
        for (int s = fp.s_cut[j]; s < sc.S; ++s) {
            int o, n;
            double wo, wn;
            if (!bracket(ptimes, fp.tpred[j] - tau[s], &o, &n, &wo, &wn)) return 5;  // the history covers the window: always bracketed
            P_next[j + 1] += K[s] * (wo * pvel[o] + wn * pvel[n]) * width[s];
        }
    // ---- short passes of block A towards block B: windows that end at the sub-block boundaries below L - 1 and at L - 1 ----
    int covered = 0;
    for (int i0 = 1; i0 <= L - 1; ++i0) {
        if (!hc::next_window_end(plA, L, i0)) continue;
        const int kw = hc::next_window_length(plA, L, i0);
        if (i0 - kw != covered) return 6;  // the windows tile the samples 1 .. L - 1
        covered = i0;
        const hc::MiniPass mp = hc::mini_pass_next(plA, L, i0, kw, tau);
        if (mp.kw != kw || mp.n_steps != L) return 6;
        for (int j = 0; j < L; ++j)
            for (int s = mp.s_cut[j]; s < mp.n_samples; ++s) {
                double wo, wn;
                int lo;
                if (!hc::mini_bracket(mp.time, mp.kw, mp.tpred[j] - tau[s], &wo, &wn, &lo)) return 4;
                const double vn = (lo >= 1 && lo <= mp.kw) ? vA[i0 + 1 - lo] : 0.0;
                const double vo = (lo + 1 <= mp.kw) ? vA[i0 - lo] : 0.0;
                P_next[j + 1] += K[s] * (wo * width[s] * vo + wn * width[s] * vn);
            }
        // nothing of the window may lie beyond n_samples: the query of the last step at the next IRF sample is older than the window
        if (mp.n_samples < sc.S) {
            double wo, wn;
            int lo;
            if (hc::mini_bracket(mp.time, mp.kw, mp.tpred[L - 1] - tau[mp.n_samples], &wo, &wn, &lo) && (wo != 0.0 || wn != 0.0)) return 7;
        }
    }
    if (covered != L - 1) return 6;
    // ---- block B, planned from the complete history (block A's steps were taken at their predicted times) ----
    std::deque<double> knownB;
    std::vector<double> v_knownB;
    for (int i = L; i >= 1; --i) { knownB.push_back(plA.tgrid[i]); v_knownB.push_back(vA[i]); }
    for (int k = 0; k < sc.H0; ++k) { knownB.push_back(known[k]); v_knownB.push_back(v_known[k]); }
    // block A's LAST sample is block B's own grid index 0 (own_zero): scatter launched with the plan + the first sub-block's short pass
    hc::Plan plB;
    if (!hc::build_plan(plB, L, knownB, tau, width, sc.sub, 1, true)) return 8;
    double worst = 0.0;
    std::vector<std::vector<double>> slots(L + 1, std::vector<double>(hc::kTermMax, 0.0));
    std::vector<double> mini(L + 1, 0.0);
    vB[0] = vA[L];
    for (int s = plB.scat_lo[0]; s <= plB.scat_hi[0]; ++s)
        for (int t = 0; t < plB.n_tgt[0][s]; ++t) slots[plB.tgt_step[0][s][t]][plB.tgt_k[0][s][t]] = plB.tgt_coef[0][s][t] * (width[s] * K[s] * vB[0]);
    for (int m = 1; m <= L; ++m) {
        std::vector<double> times, vel;
        for (int i = m; i >= 1; --i) { times.push_back(plB.tgrid[i]); vel.push_back(vB[i]); }
        for (size_t k = 0; k < knownB.size(); ++k) { times.push_back(knownB[k]); vel.push_back(v_knownB[k]); }
        double ref = 0.0, scale = 0.0;
        for (int s = 0; s < sc.S; ++s) {
            int o, n;
            double wo, wn;
            if (!bracket(times, plB.tgrid[m] - tau[s], &o, &n, &wo, &wn)) continue;
            const double term = K[s] * (wo * vel[o] + wn * vel[n]) * width[s];
            ref += term;
            scale += std::fabs(term);
        }
        if (plB.s_defer[m - 1] >= 0) return 9;
        double terms = 0.0, own = 0.0;
        for (int k = 0; k < plB.n_terms[m]; ++k) terms += slots[m][k];
        for (int e = 0; e < plB.n_own[m]; ++e) own += plB.own_a[m][e] * K[plB.own_s[m][e]] * vB[m];
        const double got = P_next[m] + terms + own + mini[m];
        worst = std::fmax(worst, std::fabs(got - ref) / std::fmax(scale, 1e-300));
        if (m < L)
            for (int s = plB.scat_lo[m]; s <= plB.scat_hi[m]; ++s)
                for (int t = 0; t < plB.n_tgt[m][s]; ++t) slots[plB.tgt_step[m][s][t]][plB.tgt_k[m][s][t]] = plB.tgt_coef[m][s][t] * (width[s] * K[s] * vB[m]);
        if (plB.sub > 0 && m < L && m % plB.sub == 0 && plB.mini_s_hi[m] >= 0) {
            const hc::MiniPass mp = hc::mini_pass_setup(plB, L, m, tau);
            for (int j = 0; j < mp.n_steps; ++j)
                for (int s = mp.s_cut[j]; s < mp.n_samples; ++s) {
                    double wo, wn;
                    int lo;
                    if (!hc::mini_bracket(mp.time, mp.kw, mp.tpred[j] - tau[s], &wo, &wn, &lo)) return 4;
                    const double vn = (lo >= 1 && lo <= mp.kw) ? vB[m + 1 - lo] : 0.0;
                    const double vo = (lo + 1 <= mp.kw) ? vB[m - lo] : 0.0;
                    mini[m + 1 + j] += K[s] * (wo * width[s] * vo + wn * width[s] * vn);
                }
        }
    }
    *worst_out = worst;
    return worst <= 1e-12 ? 0 : 1;
}

}  // namespace

int main() {
    const Scenario list[] = {
        {0.01, 0.01, 256, 300, 16},   {0.01, 0.01, 256, 300, 32},   {0.01, 0.007, 256, 400, 32}, {0.01, 0.013, 256, 220, 16},
        {0.015, 0.01, 201, 330, 32},  {0.01, 0.0101, 128, 140, 32}, {0.01, 0.004, 128, 400, 32}, {0.01, 0.02, 512, 300, 16},
        {0.01, 0.01, 256, 40, 16},    /* history shorter than the IRF window: the deferred-sample rule is active */
        {0.01, 0.0101, 1001, 34, 16}, {0.015, 0.0101, 1001, 18, 16}, {0.01, 0.01, 64, 100, 32} /* block = half the window */,
        // the two-level form (sub-blocks of 8 steps + short passes): the same grid of step sizes and history lengths
        {0.01, 0.01, 256, 300, 32, 8},  {0.01, 0.01, 256, 300, 16, 8},   {0.01, 0.007, 256, 400, 32, 8},  {0.01, 0.013, 256, 220, 16, 8},
        {0.015, 0.01, 201, 330, 32, 8}, {0.01, 0.0101, 128, 140, 32, 8}, {0.01, 0.004, 128, 400, 32, 8},  {0.01, 0.02, 512, 300, 16, 8},
        {0.01, 0.01, 256, 40, 16, 8},   {0.01, 0.0101, 1001, 34, 16, 8}, {0.015, 0.0101, 1001, 18, 32, 8}, {0.01, 0.01, 64, 100, 32, 8},
        {0.01, 0.01, 256, 300, 32, 4},  {0.01, 0.0037, 256, 600, 32, 8},
        // own_zero: the newest known sample handled by the block's own machinery (scatter with the plan / first sub-block's short pass)
        {0.01, 0.01, 256, 300, 32, 0, true},   {0.01, 0.01, 256, 300, 16, 0, true},   {0.01, 0.007, 256, 400, 32, 0, true},
        {0.01, 0.013, 256, 220, 16, 0, true},  {0.015, 0.01, 201, 330, 32, 0, true},  {0.01, 0.004, 128, 400, 32, 0, true},
        {0.01, 0.01, 256, 300, 32, 8, true},   {0.01, 0.01, 256, 300, 16, 8, true},   {0.01, 0.007, 256, 400, 32, 8, true},
        {0.01, 0.013, 256, 220, 16, 8, true},  {0.015, 0.01, 201, 330, 32, 8, true},  {0.01, 0.004, 128, 400, 32, 8, true},
        {0.01, 0.01, 256, 300, 32, 4, true},   {0.01, 0.02, 512, 300, 16, 8, true},   {0.01, 0.01, 256, 40, 16, 8, true},
    };
    int failures = 0, planned = 0;
    for (const auto& sc : list)
        for (unsigned seed = 1; seed <= 3; ++seed) {
            double worst = 0.0;
            const int rc = run(sc, seed, &worst);
            if (worst >= 0.0) ++planned;
            std::printf("dt_rirf %.4f dt_step %.4f S %4d H0 %4d L %2d sub %d%s seed %u : %s (worst %.2e, rc %d)\n", sc.dt_rirf, sc.dt_step, sc.S, sc.H0,
                        sc.L, sc.sub, sc.own_zero ? " own_zero" : "", seed, rc == 0 ? (worst < 0.0 ? "not planned" : "ok") : "FAILED", worst, rc);
            failures += rc != 0;
        }
    std::printf("%d scenario runs, %d planned, %d failures\n", static_cast<int>(sizeof list / sizeof list[0]) * 3, planned, failures);
    // pass schedule "one block ahead": histories that cover the IRF window (the schedule is not used before that)
    const Scenario ahead[] = {
        {0.01, 0.01, 256, 300, 32},     {0.01, 0.01, 256, 300, 16},     {0.01, 0.007, 256, 400, 32},    {0.01, 0.013, 256, 220, 16},
        {0.015, 0.01, 201, 330, 32},    {0.01, 0.0101, 128, 140, 32},   {0.01, 0.004, 128, 400, 32},    {0.01, 0.02, 512, 300, 16},
        {0.01, 0.01, 256, 300, 32, 8},  {0.01, 0.01, 256, 300, 16, 8},  {0.01, 0.007, 256, 400, 32, 8}, {0.01, 0.013, 256, 220, 16, 8},
        {0.015, 0.01, 201, 330, 32, 8}, {0.01, 0.0101, 128, 140, 32, 8}, {0.01, 0.004, 128, 400, 32, 8}, {0.01, 0.02, 512, 300, 16, 8},
        {0.01, 0.01, 256, 300, 32, 4},  {0.01, 0.0037, 256, 600, 32, 8}, {0.01, 0.01, 1024, 1100, 32},   {0.01, 0.01, 1024, 1100, 32, 8},
    };
    int planned_ahead = 0;
    for (const auto& sc : ahead)
        for (unsigned seed = 1; seed <= 2; ++seed) {
            double worst = 0.0;
            const int rc = run_ahead(sc, seed, &worst);
            if (worst >= 0.0) ++planned_ahead;
            std::printf("ahead: dt_rirf %.4f dt_step %.4f S %4d H0 %4d L %2d sub %d seed %u : %s (worst %.2e, rc %d)\n", sc.dt_rirf, sc.dt_step, sc.S, sc.H0,
                        sc.L, sc.sub, seed, rc == 0 ? (worst < 0.0 ? "not planned" : "ok") : "FAILED", worst, rc);
            failures += rc != 0;
        }
    std::printf("one block ahead: %d planned, %d failures in total\n", planned_ahead, failures);
    return (failures == 0 && planned >= 60 && planned_ahead >= 30) ? 0 : 1;
}
