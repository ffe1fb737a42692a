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
    if (!hc::build_plan(pl, sc.L, known, tau, width, sc.sub)) {
        *worst_out = -1.0;  // not planned (allowed: e.g. block longer than half the window)
        return 0;
    }
    double worst = 0.0;
    std::vector<std::vector<double>> slots(sc.L + 1, std::vector<double>(hc::kTermMax, 0.0));
    std::vector<double> mini(sc.L + 1, 0.0);  // what the short passes of the two-level form have added to the steps' pass rows
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
        for (int k = 0; k < sc.H0; ++k) { ptimes.push_back(known[k]); pvel.push_back(v_known[k]); }
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
                    // history index k of the view = block step m + 1 - k (k = 0: the unknown step m + 1, weight masked)
                    const double vn = (lo >= 1 && lo <= mp.kw) ? v_block[m + 1 - lo] : 0.0;
                    const double vo = (lo + 1 <= mp.kw) ? v_block[m - lo] : 0.0;
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
    };
    int failures = 0, planned = 0;
    for (const auto& sc : list)
        for (unsigned seed = 1; seed <= 3; ++seed) {
            double worst = 0.0;
            const int rc = run(sc, seed, &worst);
            if (worst >= 0.0) ++planned;
            std::printf("dt_rirf %.4f dt_step %.4f S %4d H0 %4d L %2d sub %d seed %u : %s (worst %.2e)\n", sc.dt_rirf, sc.dt_step, sc.S, sc.H0, sc.L,
                        sc.sub, seed, rc == 0 ? (worst < 0.0 ? "not planned" : "ok") : "FAILED", worst);
            failures += rc != 0;
        }
    std::printf("%d scenario runs, %d planned, %d failures\n", static_cast<int>(sizeof list / sizeof list[0]) * 3, planned, failures);
    return (failures == 0 && planned >= 60) ? 0 : 1;
}
