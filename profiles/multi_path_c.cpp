// multi_path_c.cpp -- what one hc_step_multi (and hc_added_mass_mv_multi) costs a C / C++ caller: a synthetic coupled array
// row-sharded over G contexts of THIS process (all on GPU 0 of this box), no waves, prescribed motion, steady-state history.
//   g++ -O2 -std=c++17 profiles/multi_path_c.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/multi_path_c
//   /tmp/multi_path_c [N = 64] [S = 1024] [reps = 3000]
// GAP_US=<us>: host work (a busy wait) between the calls, as a Chrono loop has it -- with HC_MULTI_SPIN_US this shows what a worker
// thread that has gone to sleep costs the next call (profiles/r05/multi_spin.txt); ONLY_G=<G>: one group size only.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hydrochrono_amd.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(int N, int S, int G, int reps, std::vector<double>* ref) {
    std::vector<hc_ctx*> ctxs;
    const int base = N / G, extra = N % G;
    for (int g = 0; g < G; ++g) {
        const int b0 = g * base + std::min(g, extra), b1 = b0 + base + (g < extra ? 1 : 0);
        hc_ctx* c = nullptr;
        if (hc_create_sharded(N, b0, b1, 0, &c) != HC_OK) { std::printf("hc_create_sharded: %s\n", hc_last_error(nullptr)); return 1; }
        if (hc_synth_fill(c, 20251031ull, S, 0.01, 0, 0.01) != HC_OK || hc_finalize(c) != HC_OK || hc_set_wave_none(c, N) != HC_OK) {
            std::printf("setup: %s\n", hc_last_error(c));
            return 1;
        }
        ctxs.push_back(c);
    }
    const int D = 6 * N, n3 = 3 * N;
    std::vector<double> pos(n3), rpy(n3), lin(n3), ang(n3), out(D), aw(D, 0.5), aR(D), last;
    auto state = [&](double t) {
        for (int k = 0; k < n3; ++k) {
            pos[k] = 0.1 * std::sin(1.1 * t + k);
            rpy[k] = 0.05 * std::sin(0.7 * t + 2 * k);
            lin[k] = 0.11 * std::cos(1.1 * t + k);
            ang[k] = 0.035 * std::cos(0.7 * t + 2 * k);
        }
    };
    const int warm = S + 80;
    std::vector<double> ts, ta;
    double t = 0.0;
    for (int n = 0; n < warm + reps; ++n, t += 0.01) {
        state(t);
        const double a = now_us();
        const int rc = hc_step_multi(ctxs.data(), G, t, pos.data(), rpy.data(), lin.data(), ang.data(), out.data());
        const double b = now_us();
        if (rc != HC_OK) { std::printf("hc_step_multi: %s\n", hc_last_error(ctxs[0])); return 1; }
        std::fill(aR.begin(), aR.end(), 0.0);
        const double a2 = now_us();
        hc_added_mass_mv_multi(ctxs.data(), G, aw.data(), 1.0, aR.data(), D);
        const double b2 = now_us();
        if (n >= warm) { ts.push_back(b - a); ta.push_back(b2 - a2); }
        static const double gap_us = std::getenv("GAP_US") ? std::atof(std::getenv("GAP_US")) : 0.0;
        if (gap_us > 0.0 && n >= warm - 64) {
            const double g0 = now_us();
            while (now_us() - g0 < gap_us) {
            }
        }
    }
    last = out;
    bool same = true;
    if (ref->empty()) *ref = last;
    else same = std::equal(last.begin(), last.end(), ref->begin());
    std::sort(ts.begin(), ts.end());
    std::sort(ta.begin(), ta.end());
    double mean = 0;
    for (double v : ts) mean += v;
    int direct = 0;
    for (hc_ctx* c : ctxs) direct += hc_direct_dispatch_active(c);
    std::printf("N = %3d, S = %4d, G = %d contexts (%d with direct dispatch): hc_step_multi median %7.2f us  mean %7.2f  p10 %7.2f  p90 %7.2f | "
                "hc_added_mass_mv_multi median %6.2f us | last forces %s the G = 1 run\n", N, S, G, direct, ts[ts.size() / 2], mean / ts.size(),
                ts[ts.size() / 10], ts[ts.size() * 9 / 10], ta[ta.size() / 2], same ? "bitwise equal to" : "DIFFER from");
    // when each context's step kernel was handed to its GPU, from the entry of hc_step_multi (mean over all calls): the fan-out
    std::printf("        doorbell offsets from the entry of the call, per context (us):");
    double worst = 0.0;
    for (hc_ctx* c : ctxs) {
        hc_profile_stats p;
        hc_get_profile(c, &p);
        const double m = p.multi_calls > 0 ? 1e6 * p.multi_doorbell_offset_sum / p.multi_calls : 0.0;
        worst = std::max(worst, m);
        std::printf(" %.2f", m);
    }
    std::printf("  -> last doorbell %.2f us after entry (HC_MULTI_THREADS=%s)\n", worst, std::getenv("HC_MULTI_THREADS") ? std::getenv("HC_MULTI_THREADS") : "default");
    for (hc_ctx* c : ctxs) hc_destroy(c);
    return same ? 0 : 1;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 64, S = argc > 2 ? std::atoi(argv[2]) : 1024, reps = argc > 3 ? std::atoi(argv[3]) : 3000;
    std::vector<double> ref;
    const int only = std::getenv("ONLY_G") ? std::atoi(std::getenv("ONLY_G")) : 0;
    for (int G : {1, 2, 4, 8})
        if (G <= N && (only == 0 || only == G) && run(N, S, G, reps, &ref)) return 1;  // (ONLY_G: no G = 1 reference, the comparison is with itself)
    return 0;
}
