// host_path_c.cpp -- what one synchronous hc_step costs a C / C++ caller (no Python wrapper around it): synthetic systems of
// 1, 2 and 64 bodies generated in HBM (S = 1001 / 1001 / 1024 IRF samples), no waves, prescribed motion, steady-state history.
//   g++ -O2 -std=c++17 profiles/host_path_c.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/host_path_c
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hydrochrono_amd.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(int N, int S, double gap_us) {
    // the stepping thread next to its GPU (BIND=0: wherever the scheduler puts it)
    if (!(std::getenv("BIND") && std::atoi(std::getenv("BIND")) == 0)) (void)hc_bind_thread_to_device(0);
    hc_ctx* c = nullptr;
    if (hc_create(N, 0, &c) != HC_OK) { std::printf("hc_create: %s\n", hc_last_error(nullptr)); return 1; }
    if (hc_synth_fill(c, 20251031ull, S, 0.01, 0, 0.01) != HC_OK || hc_finalize(c) != HC_OK || hc_set_wave_none(c, N) != HC_OK) {
        std::printf("setup: %s\n", hc_last_error(c));
        return 1;
    }
    const int D = 6 * N, n3 = 3 * N;
    std::vector<double> pos(n3), rpy(n3), lin(n3), ang(n3), out(D), aw(D, 0.5), aR(D);
    auto state = [&](double t) {
        for (int k = 0; k < n3; ++k) {
            pos[k] = 0.1 * std::sin(1.1 * t + k);
            rpy[k] = 0.05 * std::sin(0.7 * t + 2 * k);
            lin[k] = 0.11 * std::cos(1.1 * t + k);
            ang[k] = 0.035 * std::cos(0.7 * t + 2 * k);
        }
    };
    const int warm = S + 80, reps = gap_us > 0.0 ? 1500 : 4000;
    std::vector<double> ts, ta;
    double t = 0.0;
    for (int n = 0; n < warm + reps; ++n, t += 0.01) {
        state(t);
        const double a = now_us();
        const int rc = hc_step(c, t, pos.data(), rpy.data(), lin.data(), ang.data(), out.data());
        const double b = now_us();
        if (rc != HC_OK) { std::printf("hc_step: %s\n", hc_last_error(c)); return 1; }
        std::fill(aR.begin(), aR.end(), 0.0);
        const double a2 = now_us();
        hc_added_mass_mv(c, aw.data(), 1.0, aR.data(), D);
        const double b2 = now_us();
        if (n >= warm) { ts.push_back(b - a); ta.push_back(b2 - a2); }
        if (gap_us > 0.0) {  // a Chrono-like loop: the host works between two force evaluations
            const double g0 = now_us();
            while (now_us() - g0 < gap_us) {
            }
        }
    }
    std::sort(ts.begin(), ts.end());
    std::sort(ta.begin(), ta.end());
    double mean = 0;
    for (double v : ts) mean += v;
    if (gap_us > 0.0) std::printf("(%.0f us of host work between calls) ", gap_us);
    std::printf("N = %2d, S = %4d, %s: hc_step median %6.2f us  mean %6.2f  p10 %6.2f  p90 %6.2f | hc_added_mass_mv median %6.2f us\n", N, S,
                hc_direct_dispatch_active(c) ? "direct AQL dispatch" : "HIP launches       ", ts[ts.size() / 2], mean / ts.size(), ts[ts.size() / 10],
                ts[ts.size() * 9 / 10], ta[ta.size() / 2]);
    hc_destroy(c);
    return 0;
}

int main(int argc, char** argv) {
    const double gap_us = argc > 1 ? std::atof(argv[1]) : 0.0;  // optional: microseconds of host work between calls
    for (int N : {1, 2, 64})
        if (run(N, N == 64 ? 1024 : 1001, gap_us)) return 1;
    return 0;
}
