// ahead_probe.cpp -- what the pass schedule (hc_set_pass_schedule) does to the latency of hc_step for a caller that leaves the GPU
// idle between two force evaluations: 64 bodies (C3 size, S = 1024) and the rows of 64 bodies of a 512-body array (one rank's share
// of C4), no waves, prescribed motion, steady-state history; host work of 0 / 30 / 100 / 300 us between calls, both schedules.
//   g++ -O2 -std=c++17 profiles/ahead_probe.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/ahead_probe
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hydrochrono_amd.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(int N, int b0, int b1, int S, double gap_us, int schedule, int slices_report) {
    // the stepping thread next to its GPU (BIND=0: wherever the scheduler puts it)
    if (!(std::getenv("BIND") && std::atoi(std::getenv("BIND")) == 0)) (void)hc_bind_thread_to_device(0);
    hc_ctx* c = nullptr;
    const int rc0 = (b1 - b0 == N) ? hc_create(N, 0, &c) : hc_create_sharded(N, b0, b1, 0, &c);
    if (rc0 != HC_OK) { std::printf("create: %s\n", hc_last_error(nullptr)); return 1; }
    if (hc_synth_fill(c, 20251031ull, S, 0.01, 0, 0.01) != HC_OK || hc_finalize(c) != HC_OK || hc_set_wave_none(c, N) != HC_OK ||
        hc_set_pass_schedule(c, schedule, 0) != HC_OK) {
        std::printf("setup: %s\n", hc_last_error(c));
        return 1;
    }
    const int n3 = 3 * N, Dl = 6 * (b1 - b0);
    std::vector<double> pos(n3), rpy(n3), lin(n3), ang(n3), out(Dl);
    auto state = [&](double t) {
        for (int k = 0; k < n3; ++k) {
            pos[k] = 0.1 * std::sin(1.1 * t + k);
            rpy[k] = 0.05 * std::sin(0.7 * t + 2 * k);
            lin[k] = 0.11 * std::cos(1.1 * t + k);
            ang[k] = 0.035 * std::cos(0.7 * t + 2 * k);
        }
    };
    const int warm = S + 100, reps = 32 * 24;
    std::vector<double> ts;
    // the states of the timed steps are made beforehand: at 512 bodies the 6144 sines and cosines of one state take the host ~80 us,
    // which would be host work between the calls that the gap argument does not show
    std::vector<std::vector<double>> pre(static_cast<size_t>(reps) * 4);
    for (int n = 0; n < reps; ++n) {
        state(0.01 * (warm + n));
        pre[4 * n] = pos; pre[4 * n + 1] = rpy; pre[4 * n + 2] = lin; pre[4 * n + 3] = ang;
    }
    double t = 0.0;
    for (int n = 0; n < warm + reps; ++n) {
        t = 0.01 * n;
        if (n < warm) state(t);
        const double* sp[4] = {pos.data(), rpy.data(), lin.data(), ang.data()};
        if (n >= warm)
            for (int q = 0; q < 4; ++q) sp[q] = pre[4 * (n - warm) + q].data();
        const double a = now_us();
        const int rc = hc_step(c, t, sp[0], sp[1], sp[2], sp[3], out.data());
        const double b = now_us();
        if (rc != HC_OK) { std::printf("hc_step: %s\n", hc_last_error(c)); return 1; }
        if (n >= warm) ts.push_back(b - a);
        if (gap_us > 0.0) {
            const double g0 = now_us();
            while (now_us() - g0 < gap_us) {
            }
        }
    }
    hc_profile_stats p{};
    hc_get_profile(c, &p);
    if (std::getenv("BY_POSITION")) {  // mean latency by position in the 32-step pattern (the timed region starts at step `warm`)
        std::printf("   by step %% 32 (first timed step = %d %% 32):", warm);
        for (int q = 0; q < 32; ++q) {
            double m = 0;
            int cnt = 0;
            for (size_t i = q; i < ts.size(); i += 32) { m += ts[i]; ++cnt; }
            std::printf(" %.0f", m / cnt);
        }
        std::printf("\n");
    }
    std::sort(ts.begin(), ts.end());
    double mean = 0;
    for (double v : ts) mean += v;
    mean /= ts.size();
    std::printf("rows of %3d of %3d bodies, gap %3.0f us, schedule %2d: hc_step mean %7.2f us  median %7.2f  p90 %7.2f  p99 %8.2f  max %8.2f   (blocks without a pass of their own: %lld, slices %lld, on the pass lane %lld; schedule answers ahead / at start: %lld / %lld)\n",
                b1 - b0, N, gap_us, schedule, mean, ts[ts.size() / 2], ts[ts.size() * 9 / 10], ts[ts.size() * 99 / 100], ts.back(), p.ahead_blocks,
                p.ahead_pass_slices, p.pass_lane_launches, p.schedule_blocks_ahead, p.schedule_blocks_at_start);
    (void)slices_report;
    hc_destroy(c);
    return 0;
}

int main(int argc, char** argv) {
    const bool wide = argc > 1 && std::atoi(argv[1]) != 0;  // 1: the C4 rank share (9.7 GB of K) as well
    if (argc > 3) {  // one configuration only (for a profiler): <wide> <schedule> <gap in us>
        return wide ? run(512, 0, 64, 1024, std::atof(argv[3]), std::atoi(argv[2]), 0) : run(64, 0, 64, 1024, std::atof(argv[3]), std::atoi(argv[2]), 0);
    }
    // schedule -1 = the library's default (adaptive: per block from the caller's gaps, hc_set_pass_schedule)
    if (const char* sr = std::getenv("SHARD_ROWS")) {  // bigger row shards of the 512-body array (128 / 256 / 512 bodies): where does "ahead" begin to pay?
        const int rows = std::atoi(sr);
        for (double gap : {0.0, 5.0, 10.0, 20.0, 50.0, 100.0})
            for (int schedule : {0, 1})
                if (run(512, 0, rows, 1024, gap, schedule, 0)) return 1;
        return 0;
    }
    const bool fine = std::getenv("FINE_GAPS") != nullptr;  // the crossover between the two schedules (threshold of the adaptive rule)
    if (fine) {
        for (double gap : {0.0, 1.0, 2.0, 3.0, 5.0, 8.0, 12.0, 20.0})
            for (int schedule : {0, 1, -1})
                if (run(64, 0, 64, 1024, gap, schedule, 0)) return 1;
        if (wide)
            for (double gap : {0.0, 2.0, 5.0, 10.0, 20.0})
                for (int schedule : {0, 1, -1})
                    if (run(512, 0, 64, 1024, gap, schedule, 0)) return 1;
        return 0;
    }
    for (double gap : {0.0, 30.0, 100.0, 300.0})
        for (int schedule : {0, 1, -1})
            if (run(64, 0, 64, 1024, gap, schedule, 0)) return 1;
    if (wide)
        for (double gap : {0.0, 10.0, 100.0, 300.0, 1000.0})
            for (int schedule : {0, 1, -1})
                if (run(512, 0, 64, 1024, gap, schedule, 0)) return 1;
    return 0;
}
