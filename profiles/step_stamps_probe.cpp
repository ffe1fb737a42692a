// step_stamps_probe.cpp -- where the time of one synchronous hc_step goes at C3 size (64 bodies, S = 1024, no waves, prescribed
// motion, steady-state history, C++ caller stepping back to back or with a gap): the host's stamps (begin of the step, doorbell of the
// step kernel, totals seen) and the STAGE CLOCK of the step kernel's workgroups (tuning build: s_memrealtime at fixed points of
// finalize_kernel<4, true>, converted to the host's HSA clock by the runtime) on one time axis.
//   g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd_tuning
//       -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/step_stamps_probe        (HYDROCHRONO_AMD_FLAVOR is not read here: the link decides)
//   /tmp/step_stamps_probe [gap_us] [schedule]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hydrochrono_amd.h"

extern "C" int hc_tuning_enable_step_stamps(hc_ctx*, int);
extern "C" int hc_tuning_step_stamps(hc_ctx*, unsigned long long seq, double* host_us3, double* wg_us, int* n_wg, int* n_stage);

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) {
    if (v.empty()) return NAN;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
static double pct(std::vector<double> v, double p) {
    if (v.empty()) return NAN;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, static_cast<size_t>(p * v.size()))];
}

int main(int argc, char** argv) {
    const double gap_us = argc > 1 ? std::atof(argv[1]) : 0.0;
    const int schedule  = argc > 2 ? std::atoi(argv[2]) : 0;
    const int N = 64, S = 1024;
    // the stepping thread next to its GPU (BIND=0: wherever the scheduler puts it)
    if (!(std::getenv("BIND") && std::atoi(std::getenv("BIND")) == 0)) (void)hc_bind_thread_to_device(0);
    hc_ctx* c = nullptr;
    if (hc_create(N, 0, &c) != HC_OK) { std::printf("hc_create: %s\n", hc_last_error(nullptr)); return 1; }
    if (hc_synth_fill(c, 20251031ull, S, 0.01, 0, 0.01) != HC_OK || hc_finalize(c) != HC_OK || hc_set_wave_none(c, N) != HC_OK ||
        hc_set_pass_schedule(c, schedule, 0) != HC_OK) {
        std::printf("setup: %s\n", hc_last_error(c));
        return 1;
    }
    const int n3 = 3 * N, D = 6 * N;
    std::vector<double> pos(n3), rpy(n3), lin(n3), ang(n3), out(D);
    auto state = [&](double t) {
        for (int k = 0; k < n3; ++k) {
            pos[k] = 0.1 * std::sin(1.1 * t + k);
            rpy[k] = 0.05 * std::sin(0.7 * t + 2 * k);
            lin[k] = 0.11 * std::cos(1.1 * t + k);
            ang[k] = 0.035 * std::cos(0.7 * t + 2 * k);
        }
    };
    const int warm = S + 100, blocks = 24;
    double t = 0.0;
    int n = 0;
    for (; n < warm; ++n) {
        t = 0.01 * n;
        state(t);
        if (hc_step(c, t, pos.data(), rpy.data(), lin.data(), ang.data(), out.data()) != HC_OK) { std::printf("hc_step: %s\n", hc_last_error(c)); return 1; }
    }
    hc_tuning_enable_step_stamps(c, 1);
    constexpr int NS = 12, NW = 64;
    static const char* stage_name[NS] = {"entry", "arguments in registers", "every load requested", "right-hand side in LDS (state in)", "first barrier passed",
                                         "contraction done (K in)", "second barrier passed", "totals formed", "stores issued", "stores acknowledged", "", ""};
    std::vector<double> call_us, h_begin_to_bell, h_bell_to_seen, bell_to_first_entry, entry_skew, last_ack_to_seen, kernel_span, push_entry, push_ack;
    std::vector<double> crit_delta[NS], mean_delta[NS], crit_abs[NS], extra_a, extra_b;
    std::vector<double> call_by_pos[32];
    int taken = 0, skipped = 0, last_tiles = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned long long seq0 = 0;
        std::vector<double> calls(32);
        for (int k = 0; k < 32; ++k, ++n) {
            t = 0.01 * n;
            state(t);
            const double a = now_us();
            if (hc_step(c, t, pos.data(), rpy.data(), lin.data(), ang.data(), out.data()) != HC_OK) { std::printf("hc_step: %s\n", hc_last_error(c)); return 1; }
            calls[k] = now_us() - a;
            if (k == 0) { hc_step_sequence(c, &seq0); }
            if (gap_us > 0.0) {
                const double g0 = now_us();
                while (now_us() - g0 < gap_us) {
                }
            }
        }
        for (int k = 0; k < 32; ++k) {
            double h3[3], wg[NW * NS];
            int nwg = 0, nst = 0;
            call_by_pos[(n - 32 + k) % 32].push_back(calls[k]);
            if (hc_tuning_step_stamps(c, seq0 + k, h3, wg, &nwg, &nst) != HC_OK) { ++skipped; continue; }
            ++taken;
            call_us.push_back(calls[k]);
            h_begin_to_bell.push_back(h3[1] - h3[0]);
            h_bell_to_seen.push_back(h3[2] - h3[1]);
            const int tiles = nwg - 1;  // the last workgroup stores the sample
            last_tiles = tiles;
            double first_entry = 1e30, last_entry = -1e30, last_ack = -1e30;
            int crit = 0;
            for (int w = 0; w < tiles; ++w) {
                first_entry = std::min(first_entry, wg[w * NS + 0]);
                last_entry  = std::max(last_entry, wg[w * NS + 0]);
                if (wg[w * NS + 9] > last_ack) { last_ack = wg[w * NS + 9]; crit = w; }
            }
            bell_to_first_entry.push_back(first_entry - h3[1]);
            entry_skew.push_back(last_entry - first_entry);
            last_ack_to_seen.push_back(h3[2] - last_ack);
            kernel_span.push_back(last_ack - first_entry);
            push_entry.push_back(wg[tiles * NS + 0] - h3[1]);
            push_ack.push_back(wg[tiles * NS + 9] - h3[1]);
            for (int s = 1; s <= 9; ++s) {
                crit_delta[s].push_back(wg[crit * NS + s] - wg[crit * NS + s - 1]);
                double m = 0;
                for (int w = 0; w < tiles; ++w) m += wg[w * NS + s] - wg[w * NS + s - 1];
                mean_delta[s].push_back(m / tiles);
            }
            for (int s = 0; s <= 9; ++s) crit_abs[s].push_back(wg[crit * NS + s] - h3[1]);
            if (wg[crit * NS + 10] > 0.0 && wg[crit * NS + 11] > 0.0) { extra_a.push_back(wg[crit * NS + 10] - h3[1]); extra_b.push_back(wg[crit * NS + 11] - h3[1]); }
        }
    }
    hc_profile_stats p{};
    hc_get_profile(c, &p);
    std::printf("C3 (64 bodies, S = 1024), gap %.0f us, pass schedule %d: %d steps with stamps (%d without: block starts / plain steps)\n", gap_us, schedule, taken, skipped);
    std::printf("hc_step as the caller sees it (stamped steps only): median %.2f us  p10 %.2f  p90 %.2f\n", med(call_us), pct(call_us, 0.1), pct(call_us, 0.9));
    std::printf("  mean call by position in the block:");
    for (int q = 0; q < 32; ++q) {
        double m = 0;
        for (double v : call_by_pos[q]) m += v;
        std::printf(" %.0f", call_by_pos[q].empty() ? 0.0 : m / call_by_pos[q].size());
    }
    std::printf("\n");
    std::printf("host:   begin of the step -> doorbell of the step kernel          %6.2f us\n", med(h_begin_to_bell));
    std::printf("host:   doorbell -> every row's total seen                          %6.2f us   (p10 %.2f  p90 %.2f)\n", med(h_bell_to_seen), pct(h_bell_to_seen, 0.1),
                pct(h_bell_to_seen, 0.9));
    std::printf("GPU:    doorbell -> entry of the first tile workgroup               %6.2f us   (p10 %.2f  p90 %.2f)\n", med(bell_to_first_entry), pct(bell_to_first_entry, 0.1),
                pct(bell_to_first_entry, 0.9));
    std::printf("GPU:    first -> last tile workgroup's entry (%d workgroups)        %6.2f us\n", last_tiles, med(entry_skew));
    std::printf("GPU:    first entry -> last acknowledged store (the kernel's span)  %6.2f us\n", med(kernel_span));
    std::printf("        last acknowledged store -> totals seen by the host          %6.2f us\n", med(last_ack_to_seen));
    std::printf("        (the workgroup that stores the sample: entry %+.2f us, stores acknowledged %+.2f us after the doorbell)\n", med(push_entry), med(push_ack));
    std::printf("stage                                   | critical workgroup: +us (median), at us after the doorbell | mean over the tile workgroups\n");
    std::printf("  %-38s|          %6s   %6.2f |\n", stage_name[0], "", med(crit_abs[0]));
    for (int s = 1; s <= 9; ++s)
        std::printf("  %-38s|          %+6.2f   %6.2f | %+6.2f\n", stage_name[s], med(crit_delta[s]), med(crit_abs[s]), med(mean_delta[s]));
    if (!extra_a.empty() && med(extra_a) > 0.0)
        std::printf("  (step_hot_kernel, critical workgroup, wave 0: hydrostatic / wave terms formed at %.2f us, its last K word in at %.2f us after the doorbell)\n", med(extra_a), med(extra_b));
    std::printf("(dispatches: %lld direct, %lld HIP; steps with the state behind the arguments: %lld; blocks ahead / at start: %lld / %lld)\n", p.direct_dispatches, p.hip_launches,
                p.slot_state_steps, p.schedule_blocks_ahead, p.schedule_blocks_at_start);
    hc_destroy(c);
    return 0;
}
