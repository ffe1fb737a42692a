// warm_probe.hip -- does a resident "keeper" wave keep the launch-to-result latency of hc_step at its tight-loop value when the host
// works 100+ us between calls?  (profiles/host_path_c.cpp: 13.6 us back to back, 19.1 us with >= 100 us between calls, on every size.)
//   hipcc --offload-arch=gfx950 -O2 profiles/r03/warm_probe.hip -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/warm_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hydrochrono_amd.h"

__global__ void keeper(volatile int* stop, int sleep_arg) {
    while (!*stop) {
        if (sleep_arg > 0) __builtin_amdgcn_s_sleep(127);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double run(hc_ctx* c, int N, double gap_us, int reps) {
    const int n3 = 3 * N, D = 6 * N;
    std::vector<double> pos(n3), rpy(n3), lin(n3), ang(n3), out(D);
    static double t = 0.0;
    std::vector<double> ts;
    for (int n = 0; n < reps; ++n, t += 0.01) {
        for (int k = 0; k < n3; ++k) { pos[k] = 0.1 * std::sin(1.1 * t + k); lin[k] = 0.11 * std::cos(1.1 * t + k); rpy[k] = ang[k] = 0.0; }
        const double a = now_us();
        if (hc_step(c, t, pos.data(), rpy.data(), lin.data(), ang.data(), out.data()) != HC_OK) { std::printf("%s\n", hc_last_error(c)); std::exit(1); }
        ts.push_back(now_us() - a);
        const double g0 = now_us();
        while (now_us() - g0 < gap_us) {}
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int N = 64, S = 1024;
    hc_ctx* c = nullptr;
    if (hc_create(N, 0, &c) != HC_OK || hc_synth_fill(c, 20251031ull, S, 0.01, 0, 0.01) != HC_OK || hc_finalize(c) != HC_OK || hc_set_wave_none(c, N) != HC_OK) return 1;
    run(c, N, 0.0, S + 100);  // fill the history
    int* stop = nullptr;
    hipHostMalloc(reinterpret_cast<void**>(&stop), sizeof(int), hipHostMallocMapped);
    hipStream_t ks;
    hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) {  // 0: no keeper, 1: keeper that sleeps between polls, 2: keeper that spins
        *stop = 0;
        if (mode > 0) hipLaunchKernelGGL(keeper, dim3(1), dim3(64), 0, ks, stop, mode == 1 ? 1 : 0);
        for (double gap : {0.0, 100.0, 300.0, 1000.0})
            std::printf("keeper %s, %5.0f us of host work between calls: hc_step median %6.2f us\n", mode == 0 ? "off     " : (mode == 1 ? "sleeping" : "spinning"), gap,
                        run(c, N, gap, 600));
        *stop = 1;
        hipStreamSynchronize(ks);
    }
    hc_destroy(c);
    return 0;
}
