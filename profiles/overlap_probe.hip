// overlap_probe.hip -- do two kernels from two HIP streams overlap on this GPU when the long one leaves CUs free?
//   hipcc --offload-arch=gfx950 -O2 profiles/overlap_probe.hip -o /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
__global__ void tiny(volatile int* flag, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *flag = v; }
int main() {
    hipStream_t s_long, s_short, s_masked;
    hipStreamCreateWithFlags(&s_long, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s_short, hipStreamNonBlocking);
    std::vector<uint32_t> mask(8, 0);
    for (int i = 0; i < 192; ++i) mask[i / 32] |= 1u << (i % 32);
    hipError_t e = hipExtStreamCreateWithCUMask(&s_masked, (uint32_t)mask.size(), mask.data());
    std::printf("masked stream: %s\n", hipGetErrorString(e));
    int* flag;
    hipHostMalloc(&flag, sizeof(int), hipHostMallocMapped);
    int* dsink;
    hipMalloc(&dsink, sizeof(int));
    const long long ticks_200us = 200 * 100;  // wall_clock64 runs at 100 MHz
    auto trial = [&](const char* name, hipStream_t sl, int blocks_long, int threads_long, bool with_long) {
        std::vector<double> lat;
        for (int it = 0; it < 60; ++it) {
            *flag = 0;
            if (with_long) hipLaunchKernelGGL(spin, dim3(blocks_long), dim3(threads_long), 0, sl, ticks_200us, dsink);
            const double w0 = now_us();
            while (now_us() - w0 < 40.0) {}
            const double a = now_us();
            hipLaunchKernelGGL(tiny, dim3(24), dim3(256), 0, s_short, flag, it + 1);
            while (*(volatile int*)flag != it + 1) {}
            lat.push_back(now_us() - a);
            hipDeviceSynchronize();
        }
        std::sort(lat.begin(), lat.end());
        std::printf("%-58s short kernel latency median %6.1f us  p90 %6.1f\n", name, lat[lat.size() / 2], lat[lat.size() * 9 / 10]);
    };
    trial("alone", s_long, 0, 0, false);
    trial("long: 128 blocks x 256 threads (half the CUs)", s_long, 128, 256, true);
    trial("long: 256 blocks x 256 threads (every CU, 4 of 32+ wave slots)", s_long, 256, 256, true);
    trial("long: 2048 blocks x 1024 threads (all wave slots)", s_long, 2048, 1024, true);
    trial("long on the CU-masked stream: 2048 x 1024", s_masked, 2048, 1024, true);
    return 0;
}
