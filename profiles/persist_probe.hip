// persist_probe.hip -- what a step would cost WITHOUT a dispatch: a resident kernel that polls a mailbox the host writes through the
// PCIe BAR (fine-grained device memory), against the same work as one kernel dispatch per step through HIP.
//   hipcc --offload-arch=gfx950 -O2 profiles/persist_probe.hip -o /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int kWG = 25, kState = 769;  // 64 bodies: 12 N + 1 doubles

// mailbox layout (doubles): [0] sequence number (written LAST by the host), [8 ..] state
__global__ void __launch_bounds__(256) resident(const double* mail, unsigned long long* tagged, unsigned long long first, unsigned long long last) {
    __shared__ double u[768];
    __shared__ int quit;
    if (threadIdx.x == 0) quit = 0;
    __syncthreads();
    for (unsigned long long seq = first; seq <= last; ++seq) {
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(reinterpret_cast<const unsigned long long*>(mail), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (__hip_atomic_load(reinterpret_cast<const unsigned long long*>(mail) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { quit = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (quit) return;
        for (int i = threadIdx.x; i < 768; i += 256) u[i] = __hip_atomic_load(mail + 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (threadIdx.x < 16) {
            double s = 0.0;
            for (int k = 0; k < 48; ++k) s += u[threadIdx.x * 48 + k];
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            // (a RUNNING kernel's stores to host memory stay in L2 until something writes them back: write through, system scope)
            const u64x2 v{(unsigned long long)__double_as_longlong(s), seq};
            unsigned long long* dst = tagged + 2 * (blockIdx.x * 16 + threadIdx.x);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) one_step(const double* mail, unsigned long long* tagged, unsigned long long seq) {
    __shared__ double u[768];
    for (int i = threadIdx.x; i < 768; i += 256) u[i] = mail[8 + i];
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0;
        for (int k = 0; k < 48; ++k) s += u[threadIdx.x * 48 + k];
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u64x2*>(tagged + 2 * (blockIdx.x * 16 + threadIdx.x)) = u64x2{(unsigned long long)__double_as_longlong(s), seq};
    }
}
int main() {
    double* mail = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&mail), 8192 * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) { std::printf("no fine-grained memory\n"); return 1; }
    unsigned long long* tagged = nullptr;
    hipHostMalloc(reinterpret_cast<void**>(&tagged), 2 * 16 * kWG * sizeof(unsigned long long), hipHostMallocMapped);
    std::memset(tagged, 0, 2 * 16 * kWG * sizeof(unsigned long long));
    unsigned long long* dtag = nullptr;
    hipHostGetDevicePointer(reinterpret_cast<void**>(&dtag), tagged, 0);
    std::vector<double> state(kState, 0.25);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    volatile unsigned long long* g = tagged;
    auto wait_all = [&](unsigned long long seq) {
        const double t0 = now_us();
        for (int r = 16 * kWG - 1; r >= 0; --r)
            while (g[2 * r + 1] != seq) {
                _mm_pause();
                if (now_us() - t0 > 2e6) {
                    int got = 0;
                    for (int q = 0; q < 16 * kWG; ++q) got += g[2 * q + 1] == seq;
                    std::printf("timeout waiting for seq %llu: %d of %d granules arrived; mailbox word reads back %llu\n", seq, got, 16 * kWG,
                                (unsigned long long)reinterpret_cast<volatile unsigned long long*>(mail)[0]);
                    reinterpret_cast<volatile unsigned long long*>(mail)[1] = 1;  // tell the resident kernel to leave
                    _mm_sfence();
                    hipDeviceSynchronize();
                    std::printf("kernel left: %s\n", hipGetErrorString(hipGetLastError()));
                    std::exit(2);
                }
            }
    };
    const int n = 3000;
    for (int mode = 0; mode < 2; ++mode) {
        std::memset(mail, 0, 64);
        _mm_sfence();
        std::vector<double> lat;
        const unsigned long long base = 1000ull * (mode + 1);
        if (mode == 0) hipLaunchKernelGGL(resident, dim3(kWG), dim3(256), 0, s, mail, dtag, base + 1, base + n);
        for (int k = 1; k <= n; ++k) {
            const unsigned long long seq = base + k;
            state[5] = 0.001 * k;
            const double a = now_us();
            std::memcpy(mail + 8, state.data(), kState * sizeof(double));
            _mm_sfence();
            if (mode == 0) {
                reinterpret_cast<volatile unsigned long long*>(mail)[0] = seq;
                _mm_sfence();
            } else {
                hipLaunchKernelGGL(one_step, dim3(kWG), dim3(256), 0, s, mail, dtag, seq);
            }
            wait_all(seq);
            lat.push_back(now_us() - a);
        }
        hipStreamSynchronize(s);
        std::sort(lat.begin(), lat.end());
        std::printf("%-44s median %6.2f us  p10 %6.2f  p90 %6.2f\n", mode == 0 ? "resident kernel polling a BAR mailbox:" : "one HIP launch per step:", lat[n / 2], lat[n / 10], lat[n * 9 / 10]);
    }
    return 0;
}
