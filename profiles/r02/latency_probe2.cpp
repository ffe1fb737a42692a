// latency_probe2.cpp -- where can the per-step state live so that the step kernel needs one PCIe round trip, not two?
//   hipcc --offload-arch=gfx950 -O2 profiles/r02/latency_probe2.cpp -o profiles/r02/latency_probe2
// All variants: 24 workgroups sum a 768-double state vector and store tagged granules to mapped pinned memory; the host
// spins on the tags.  Run it twice: plainly and with HIP_FORCE_DEV_KERNARG=1.
//   C  state in mapped pinned host memory (zero-copy reads over PCIe)            -- round-2 baseline
//   G  state in fine-grained DEVICE memory that the host writes through the PCIe BAR (if the allocation is host-visible)
//   H  state inside the kernel argument block (3 KB: velocities only; 6 KB: whole state)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                               \
        }                                                                           \
    } while (0)

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
struct alignas(16) Granule { double value; unsigned long long seq; };

__device__ __forceinline__ void finish(double acc, Granule* out, unsigned long long seq) {
    __shared__ double red[4];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16)
        *reinterpret_cast<u64x2*>(&out[blockIdx.x * 16 + threadIdx.x]) =
            u64x2{(unsigned long long)__double_as_longlong(red[0] + red[1] + red[2] + red[3] + threadIdx.x), seq};
}

__global__ void __launch_bounds__(256) ptr_kernel(const double* __restrict__ state, int n, Granule* out, unsigned long long seq) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += state[i];
    finish(acc, out, seq);
}

template <int NV>
struct ArgState { double v[NV]; };
template <int NV>
__global__ void __launch_bounds__(256) arg_kernel(ArgState<NV> s, Granule* out, unsigned long long seq) {
    double acc = 0.0;
    // uniform loop: the argument block is read with scalar loads
    for (int i = 0; i < NV; ++i) acc += s.v[i];
    finish(threadIdx.x == 0 ? acc : 0.0, out, seq);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    std::printf("%-66s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

int main() {
    const int iters = 2000, nstate = 768, nwg = 24;
    std::printf("HIP_FORCE_DEV_KERNARG=%s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *h_state, *d_state_map;
    Granule *h_out, *d_out;
    CK(hipHostMalloc((void**)&h_state, nstate * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_state_map, h_state, 0));
    CK(hipHostMalloc((void**)&h_out, nwg * 16 * sizeof(Granule), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    for (int i = 0; i < nstate; ++i) h_state[i] = 1e-3 * i;
    std::memset(h_out, 0, nwg * 16 * sizeof(Granule));
    unsigned long long seq = 0;
    auto wait_tags = [&](int wgs, unsigned long long want) {
        for (int r = wgs * 16 - 1; r >= 0; --r) {
            volatile unsigned long long* p = &h_out[r].seq;
            while (*p != want) __builtin_ia32_pause();
        }
    };
    std::vector<double> t;
    for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(ptr_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        CK(hipStreamSynchronize(s));
    }
    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(ptr_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
    }
    report("C  24 WGs, state zero-copy from pinned host memory", t);
    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(ptr_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, 0, d_out, ++seq);
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
    }
    report("C0 24 WGs, no state read at all (launch + kernarg + tagged store floor)", t);

    // G: fine-grained device memory written by the host through the BAR
    double* d_fg = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&d_fg, nstate * sizeof(double), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        std::printf("G  hipExtMallocWithFlags(finegrained) failed: %s\n", hipGetErrorString(e));
    } else {
        bool host_ok = false;
        struct sigaction sa, old1, old2;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_segv;
        sigaction(SIGSEGV, &sa, &old1);
        sigaction(SIGBUS, &sa, &old2);
        if (sigsetjmp(jb, 1) == 0) {
            volatile double* p = d_fg;
            p[0] = 1.0;
            host_ok = (p[0] == 1.0);
        }
        sigaction(SIGSEGV, &old1, nullptr);
        sigaction(SIGBUS, &old2, nullptr);
        std::printf("G  fine-grained device memory is %s from the host\n", host_ok ? "writable" : "NOT accessible");
        if (host_ok) {
            t.clear();
            for (int i = 0; i < iters; ++i) {
                const double a = now_us();
                for (int k = 0; k < nstate; ++k) d_fg[k] = 1e-3 * k + i;
                __builtin_ia32_sfence();
                hipLaunchKernelGGL(ptr_kernel, dim3(nwg), dim3(256), 0, s, d_fg, nstate, d_out, ++seq);
                wait_tags(nwg, seq);
                t.push_back(now_us() - a);
            }
            report("G  24 WGs, state in device memory written by the host via BAR", t);
            std::printf("   check: value %.3f (expect %.3f)\n", h_out[0].value, 1e-3 * (767.0 * 768 / 2) + 768.0 * (iters - 1));
        }
    }
    // H: state inside the argument block
    {
        static ArgState<384> a3;
        for (int k = 0; k < 384; ++k) a3.v[k] = 1e-3 * k;
        t.clear();
        bool ok = true;
        for (int i = 0; i < iters && ok; ++i) {
            const double a = now_us();
            hipLaunchKernelGGL((arg_kernel<384>), dim3(nwg), dim3(256), 0, s, a3, d_out, ++seq);
            if (hipGetLastError() != hipSuccess) { ok = false; break; }
            wait_tags(nwg, seq);
            t.push_back(now_us() - a);
        }
        if (ok) report("H  24 WGs, 3 KB of state inside the kernel arguments", t);
        else std::printf("H  3 KB kernel arguments: launch refused\n");
    }
    {
        static ArgState<768> a6;
        for (int k = 0; k < 768; ++k) a6.v[k] = 1e-3 * k;
        t.clear();
        bool ok = true;
        for (int i = 0; i < iters && ok; ++i) {
            const double a = now_us();
            hipLaunchKernelGGL((arg_kernel<768>), dim3(nwg), dim3(256), 0, s, a6, d_out, ++seq);
            if (hipGetLastError() != hipSuccess) { ok = false; break; }
            wait_tags(nwg, seq);
            t.push_back(now_us() - a);
        }
        if (ok) report("H  24 WGs, 6 KB of state inside the kernel arguments", t);
        else std::printf("H  6 KB kernel arguments: launch refused\n");
    }
    CK(hipStreamSynchronize(s));
    return 0;
}
