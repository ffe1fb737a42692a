// latency_probe.cpp -- host-boundary latency experiments behind the synchronous hc_step design (round 2).
//   hipcc --offload-arch=gfx950 -O2 profiles/r02/latency_probe.cpp -o /tmp/latency_probe && /tmp/latency_probe
// Measures, on an otherwise idle GPU, the wall time from the host's launch call to the moment the host can read the result:
//   A  hipStreamSynchronize after one small kernel that stores into mapped pinned memory
//   B  the same kernel, host spins on a {value, sequence} 16-byte granule in coherent mapped memory
//   C  B with 24 workgroups that first read a 3 KB state vector from mapped pinned memory (zero-copy input)
//   D  C with the state copied by hipMemcpyAsync first (device-resident input)
//   E  C followed by a second, larger launch enqueued before the host starts to spin (work off the critical path)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
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

struct alignas(16) Granule {
    double value;
    unsigned long long seq;
};

__global__ void __launch_bounds__(256) tag_kernel(const double* __restrict__ state, int n_state, Granule* out, unsigned long long seq) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_state; i += blockDim.x) acc += state[i];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        Granule g;
        g.value = red[0] + red[1] + red[2] + red[3] + threadIdx.x;
        g.seq   = seq;
        // one 16-byte store per row: value and sequence number travel in the same write
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u64x2*>(&out[blockIdx.x * 16 + threadIdx.x]) = u64x2{(unsigned long long)__double_as_longlong(g.value), g.seq};
    }
}

__global__ void __launch_bounds__(256) busy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * 1.0000001;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void report(const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    std::printf("%-58s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}

int main() {
    const int iters = 2000, nstate = 768, nwg = 24;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *h_state, *d_state_map, *d_state;
    Granule *h_out, *d_out;
    CK(hipHostMalloc((void**)&h_state, nstate * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_state_map, h_state, 0));
    CK(hipHostMalloc((void**)&h_out, nwg * 16 * sizeof(Granule), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    CK(hipMalloc((void**)&d_state, nstate * sizeof(double)));
    const int nbig = 4 << 20;
    double *d_a, *d_b;
    CK(hipMalloc((void**)&d_a, nbig * sizeof(double)));
    CK(hipMalloc((void**)&d_b, nbig * sizeof(double)));
    CK(hipMemset(d_a, 0, nbig * sizeof(double)));
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

    // warm-up
    for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(tag_kernel, dim3(1), dim3(256), 0, s, d_state_map, 16, d_out, ++seq);
        CK(hipStreamSynchronize(s));
    }

    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(1), dim3(256), 0, s, d_state_map, 16, d_out, ++seq);
        CK(hipStreamSynchronize(s));
        t.push_back(now_us() - a);
    }
    report("A  1 WG, hipStreamSynchronize", t);

    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(1), dim3(256), 0, s, d_state_map, 16, d_out, ++seq);
        wait_tags(1, seq);
        t.push_back(now_us() - a);
    }
    report("B  1 WG, spin on tagged granules", t);
    CK(hipStreamSynchronize(s));

    t.clear();
    std::vector<double> tl;
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        const double b = now_us();
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
        tl.push_back(b - a);
    }
    report("C  24 WGs read 6 KB state zero-copy, spin", t);
    report("   (host time inside the launch call)", tl);
    CK(hipStreamSynchronize(s));

    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        CK(hipMemcpyAsync(d_state, h_state, nstate * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s, d_state, nstate, d_out, ++seq);
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
    }
    report("D  hipMemcpyAsync H2D + 24 WGs, spin", t);
    CK(hipStreamSynchronize(s));

    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        CK(hipStreamSynchronize(s));
        t.push_back(now_us() - a);
    }
    report("C' 24 WGs zero-copy, hipStreamSynchronize", t);

    t.clear();
    std::vector<double> t2;
    for (int i = 0; i < iters; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        hipLaunchKernelGGL(busy_kernel, dim3(nbig / 256 / 4), dim3(256), 0, s, d_a, d_b, nbig / 4);  // 8 MB read + 8 MB write
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
        // tight loop: the next iteration's first kernel queues behind busy_kernel
    }
    report("E  C + a second launch enqueued before the spin (tight loop)", t);
    CK(hipStreamSynchronize(s));

    // F: like E but with idle host time between iterations (the trailing work drains meanwhile)
    t.clear();
    for (int i = 0; i < iters / 4; ++i) {
        const double a = now_us();
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s, d_state_map, nstate, d_out, ++seq);
        hipLaunchKernelGGL(busy_kernel, dim3(nbig / 256 / 4), dim3(256), 0, s, d_a, d_b, nbig / 4);
        wait_tags(nwg, seq);
        t.push_back(now_us() - a);
        const double w = now_us();
        while (now_us() - w < 40.0) __builtin_ia32_pause();
    }
    report("F  E with 40 us of host work between steps", t);
    CK(hipStreamSynchronize(s));
    std::printf("last value %.6f\n", h_out[0].value);
    return 0;
}
