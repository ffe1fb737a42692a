// latency_probe4.cpp -- can a short kernel keep its launch-to-result latency while a long, HBM-saturating kernel runs on a
// stream whose CU mask leaves a few CUs free?   hipcc --offload-arch=gfx950 -O2 profiles/r02/latency_probe4.cpp -o profiles/r02/latency_probe4
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef double dvec2 __attribute__((ext_vector_type(2)));
struct alignas(16) Granule { double value; u64 seq; };

__global__ void __launch_bounds__(256) tag_kernel(const double* __restrict__ src, int n, Granule* out, u64 seq) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += src[(size_t)blockIdx.x * n + i];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16)
        *reinterpret_cast<u64x2*>(&out[blockIdx.x * 16 + threadIdx.x]) = u64x2{(u64)__double_as_longlong(red[0] + red[1] + red[2] + red[3]), seq};
}

// long streaming kernel with big, long-lived workgroups (like the look-ahead pass): each workgroup sums its slice
__global__ void __launch_bounds__(256, 1) stream_kernel(const dvec2* __restrict__ src, size_t per_wg, double* __restrict__ out) {
    const dvec2* p = src + (size_t)blockIdx.x * per_wg;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * 4) {
        dvec2 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 256), c = __builtin_nontemporal_load(p + i + 512),
              d = __builtin_nontemporal_load(p + i + 768);
        acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    std::printf("%-70s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nwg = 24, n = 6144;  // 24 workgroups x 48 KB, like the step kernel of C3
    hipStream_t s_main;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    double* d_small;
    CK(hipMalloc((void**)&d_small, (size_t)nwg * n * sizeof(double)));
    CK(hipMemset(d_small, 0, (size_t)nwg * n * sizeof(double)));
    Granule *h_out, *d_out;
    CK(hipHostMalloc((void**)&h_out, nwg * 16 * sizeof(Granule), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    std::memset(h_out, 0, nwg * 16 * sizeof(Granule));
    const size_t big = (size_t)1200 << 20;  // 1.2 GB
    dvec2* d_big;
    double* d_sink;
    CK(hipMalloc((void**)&d_big, big));
    CK(hipMemset(d_big, 0, big));
    CK(hipMalloc((void**)&d_sink, 4096 * sizeof(double)));
    u64 seq = 0;
    auto wait_tags = [&](u64 want) {
        for (int r = nwg * 16 - 1; r >= 0; --r) {
            volatile u64* p = &h_out[r].seq;
            while (*p != want) __builtin_ia32_pause();
        }
    };
    auto measure = [&](const char* name, hipStream_t s_long, int long_wgs) {
        std::vector<double> t, tl;
        for (int rep = 0; rep < 40; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            if (s_long) {
                hipEventRecord(e0, s_long);
                hipLaunchKernelGGL(stream_kernel, dim3(long_wgs), dim3(256), 0, s_long, d_big, big / 16 / long_wgs, d_sink);
                hipEventRecord(e1, s_long);
            }
            const double t0 = now_us();
            while (now_us() - t0 < 150.0) {  // ~10 short launches while the long kernel runs
                const double a = now_us();
                hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, s_main, d_small, n, d_out, ++seq);
                wait_tags(seq);
                t.push_back(now_us() - a);
            }
            if (s_long) {
                hipStreamSynchronize(s_long);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                tl.push_back(ms * 1e3);
            }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        report(name, t);
        if (!tl.empty()) report("      duration of the long kernel (1.2 GB)", tl);
    };
    measure("short kernel alone", nullptr, 0);
    hipStream_t s_all;
    CK(hipStreamCreateWithFlags(&s_all, hipStreamNonBlocking));
    measure("short kernel beside a long kernel on all CUs (256 WGs)", s_all, 256);
    // CU mask: leave the first 4 CUs of each of the 8 XCDs (32 CUs) to the other streams.  Bit i = CU i; on MI300-class parts
    // consecutive bits go round-robin over the XCDs, so bits 0..31 are CUs 0..3 of every XCD.
    for (int reserve : {16, 32}) {
        uint32_t mask[8];
        for (int w = 0; w < 8; ++w) mask[w] = 0xFFFFFFFFu;
        for (int b = 0; b < reserve; ++b) mask[b / 32] &= ~(1u << (b % 32));
        hipStream_t s_mask;
        hipError_t e = hipExtStreamCreateWithCUMask(&s_mask, 8, mask);
        if (e != hipSuccess) {
            std::printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e));
            continue;
        }
        char name[128];
        std::snprintf(name, sizeof name, "short kernel beside a long kernel on a stream masked to %d CUs", 256 - reserve);
        measure(name, s_mask, 256 - reserve);
    }
    return 0;
}
