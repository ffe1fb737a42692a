// latency_probe3.cpp -- a resident "server" kernel instead of a launch per step.
//   hipcc --offload-arch=gfx950 -O2 profiles/r02/latency_probe3.cpp -o profiles/r02/latency_probe3
// 24 workgroups stay resident and poll a sequence word in fine-grained device memory that the host writes through the PCIe
// BAR (after the 768-double state).  On a new sequence number each workgroup sums the state and stores its 16 tagged
// granules to mapped pinned host memory; the host spins on them.  Measures the host round trip, alone and with another
// kernel (8 MB copy) launched on a second stream right after each doorbell.  The server exits after ~1 ms without work.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <xmmintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
struct alignas(16) Granule { double value; u64 seq; };

__global__ void __launch_bounds__(256) server_kernel(const double* state, int n, const u64* door, Granule* out, u64 first_seq, u64* exited, int fence_mode) {
    __shared__ double red[4];
    __shared__ u64 cur;
    u64 expect = first_seq;
    for (;;) {
        if (threadIdx.x == 0) {
            u64 v;
            long long t0 = wall_clock64();
            u64 polls = 0;
            for (;;) {
                ++polls;
                v = __hip_atomic_load(door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v >= expect) break;
                if (fence_mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");        // system-scope acquire: invalidate caches
                if (fence_mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (wall_clock64() - t0 > 100000) {  // 100 MHz clock: 1 ms
                    if (blockIdx.x == 0) {
                        exited[1] = polls;
                        exited[2] = v;
                        exited[3] = (u64)(wall_clock64() - t0);
                    }
                    v = ~0ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            cur = v;
        }
        __syncthreads();
        const u64 v = cur;
        if (v == ~0ull) break;
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += __hip_atomic_load(state + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x < 16)
            *reinterpret_cast<u64x2*>(&out[blockIdx.x * 16 + threadIdx.x]) =
                u64x2{(u64)__double_as_longlong(red[0] + red[1] + red[2] + red[3] + threadIdx.x), v};
        __syncthreads();
        expect = v + 1;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(exited, first_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) busy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * 1.0000001;
}

static hsa_amd_hdp_flush_t g_hdp = {nullptr, nullptr};
static hsa_status_t find_gpu(hsa_agent_t agent, void*) {
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_GPU && !g_hdp.HDP_MEM_FLUSH_CNTL)
        hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &g_hdp);
    return HSA_STATUS_SUCCESS;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    std::printf("%-66s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}

int main() {
    const int iters = 2000, nstate = 768, nwg = 24;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s_srv, s_work;
    CK(hipStreamCreateWithFlags(&s_srv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_work, hipStreamNonBlocking));
    double* fg = nullptr;  // [nstate doubles][door]
    CK(hipExtMallocWithFlags((void**)&fg, (nstate + 8) * sizeof(double), getenv("PROBE_FG") ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    u64* door = reinterpret_cast<u64*>(fg + nstate);
    Granule *h_out, *d_out;
    CK(hipHostMalloc((void**)&h_out, nwg * 16 * sizeof(Granule), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    u64 *h_exit, *d_exit;
    CK(hipHostMalloc((void**)&h_exit, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_exit, h_exit, 0));
    *h_exit = 0;
    std::memset(h_out, 0, nwg * 16 * sizeof(Granule));
    const int nbig = 4 << 20;
    double *d_a, *d_b;
    CK(hipMalloc((void**)&d_a, nbig * sizeof(double)));
    CK(hipMalloc((void**)&d_b, nbig * sizeof(double)));
    CK(hipMemset(d_a, 0, nbig * sizeof(double)));
    *door = 0;
    _mm_sfence();
    bool failed = false;
    auto wait_tags = [&](u64 want) {
        const double t0 = now_us();
        for (int r = nwg * 16 - 1; r >= 0 && !failed; --r) {
            volatile u64* p = &h_out[r].seq;
            while (*p != want) {
                __builtin_ia32_pause();
                if (now_us() - t0 > 2e6) {
                    std::printf("timeout: want %llu, row %d has %llu, door (host view) %llu, exit flag %llu polls %llu last seen %llu ticks %llu\n", want, r, (u64)*p,
                                *reinterpret_cast<volatile u64*>(door), h_exit[0], h_exit[1], h_exit[2], h_exit[3]);
                    failed = true;
                    break;
                }
            }
        }
    };
    u64 seq = 0;
    if (getenv("PROBE_HDP")) {
        hsa_init();
        hsa_iterate_agents(find_gpu, nullptr);
        std::printf("HDP flush registers: mem %p reg %p\n", (void*)g_hdp.HDP_MEM_FLUSH_CNTL, (void*)g_hdp.HDP_REG_FLUSH_CNTL);
    }
    auto ring = [&](int i) {
        for (int k = 0; k < nstate; ++k) fg[k] = 1e-3 * k + i;
        _mm_sfence();
        *reinterpret_cast<volatile u64*>(door) = ++seq;
        _mm_sfence();
        if (g_hdp.HDP_MEM_FLUSH_CNTL) {
            *reinterpret_cast<volatile uint32_t*>(g_hdp.HDP_MEM_FLUSH_CNTL) = 1u;  // make the BAR stores visible to the GPU
            _mm_sfence();
        }
    };
    std::vector<double> t;
    const int fence_mode = getenv("PROBE_FENCE") ? atoi(getenv("PROBE_FENCE")) : 0;
    std::printf("memory: %s, poll fence mode %d\n", getenv("PROBE_FG") ? "fine-grained" : "uncached", fence_mode);
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(server_kernel, dim3(nwg), dim3(256), 0, s_srv, fg, nstate, door, d_out, seq + 1, d_exit, fence_mode);
        CK(hipGetLastError());
        t.clear();
        for (int i = 0; i < iters; ++i) {
            const double a = now_us();
            ring(i);
            if (variant == 1) hipLaunchKernelGGL(busy_kernel, dim3(nbig / 256 / 4), dim3(256), 0, s_work, d_a, d_b, nbig / 4);
            wait_tags(seq);
            if (failed) { hipStreamSynchronize(s_srv); return 1; }
            t.push_back(now_us() - a);
        }
        report(variant == 0 ? "P  resident server, doorbell through the BAR" : "P' the same + a kernel launched on another stream per step", t);
        std::printf("   check: value %.3f (expect %.3f)\n", h_out[0].value, 1e-3 * (767.0 * 768 / 2) + 768.0 * (iters - 1));
        CK(hipStreamSynchronize(s_work));
        CK(hipStreamSynchronize(s_srv));  // the server leaves by itself after 1 ms without a doorbell
        std::printf("   server exited (first_seq written back: %llu)\n", *h_exit);
    }
    return 0;
}
