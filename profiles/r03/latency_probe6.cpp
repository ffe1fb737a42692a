// latency_probe6.cpp -- launch-to-result latency AFTER THE QUEUE HAS BEEN IDLE.  profiles/host_path_c.cpp shows hc_step at 13.6 us
// back to back but 19.1 us when the host works >= 100 us between calls (a Chrono loop does), for every system size, and a resident
// "keeper" wave does not change it (warm_probe.hip).  Here: the same small kernel through a hand-written AQL packet
//   cold     the queue sits empty for `gap` us, then packet + doorbell;
//   armed    a BARRIER-AND packet waiting on an HSA signal and the kernel packet behind it are put into the queue BEFORE the gap (the
//            packet processor is parked on the barrier); after the gap the host writes the kernel arguments and releases the signal;
//   armed2   only the barrier is queued before the gap; the kernel packet + doorbell come after it, then the signal is released.
//   hipcc --offload-arch=gfx950 -O2 --genco --no-gpu-bundle-output -DPROBE_DEVICE_ONLY profiles/r03/latency_probe6.cpp -o /tmp/probe6.co
//   hipcc --offload-arch=gfx950 -O2 profiles/r03/latency_probe6.cpp -o /tmp/latency_probe6 -lhsa-runtime64 && /tmp/latency_probe6 /tmp/probe6.co
#include <hip/hip_runtime.h>

extern "C" __global__ void __launch_bounds__(256) tag6(const double* __restrict__ state, int n_state, unsigned long long* out,
                                                        unsigned long long seq, int nthreads) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_state; i += nthreads) acc += state[i];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const double v = red[0] + red[1] + red[2] + red[3] + threadIdx.x;
        *reinterpret_cast<u64x2*>(out + 2 * (blockIdx.x * 16 + threadIdx.x)) = u64x2{(unsigned long long)__double_as_longlong(v), seq};
    }
}

#ifndef PROBE_DEVICE_ONLY
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>
#include <xmmintrin.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); std::printf("%s failed: %s\n", #x, m_ ? m_ : "?"); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hsa_agent_t g_gpu;
static bool g_have_gpu = false;
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}
struct KernArgs { const double* state; int n_state; int pad0; unsigned long long* out; unsigned long long seq; int nthreads; int pad1; };

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 2) { std::printf("usage: latency_probe6 probe6.co\n"); return 1; }
    CK(hipSetDevice(0));
    const int nwg = 25, n_state = 768;
    double* d_state;
    CK(hipMalloc(&d_state, n_state * sizeof(double)));
    CK(hipMemset(d_state, 0, n_state * sizeof(double)));
    unsigned long long* h_tag;
    CK(hipHostMalloc(&h_tag, nwg * 16 * 16, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h_tag, 0, nwg * 16 * 16);
    unsigned long long* d_tag;
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_tag), h_tag, 0));
    volatile unsigned long long* tag = h_tag;
    auto wait_seq = [&](unsigned long long seq) {
        const double t0 = now_us();
        for (int r = nwg * 16 - 1; r >= 0; --r)
            while (tag[2 * r + 1] != seq) {
                _mm_pause();
                if (now_us() - t0 > 2e6) { std::printf("timeout waiting for sequence %llu\n", seq); std::fflush(stdout); _exit(3); }
            }
    };
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    if (!g_have_gpu) return 1;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (co.empty()) { std::printf("cannot read %s\n", argv[1]); return 1; }
    hsa_code_object_reader_t reader;
    HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
    hsa_profile_t profile;
    HK(hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_PROFILE, &profile));
    hsa_executable_t exe;
    HK(hsa_executable_create_alt(profile, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    HK(hsa_executable_get_symbol_by_name(exe, "tag6.kd", &g_gpu, &sym));
    uint64_t kobj = 0;
    uint32_t group = 0, priv = 0;
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    hsa_queue_t* q = nullptr;
    HK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    char* bar_ka = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&bar_ka), 64 * 256, hipDeviceMallocFinegrained));
    hsa_signal_t gate;
    HK(hsa_signal_create(1, 0, nullptr, &gate));
    const uint32_t mask = q->size - 1;
    auto write_kernarg = [&](uint64_t idx, unsigned long long s) {
        KernArgs ka{d_state, n_state, 0, d_tag, s, 256, 0};
        std::memcpy(bar_ka + (idx & 63) * 256, &ka, sizeof ka);
        _mm_sfence();
    };
    auto put_dispatch = [&](uint64_t idx) {
        hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (idx & mask);
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = 256 * nwg; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = priv;
        p->group_segment_size   = group;
        p->kernel_object        = kobj;
        p->kernarg_address      = bar_ka + (idx & 63) * 256;
        p->reserved2            = 0;
        p->completion_signal.handle = 0;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
    };
    auto put_barrier = [&](uint64_t idx) {
        hsa_barrier_and_packet_t* p = reinterpret_cast<hsa_barrier_and_packet_t*>(q->base_address) + (idx & mask);
        std::memset(reinterpret_cast<char*>(p) + 4, 0, sizeof(*p) - 4);
        p->dep_signal[0] = gate;
        const uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER);
        __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
    };
    auto spin = [&](double us) { const double t0 = now_us(); while (now_us() - t0 < us) {} };
    auto report = [&](const char* name, double gap, std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        std::printf("%-8s gap %6.0f us: launch -> all results on the host  median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, gap, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
    };
    unsigned long long seq = 0;
    const int reps = 800;
    for (double gap : {0.0, 30.0, 100.0, 300.0, 1000.0}) {
        std::vector<double> cold, armed;
        for (int i = 0; i < reps; ++i) {  // cold: empty queue during the gap
            ++seq;
            spin(gap);
            const double a = now_us();
            const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            write_kernarg(idx, seq);
            put_dispatch(idx);
            hsa_signal_store_screlease(q->doorbell_signal, idx);
            wait_seq(seq);
            cold.push_back(now_us() - a);
        }
        for (int i = 0; i < reps; ++i) {  // armed: barrier + dispatch queued before the gap, released by a signal after it
            ++seq;
            hsa_signal_store_relaxed(gate, 1);
            const uint64_t ib = hsa_queue_add_write_index_relaxed(q, 2);
            put_barrier(ib);
            put_dispatch(ib + 1);
            hsa_signal_store_screlease(q->doorbell_signal, ib + 1);
            spin(gap);
            const double a = now_us();
            write_kernarg(ib + 1, seq);
            hsa_signal_store_screlease(gate, 0);
            wait_seq(seq);
            armed.push_back(now_us() - a);
        }
        std::vector<double> armed2;
        for (int i = 0; i < reps; ++i) {  // armed2: ONLY the barrier is queued before the gap; the kernel packet is written after it, then the gate opens
            ++seq;
            hsa_signal_store_relaxed(gate, 1);
            const uint64_t ib = hsa_queue_add_write_index_relaxed(q, 1);
            put_barrier(ib);
            hsa_signal_store_screlease(q->doorbell_signal, ib);
            spin(gap);
            const double a = now_us();
            const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            write_kernarg(idx, seq);
            put_dispatch(idx);
            hsa_signal_store_screlease(q->doorbell_signal, idx);
            hsa_signal_store_screlease(gate, 0);
            wait_seq(seq);
            armed2.push_back(now_us() - a);
        }
        report("cold", gap, cold);
        report("armed", gap, armed);
        report("armed2", gap, armed2);
    }
    hsa_queue_destroy(q);
    return 0;
}
#endif
