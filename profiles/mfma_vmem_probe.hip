// mfma_vmem_probe.hip -- what a vector-memory load costs the FP64 matrix pipe of an MI355X.  One wave per SIMD runs 12 independent
// v_mfma_f64_16x16x4_f64 per loop iteration (the pipe sustains one per 64 cycles, profiles/mfma_f64_peak.hip) with V = 0 / 2 / 4 / 6
// global_load_dwordx4 spread between them -- the pass kernels have 14 loads per 48 MFMAs at depth 64 and 10 per 24 at depth 32 -- whose
// results are never waited for inside the loop.  Two address patterns: every iteration the same 1 KB per wave (L2 / L1 hits), and a
// stream through a large buffer (HBM, like K).  Printed: cycles per MFMA of a wave; 64.0 = the loads are free.
//   hipcc --offload-arch=gfx950 -O3 profiles/mfma_vmem_probe.hip -o /tmp/mfma_vmem_probe && /tmp/mfma_vmem_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define MF(lo, hi) "v_mfma_f64_16x16x4_f64 a[" #lo ":" #hi "], %[a], %[b], a[" #lo ":" #hi "]\n"
#define LD(lo, hi) "global_load_dwordx4 v[" #lo ":" #hi "], %[p], off\n v_lshl_add_u64 %[p], %[p], 0, %[st]\n"
#define NOLD(lo, hi) ""
#define M0(lo, hi) "v_mfma_f64_16x16x4_f64 a[" #lo ":" #hi "], %[a], %[b], 0\n"

#define KERNEL(name, L0, L1, L2, L3, L4, L5)                                                                                              \
    __global__ void __launch_bounds__(256) name(double* out, const char* buf, unsigned long long stride, int iters, double a0, double b0) { \
        const double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;                                                             \
        const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);                                                                       \
        unsigned long long p = reinterpret_cast<unsigned long long>(buf) + (unsigned long long)wave_global * 1024ull +                     \
                               (threadIdx.x & 63) * 16ull;                                                                                 \
        if (stride == 1) { /* the gather of the pass kernels: 4 ring rows (8.8 KB apart) x 16 overlapping 16-byte windows, 8 bytes apart */ \
            p      = reinterpret_cast<unsigned long long>(buf) + ((unsigned long long)(wave_global * 4 + ((threadIdx.x & 63) >> 4))) * 8800ull + \
                (1000ull - (threadIdx.x & 15)) * 8ull;                                                                                     \
            stride = 0;                                                                                                                    \
        }                                                                                                                                  \
        const long long t0 = clock64();                                                                                                    \
        int r;                                                                                                                             \
        asm volatile(M0(0, 7) M0(8, 15) M0(16, 23) M0(24, 31) M0(32, 39) M0(40, 47) M0(48, 55) M0(56, 63) M0(64, 71) M0(72, 79) M0(80, 87)   \
                         M0(88, 95) "s_mov_b32 s20, %[it]\n"                                                                              \
                     "1:\n" MF(0, 7) MF(8, 15) L0(100, 103) MF(16, 23) MF(24, 31) L1(104, 107) MF(32, 39) MF(40, 47) L2(108, 111)          \
                         MF(48, 55) MF(56, 63) L3(112, 115) MF(64, 71) MF(72, 79) L4(116, 119) MF(80, 87) MF(88, 95) L5(120, 123)          \
                     "s_sub_u32 s20, s20, 1\n"                                                                                             \
                     "s_cmp_lg_u32 s20, 0\n"                                                                                               \
                     "s_cbranch_scc1 1b\n"                                                                                                 \
                     "s_waitcnt vmcnt(0)\n"                                                                                                \
                     "s_nop 15\n"                                                                                                          \
                     "v_accvgpr_read_b32 %[r], a1\n"                                                                                       \
                     : [r] "=&v"(r), [p] "+v"(p)                                                                                           \
                     : [a] "v"(a), [b] "v"(b), [it] "s"(iters), [st] "s"(stride)                                                           \
                     : "s20", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",     \
                       "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "a0", "a1", \
                       "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",      \
                       "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34",     \
                       "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50",     \
                       "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66",     \
                       "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82",     \
                       "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");                         \
        const long long t1 = clock64();                                                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                                                           \
        if (threadIdx.x == 0) out[gridDim.x * 256 + blockIdx.x] = static_cast<double>(t1 - t0);                                            \
    }

KERNEL(mfma12_ld0, NOLD, NOLD, NOLD, NOLD, NOLD, NOLD)
KERNEL(mfma12_ld2, LD, NOLD, NOLD, LD, NOLD, NOLD)
KERNEL(mfma12_ld4, LD, LD, NOLD, LD, LD, NOLD)
KERNEL(mfma12_ld6, LD, LD, LD, LD, LD, LD)


// The same 12 MFMAs, now CONSUMING what the loads bring, the way the pass kernels do: two "gathers" per iteration feed the B operand
// through v_mul_f64 / v_fmac_f64, two "K loads" are the A operands, s_waitcnt vmcnt in front of each group of 6 MFMAs, every load one
// iteration (768 matrix-pipe cycles) ahead of its use.  (Values are meaningless; only the timing counts.)
#define MFAB(lo, hi, A, B) "v_mfma_f64_16x16x4_f64 a[" #lo ":" #hi "], v[" A "], v[" B "], a[" #lo ":" #hi "]\n"
#define LDTO(lo, hi, P) "global_load_dwordx4 v[" #lo ":" #hi "], " P ", off\n"
__global__ void __launch_bounds__(256) mfma12_consume(double* out, const char* buf, unsigned long long stride, int iters, double a0, double b0) {
    const double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long p = reinterpret_cast<unsigned long long>(buf) + (unsigned long long)wave_global * 1024ull + (threadIdx.x & 63) * 16ull;
    unsigned long long g = reinterpret_cast<unsigned long long>(buf) + (1ull << 30) + ((unsigned long long)(wave_global * 4 + ((threadIdx.x & 63) >> 4))) * 8800ull +
                           (1000ull - (threadIdx.x & 15)) * 8ull;
    const long long t0 = clock64();
    int r;
    asm volatile(M0(0, 7) M0(8, 15) M0(16, 23) M0(24, 31) M0(32, 39) M0(40, 47) M0(48, 55) M0(56, 63) M0(64, 71) M0(72, 79) M0(80, 87) M0(88, 95)
                 LDTO(100, 103, "%[g]") LDTO(108, 111, "%[p]") LDTO(104, 107, "%[g]") LDTO(112, 115, "%[p]")
                 "s_mov_b32 s20, %[it]\n"
                 "1:\n"
                 "s_waitcnt vmcnt(2)\n"
                 "v_mul_f64 v[130:131], v[100:101], %[b]\n"
                 "v_fmac_f64 v[130:131], v[102:103], %[a]\n"
                 "s_nop 1\n"
                 MFAB(0, 7, "108:109", "130:131") MFAB(8, 15, "108:109", "130:131") MFAB(16, 23, "108:109", "130:131")
                 LDTO(100, 103, "%[g]")
                 MFAB(24, 31, "110:111", "130:131") MFAB(32, 39, "110:111", "130:131") MFAB(40, 47, "110:111", "130:131")
                 LDTO(108, 111, "%[p]") "v_lshl_add_u64 %[p], %[p], 0, %[st]\n"
                 "s_waitcnt vmcnt(2)\n"
                 "v_mul_f64 v[132:133], v[104:105], %[b]\n"
                 "v_fmac_f64 v[132:133], v[106:107], %[a]\n"
                 "s_nop 1\n"
                 MFAB(48, 55, "112:113", "132:133") MFAB(56, 63, "112:113", "132:133") MFAB(64, 71, "112:113", "132:133")
                 LDTO(104, 107, "%[g]")
                 MFAB(72, 79, "114:115", "132:133") MFAB(80, 87, "114:115", "132:133") MFAB(88, 95, "114:115", "132:133")
                 LDTO(112, 115, "%[p]") "v_lshl_add_u64 %[p], %[p], 0, %[st]\n"
                 "s_sub_u32 s20, s20, 1\n"
                 "s_cmp_lg_u32 s20, 0\n"
                 "s_cbranch_scc1 1b\n"
                 "s_waitcnt vmcnt(0)\n"
                 "s_nop 15\n"
                 "v_accvgpr_read_b32 %[r], a1\n"
                 : [r] "=&v"(r), [p] "+v"(p)
                 : [a] "v"(a), [b] "v"(b), [it] "s"(iters), [st] "s"(stride), [g] "v"(g)
                 : "s20", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
                   "v114", "v115", "v130", "v131", "v132", "v133", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13",
                   "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32",
                   "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51",
                   "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70",
                   "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89",
                   "a90", "a91", "a92", "a93", "a94", "a95");
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) out[gridDim.x * 256 + blockIdx.x] = static_cast<double>(t1 - t0);
}

template <class K>
static void run(K kernel, int nld, const char* buf, unsigned long long stride, int iters, const char* what) {
    const int grid = 256;
    double* d = nullptr;
    (void)hipMalloc(&d, sizeof(double) * (grid * 256 + grid));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, buf, stride, iters / 10, 1.25, 0.75);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, buf, stride, iters, 1.25, 0.75);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfmas = 12.0 * iters, flops = mfmas * 2048.0 * 4 * grid;
    const double bytes = 1024.0 * nld * iters * 4 * grid;
    std::printf("%d loads per 12 MFMAs, %-28s: %.3f ms, FP64 %.3f of 78.6 TFLOP/s, %.2f TB/s loaded; per MFMA %.1f ns (26.7 = 64 cycles at 2.4 GHz)"
                " -> a load costs the pipe %.1f ns\n", nld, what, best, flops / best / 1e9 / 78.6, bytes / best / 1e9,
                best * 1e6 / mfmas, nld ? (best * 1e6 / mfmas - 27.05) * 12.0 / nld : 0.0);
    (void)hipFree(d);
}

int main() {
    const size_t big = (size_t)8 << 30;
    char* buf = nullptr;
    if (hipMalloc(&buf, big) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, big);
    const int iters = 1200;  // stream: 6 loads x 1 KB x 1024 waves x 1200 = 7.5 GB
    const unsigned long long st = 1024ull * 1024ull * 6;  // a load of all 1024 waves covers 1 MB; the next load of a wave is 1 MB on
    run(mfma12_ld0, 0, buf, 0, iters, "");
    run(mfma12_ld2, 2, buf, 0, iters, "same 1 KB per wave (cache)");
    run(mfma12_ld4, 4, buf, 0, iters, "same 1 KB per wave (cache)");
    run(mfma12_ld6, 6, buf, 0, iters, "same 1 KB per wave (cache)");
    run(mfma12_ld2, 2, buf, 1, iters, "gather pattern (cache)");
    run(mfma12_ld4, 4, buf, 1, iters, "gather pattern (cache)");
    run(mfma12_ld6, 6, buf, 1, iters, "gather pattern (cache)");
    run(mfma12_ld2, 2, buf, st / 6 * 1, iters, "stream from HBM");
    run(mfma12_ld4, 4, buf, st / 6 * 1, iters, "stream from HBM");
    run(mfma12_ld6, 6, buf, st / 6 * 1, iters, "stream from HBM");
    run(mfma12_consume, 4, buf, 0, iters, "consumed, K part from cache");
    run(mfma12_consume, 4, buf, st / 6, iters, "consumed, K part streams (2 of the 4)");
    (void)hipFree(buf);
    return 0;
}
