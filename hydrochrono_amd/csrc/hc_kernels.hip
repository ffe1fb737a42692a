// hc_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hydro-force path.
//
// Per step:
//   plain step      conv_step_kernel (all of K, streamed FP64 GEMV)                    -> finalize_kernel
//   look-ahead      once per block of 32 / 16 steps: conv_block_kernel (K read ONCE for the block, FP64 MFMA GEMM over what the
//                   known history contributes to every step of the block) -> reduce_block_kernel;
//                   every step: finalize_kernel alone (adds the step's own newest-sample part and the scatter results of the
//                   earlier block steps), then scatter_kernel (what this step's sample contributes to the later ones)
// conv_step_kernel  partial[chunk][row] = K[row, chunk] . u[chunk]; u[s][col] -- the body velocity history interpolated
//                   at t - tau_s, times the trapezoid width -- is formed in registers from the velocity ring via a
//                   per-workgroup bracket table in LDS; the irregular-wave excitation Kex . eta(t - tau_j) rides in the
//                   same launch as extra column chunks.  HBM-bound: K is read exactly once, 16 B per lane, coalesced.
// finalize_kernel   the step kernel: fixed-order reductions, own-sample part, hydrostatics, regular-wave / spectral term,
//                   total = hydrostatic - radiation + waves, velocity-ring push of this step's sample, tagged host results.
// Reference semantics: src/hydro_forces.cpp:263-322,537-691,727-767; src/wave_types.cpp:315-327,776-844.
#include "hc_kernels.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace hc {

static constexpr int kConvThreads = 256;  // 4 waves of 64
static constexpr int kWave        = 64;

typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef double dvec2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte load at 8-byte alignment
typedef double dvec4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t panel_offset(int ngp, int row, int f) {
    const int rt = row >> 4, ri = row & 15, gp = f >> 3, j = (f >> 2) & 1, kk = f & 3;
    return ((static_cast<size_t>(rt) * ngp + gp) * 64 + (kk * 16 + ri)) * 2 + j;
}

// ------------------------------------------------------------------------------------------------
// Ingest re-layout: BEMIO file order [i][col][s] (s fastest) -> panel order, rho folded in
// (HydroData::GetRIRFVal, src/h5fileinfo.cpp:321-323).  Init-time; one thread per element of the body's 6 rows.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) relayout_rirf_kernel(const double* __restrict__ Kb, double* __restrict__ K, int ngp, int D, int S,
                                                             int row0, double scale) {
    const size_t F   = (size_t)S * D;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 6 * F) return;
    const int i   = (int)(gid / F);
    const int f   = (int)(gid % F);
    const int s   = f / D, col = f - s * D;
    K[panel_offset(ngp, row0 + i, f)] = Kb[((size_t)i * D + col) * S + s] * scale;
}

void launch_relayout_rirf(const double* d_Kb, double* d_K, int ngp, int D, int S, int row0, double scale, hipStream_t stream) {
    const size_t n = (size_t)6 * S * D;
    hipLaunchKernelGGL(relayout_rirf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_Kb, d_K, ngp, D, S, row0, scale);
}

__global__ void __launch_bounds__(256) relayout_rowmajor_kernel(const double* __restrict__ src, int rows, int cols, double* __restrict__ dst,
                                                                 int ngp, int row0) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)rows * cols) return;
    const int r = (int)(gid / cols), c = (int)(gid % cols);
    dst[panel_offset(ngp, row0 + r, c)] = src[gid];
}

void launch_relayout_rowmajor(const double* d_src, int rows, int cols, double* d_panel, int ngp, int row0, hipStream_t stream) {
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(relayout_rowmajor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_src, rows, cols, d_panel, ngp, row0);
}

__global__ void __launch_bounds__(256) unrelayout_kernel(Panel K, int Dloc, int D, int S, double* __restrict__ out) {
    const size_t n      = (size_t)Dloc * D * S;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gid < n; gid += stride) {
        const int s   = (int)(gid % S);
        const int col = (int)((gid / S) % D);
        const int row = (int)(gid / ((size_t)S * D));
        out[gid]      = K.base[panel_offset(K.ngp, row, s * D + col)];
    }
}

void launch_unrelayout(const Panel& K, int Dloc, int D, int S, double* d_out, hipStream_t stream) {
    const size_t n      = (size_t)Dloc * D * S;
    const size_t blocks = std::min<size_t>((n + 255) / 256, (size_t)1 << 22);
    hipLaunchKernelGGL(unrelayout_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, K, Dloc, D, S, d_out);
}

// one (row, col) series along s (diagnostics CSV of the TaperedDirect preprocessing)
__global__ void __launch_bounds__(256) extract_series_kernel(Panel K, int row, int col, int D, int S, double* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) out[s] = K.base[panel_offset(K.ngp, row, s * D + col)];
}

void launch_extract_series(const Panel& K, int row, int col, int D, int S, double* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(extract_series_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, K, row, col, D, S, d_out);
}

// ------------------------------------------------------------------------------------------------
// History access and the bracket search shared by both convolution kernels.
// ------------------------------------------------------------------------------------------------
// The argument block of a kernel on the step's critical path, requested whole at entry.  The compiler loads arguments where it needs
// them, a cache line's worth at a time with a wait each -- the step kernel ran through five such scalar-cache misses one after the other
// before its first K word was requested, and an argument block of the direct path lies in uncached device memory (hc_direct.hpp).  One
// word of every 64-byte line, all in flight at once, and ONE wait (the empty asm statement is their only use: it needs them all in
// registers at the same time); the argument loads that follow hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void touch_args() {
#if defined(__HIP_DEVICE_COMPILE__)
    const int* kp = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int n = (BYTES + 63) / 64;
    static_assert(n <= 48, "touch_args: argument block too large");
    constexpr int groups = (n + 15) / 16;
    int w[16 * groups];
#pragma unroll
    for (int i = 0; i < 16 * groups; ++i) w[i] = kp[16 * (i < n ? i : 0)];
    __builtin_amdgcn_sched_barrier(0);  // (whatever the caller requested before stays in front of the wait)
#pragma unroll
    for (int g = 0; g < groups; ++g) {
        const int* v = w + 16 * g;
        asm volatile("" ::"s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]), "s"(v[8]), "s"(v[9]), "s"(v[10]), "s"(v[11]),
                     "s"(v[12]), "s"(v[13]), "s"(v[14]), "s"(v[15]));
    }
#endif
}

__device__ __forceinline__ double state_velocity(const double* __restrict__ state, int N, int col) {
    const int b = col / 6, d = col - 6 * b;
    return d < 3 ? state[6 * N + 3 * b + d] : state[9 * N + 3 * b + (d - 3)];
}

__device__ __forceinline__ double hist_time(const HistoryView& h, int k) {
    return k == 0 ? h.t : h.ring_t[(h.head - k + h.Hcap) % h.Hcap];
}


// InterpolateVelocity6D weights (src/hydro_forces.cpp:343-371) of query time q inside the bracket (sample lo = newer, lo + 1 = older).
__device__ __forceinline__ Bracket bracket_weights(const HistoryView& h, double q, int lo, double newer, double older, int* error_flag) {
    Bracket b;
    b.wo = 0.0; b.wn = 0.0; b.off_older = 0; b.off_newer = 0;
    if (q == older) { b.wo = 1.0; b.wn = 0.0; }
    else if (q == newer) { b.wo = 0.0; b.wn = 1.0; }
    else if (q > older && q < newer) {
        const double td = newer - older;
        b.wo = (td != 0.0) ? ((newer - q) / td) : 0.0;
        b.wn = 1.0 - b.wo;
    } else {
        *error_flag = 1;  // "query_time not bracketed by history" (:370)
        return b;
    }
    b.off_older = ((h.head - (lo + 1) + h.Hcap) % h.Hcap) * h.D;
    b.off_newer = (lo == 0) ? -1 : ((h.head - lo + h.Hcap) % h.Hcap) * h.D;
    return b;
}

// AdvanceToBracket + InterpolateVelocity6D weights (src/hydro_forces.cpp:343-381) for a query time q <= h.t against
// the history whose newest sample (k = 0) is the current state at h.t.  Finds the smallest i in [0, H-2] with
// time(i+1) <= q; i == H-1 means "no older sample" and the IRF step contributes nothing (:604-606).
__device__ Bracket find_bracket(const HistoryView& h, double q, int* error_flag) {
    int lo = 0, hi = h.H - 1;
    double newer = 0.0, older = 0.0;
    bool have = false;  // the two sample times of the bracket are already in registers (one round trip less per table entry)
    if (h.H >= 3) {
        // histories are close to uniformly spaced: try the index the last step size predicts, then fall back
        int gi = (int)((h.t - q) / h.dt_hint) - 1;
        gi     = max(0, min(gi, h.H - 2));
#pragma unroll 1
        for (int k = 0; k < 3; ++k, ++gi) {
            if (gi > h.H - 2) break;
            const double t_o = hist_time(h, gi + 1), t_n = hist_time(h, gi);
            if (t_o <= q && (gi == 0 || t_n > q)) { lo = hi = gi; newer = t_n; older = t_o; have = true; break; }
        }
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (hist_time(h, mid + 1) <= q) hi = mid; else lo = mid + 1;
    }
    if (lo >= h.H - 1) {
        Bracket b;
        b.wo = 0.0; b.wn = 0.0; b.off_older = 0; b.off_newer = 0;
        return b;
    }
    if (!have) { newer = hist_time(h, lo); older = hist_time(h, lo + 1); }
    return bracket_weights(h, q, lo, newer, older, error_flag);
}

// interpolated velocity of column `col` for bracket b (exact copies where the reference returns early).  Branch-free:
// both ring loads are issued unconditionally (slots of masked brackets are 0, a valid row), so a caller that stages
// several values gets all its loads in flight at once instead of one dependent round trip per branch.
// vstate = velocity of `col` in the current state (used when the newer sample is the current one).
__device__ __forceinline__ double interp_velocity(const HistoryView& h, const Bracket& b, int col, double vstate) {
    const double vo  = h.ring_v[b.off_older + col];
    const double vnr = h.ring_v[max(b.off_newer, 0) + col];
    const double vn  = (b.off_newer >= 0) ? vnr : vstate;
    double v = b.wo * vo + b.wn * vn;
    v = (b.wo == 0.0) ? vn : v;
    v = (b.wn == 0.0) ? vo : v;
    v = (b.wo == 0.0 && b.wn == 0.0) ? 0.0 : v;
    return v;
}

// Same value for finite history data without the exact-copy selects: a weight of exactly 0 or 1 already reproduces the
// copied sample (1*v + 0*w == v), and a masked bracket (both weights 0) gives 0.  Used by the look-ahead pass, whose
// staging loop is instruction-bound.
__device__ __forceinline__ double interp_velocity_lean(const HistoryView& h, const Bracket& b, int col, double vstate) {
    const double vo  = h.ring_v[b.off_older + col];
    const double vnr = h.ring_v[max(b.off_newer, 0) + col];
    const double vn  = (b.off_newer >= 0) ? vnr : vstate;
    return b.wo * vo + b.wn * vn;
}

// eta(q) by linear interpolation in the free-surface table (src/wave_types.cpp:797-831), in pieces so that a caller can
// put the loads of several query times in flight together.
__device__ __forceinline__ int eta_guess(const EtaTable& a, double q) {
    const int idx = (int)floor((q - a.eta_t0) / a.eta_dt);  // eta_t0 = eta_t[0], passed by value: no load before the index is known
    return max(0, min(idx, a.nt - 2));
}
// does interval idx (times t1, t2) end the reference's search for q?
__device__ __forceinline__ bool eta_guess_ok(const EtaTable& a, int idx, double q, double t1, double t2) {
    return (idx == 0 || t1 <= q) && (idx == a.nt - 2 || t2 > q);
}
__device__ __forceinline__ double eta_interp(double q, double t1, double t2, double e1, double e2, int* error_flag) {
    if (q == t1) return e1;
    if (q == t2) return e2;
    if (q > t1 && q < t2) {
        const double w1 = (t2 - q) / (t2 - t1);
        const double w2 = 1.0 - w1;
        return w1 * e1 + w2 * e2;
    }
    *error_flag = 2;  // outside the table: the host has already refused the step (src/wave_types.cpp:833-840)
    return 0.0;
}
// the search itself, from a starting index (rarely needed: the table is close to uniform)
__device__ __forceinline__ double eta_search(const EtaTable& a, int idx, double q, int* error_flag) {
    while (idx > 0 && a.eta_t[idx] > q) --idx;
    while (idx < a.nt - 2 && a.eta_t[idx + 1] <= q) ++idx;
    return eta_interp(q, a.eta_t[idx], a.eta_t[idx + 1], a.eta[idx], a.eta[idx + 1], error_flag);
}

// e[j] = eta(t - ex_tau[j]) * ex_width[j]: times and values of the guessed interval are requested together (one round trip)
__device__ __forceinline__ double eta_at(const EtaTable& a, double t, int j, int* error_flag) {
    if (j >= a.L) return 0.0;
    const double q   = t - a.ex_tau[j];
    const int idx    = eta_guess(a, q);
    const double t1 = a.eta_t[idx], t2 = a.eta_t[idx + 1], e1 = a.eta[idx], e2 = a.eta[idx + 1];
    const double val = eta_guess_ok(a, idx, q, t1, t2) ? eta_interp(q, t1, t2, e1, e2, error_flag) : eta_search(a, idx, q, error_flag);
    return val * a.ex_width[j];
}

// ------------------------------------------------------------------------------------------------
// conv_step_kernel<MT,U>: streamed FP64 GEMV over panel-layout K.  HBM-bound, 0.25 flop/B.
//   grid = ngroups * (nchunks_rad + nchunks_ex) workgroups of 4 waves; a workgroup owns MT row tiles (16 rows each)
//   x one chunk of column groups; wave w takes column groups gp0+w, gp0+w+4, ...  Per column group a lane issues MT
//   non-temporal global_load_dwordx4 (1 KiB coalesced per wave and tile, straight to VGPRs -- K is used once), forms the
//   right-hand side for its two columns and keeps MT FP64 accumulators.  Reduction: two xor-shuffles over the 4 column
//   lanes of a row, then a 4-entry LDS step over the waves -- a FIXED order, bitwise reproducible.
//   The right-hand side of the chunk is staged in LDS first (a few KB): radiation -- one thread per IRF sample the chunk
//   touches finds the history bracket and weights, then every thread forms u = interp(v) * width_s for a few columns
//   from two ring loads (L2 hits); excitation -- eta(t - tau_j), linearly interpolated in the precomputed table
//   (src/wave_types.cpp:797-831), times width_j.  The streaming loop then only reads K and two LDS words per 16 bytes.
// ------------------------------------------------------------------------------------------------
template <int MT, int U>
__global__ void __launch_bounds__(kConvThreads) conv_step_kernel(StepArgs a) {
    // dynamic LDS: right-hand side of the chunk [chunk columns], then the bracket table [samples] + widths
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* rhs  = reinterpret_cast<double*>(smem_raw);
    Bracket* tab = reinterpret_cast<Bracket*>(rhs + a.rhs_capacity);
    double* wtab = reinterpret_cast<double*>(tab + a.max_steps_per_chunk);
    __shared__ double red[kConvThreads / kWave][MT][16];
    touch_args<sizeof(StepArgs)>();

    // block index -> (chunk, row group): the row groups of one chunk stage the same right-hand side, so they get block
    // indices congruent mod 8 (same XCD under round-robin placement) and close together: [octet of chunks][group][chunk % 8]
    const int nct   = a.nchunks_rad + a.nchunks_ex;
    const int r8    = (int)blockIdx.x % (8 * a.ngroups);
    const int chunk = ((int)blockIdx.x / (8 * a.ngroups)) * 8 + (r8 & 7);
    const int grp   = r8 >> 3;
    if (chunk >= nct) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 4;

    const bool radiation = chunk < a.nchunks_rad;
    const Panel& M = radiation ? a.K : a.Kex;
    // column groups [gp0, gp1) of this chunk and, for radiation, its valid column range [c0, c1)
    int gp0, gp1, c0 = 0, c1 = 0;
    if (radiation) {
        gp0 = chunk * a.chunk_gp;
        gp1 = min((a.F_limit + 7) >> 3, gp0 + a.chunk_gp);
        c0  = gp0 * 8;
        c1  = min(a.F_limit, gp1 * 8);
    } else {
        gp0 = (chunk - a.nchunks_rad) * a.chunk_gp_ex;
        gp1 = min(a.Kex.ngp, gp0 + a.chunk_gp_ex);
    }
    const double* __restrict__ kbase = M.base + ((size_t)(grp * MT) * M.ngp) * 128 + lane * 2;
    const size_t tile_stride = (size_t)M.ngp * 128;

    if (radiation) {
        // ---- stage u[f] = interp(v_col)(t - tau_s) * width_s for the chunk's columns ----
        const int D = a.hist.D;
        const int s0 = c0 / D;
        const int ns = (c1 - 1) / D - s0 + 1;
        for (int k = tid; k < ns; k += kConvThreads) {
            tab[k]  = find_bracket(a.hist, a.hist.t - a.tau[s0 + k], a.error_flag);
            wtab[k] = a.width[s0 + k];
        }
        __syncthreads();
        for (int f = gp0 * 8 + tid; f < gp1 * 8; f += kConvThreads) {
            double u = 0.0;
            if (f >= c0 && f < c1) {
                const int s = f / D, col = f - s * D;
                u = interp_velocity(a.hist, tab[s - s0], col, state_velocity(a.hist.state, a.hist.N, col)) * wtab[s - s0];
            }
            rhs[f - gp0 * 8] = u;
        }
    } else {
        // ---- stage e[j] = eta(t - tau_j) * width_j ----
        for (int j = gp0 * 8 + tid; j < gp1 * 8; j += kConvThreads) rhs[j - gp0 * 8] = eta_at(a.ex, a.hist.t, j, a.error_flag);
    }
    __syncthreads();

    double acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.0;
#pragma unroll U
    for (int gp = gp0 + wave; gp < gp1; gp += 4) {
        dvec2 kv[MT];
        // a matrix streamed exactly once per step uses non-temporal loads (keeps the ring in L2)
#pragma unroll
        for (int m = 0; m < MT; ++m)
            kv[m] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kbase + (size_t)m * tile_stride + (size_t)gp * 128));
        const double u0 = rhs[(gp - gp0) * 8 + kk], u1 = rhs[(gp - gp0) * 8 + 4 + kk];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = fma(kv[m].x, u0, acc[m]);
            acc[m] = fma(kv[m].y, u1, acc[m]);
        }
    }

#pragma unroll
    for (int m = 0; m < MT; ++m) {
        double v = acc[m];
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        if (lane < 16) red[wave][m][lane] = v;
    }
    __syncthreads();
    if (tid < MT * 16) {
        const int m = tid >> 4, r = tid & 15;
        a.partials[(size_t)chunk * a.Dpad + (grp * MT + m) * 16 + r] = ((red[0][m][r] + red[1][m][r]) + red[2][m][r]) + red[3][m][r];
    }
}

template <int MT>
static void launch_conv_step_mt(const StepArgs& a, int unroll, int nblocks, size_t smem, hipStream_t stream) {
#ifdef HC_TUNING  // (unroll factors 1 and 3: sweeps only)
    if (unroll == 1) { hipLaunchKernelGGL((conv_step_kernel<MT, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, a); return; }
    if (unroll == 3) { hipLaunchKernelGGL((conv_step_kernel<MT, 3>), dim3(nblocks), dim3(kConvThreads), smem, stream, a); return; }
#endif
    (void)unroll;
    hipLaunchKernelGGL((conv_step_kernel<MT, 2>), dim3(nblocks), dim3(kConvThreads), smem, stream, a);
}

StepLaunch step_launch_config(const StepArgs& a, int mt) {
    StepLaunch l;
    l.nblocks = ((a.nchunks_rad + a.nchunks_ex + 7) >> 3) * 8 * a.ngroups;  // octets of chunks (kernel's block mapping)
#ifdef HC_TUNING
    static const int unroll = [] {
        const char* e = std::getenv("HC_CONV_UNROLL");
        return e ? std::atoi(e) : 2;
    }();
#else
    constexpr int unroll = 2;
#endif
    l.smem = (size_t)a.rhs_capacity * sizeof(double) + (size_t)max(1, a.max_steps_per_chunk) * (sizeof(Bracket) + sizeof(double));
    l.MT   = (mt == 4 || mt == 2) ? mt : 1;
    l.U    = (unroll == 1 || unroll == 3) ? unroll : 2;
    return l;
}

void launch_conv_step(const StepArgs& a, int mt, hipStream_t stream) {
    const StepLaunch l = step_launch_config(a, mt);
    if (l.nblocks <= 0) return;
    if (l.MT == 4) launch_conv_step_mt<4>(a, l.U, l.nblocks, l.smem, stream);
    else if (l.MT == 2) launch_conv_step_mt<2>(a, l.U, l.nblocks, l.smem, stream);
    else launch_conv_step_mt<1>(a, l.U, l.nblocks, l.smem, stream);
}

// ------------------------------------------------------------------------------------------------
// conv_block_kernel<MT, R, NB>: the look-ahead pass.  C[row, j] = sum_f K[row, f] * U[f, j], j = 0 .. 16*NB - 1, with
//   U[(s,col), j] = interp(v_col)(tpred[j] - tau_s) * width_s   for s >= s_cut[j], else 0
// on v_mfma_f64_16x16x4_f64: the 16-byte word a lane streams from a K panel is the A operand of two MFMAs per block of 16
// steps, so K leaves HBM ONCE for 16 (NB = 1) or 32 (NB = 2) steps.  Same grid / chunk mapping as conv_step_kernel;
// partials [chunk][j][row], reduced in fixed order by reduce_block_kernel.
//
// Radiation work item = a rolling software pipeline.  The wave keeps R column groups ("fragments": MT 16-byte K words +
// the ring values of its B operands) in flight at all times: in every step of the loop it waits for the oldest fragment
// only (a counted vmcnt), forms the B operands in registers
//     u = wo' * ring[off_older + col] + wn' * ring[off_newer + col]        (wo', wn' = weights x trapezoid width),
// issues 2*MT*NB MFMAs and immediately re-fills the freed registers with the fragment R groups ahead -- so the wave has
// loads outstanding while its MFMAs run, and nothing of the right-hand side goes through LDS inside the loop.  Each lane
// gathers exactly its own B operands (column 8gp + kk (+4), step j = 16*tb + (lane & 15)); the bracket of (IRF sample, step)
// comes from a table in LDS built once per workgroup.  The not-yet-known sample of the pass is zero (hc_step.cpp:
// launch_pass), so a bracket whose newer end is that sample simply gets wn' = 0.
//   UNI: D % 8 == 0 -- a column group never straddles two IRF samples, the sample index is wave-uniform and the bracket
//        registers are reloaded only when it changes.
// Excitation work items (Kex x eta(tpred[j] - tau_l), 0.3 % of the bytes) keep the LDS-staged form of round 1.
// ------------------------------------------------------------------------------------------------
static constexpr int kWaveGp    = 4;                  // column groups a wave handles per sub-tile of an excitation item (32 columns)
static constexpr int kUStride   = kWaveGp * 8 + 2;    // LDS row stride of a wave's U[j][col] in doubles: conflict-free ds_read_b64 of the B operand
static constexpr int kUWave     = 16 * kUStride;      // doubles per wave

// Excitation chunk `e` of Kex for the 16 predicted times tpred[j0 .. j0+15].
template <int MT>
__device__ __forceinline__ void block_exc_work(const BlockArgs& a, const int grp, const int e, const int j0, double* Uall) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 4, jstep = lane & 15;
    const Panel& M = a.Kex;
    const int gp0 = e * a.chunk_gp_ex;
    const int gp1 = min(a.Kex.ngp, gp0 + a.chunk_gp_ex);

    dvec4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = dvec4{0.0, 0.0, 0.0, 0.0};

    const double* __restrict__ kbase = M.base + ((size_t)(grp * MT) * M.ngp) * 128 + lane * 2;
    const size_t tile_stride = (size_t)M.ngp * 128;
    double* Us = Uall + wave * kUWave;
    // staging role of this lane: column c8 of column group `sit` of the wave's sub-tile, steps jh, jh+2, ..., jh+14
    const int c8 = lane & 7, sit = (lane >> 3) & 3, jh = lane >> 5;

    for (int sub0 = gp0; sub0 < gp1; sub0 += 4 * kWaveGp) {
        // 1. put this wave's K fragments of the sub-tile in flight
        dvec2 kv[kWaveGp][MT];
#pragma unroll
        for (int it = 0; it < kWaveGp; ++it) {
            const int gp = sub0 + wave + 4 * it;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (gp < gp1)
                    kv[it][m] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kbase + (size_t)m * tile_stride + (size_t)gp * 128));
                else
                    kv[it][m] = dvec2{0.0, 0.0};
            }
        }
        // 2. stage U[j][l] = eta(tpred[j0 + j] - ex_tau[l]) * ex_width[l]  (0 for l >= L) for the wave's 32 columns x 16 steps
        //    (same arithmetic as eta_at; the table loads of 4 steps go out together)
        {
            const int f = (sub0 + wave + 4 * sit) * 8 + c8;
            constexpr int NQ = 4;
            const bool in   = f < a.ex.L;
            const int fl    = in ? f : 0;
            const double tf = a.ex.ex_tau[fl], wf = a.ex.ex_width[fl];
#pragma unroll 1
            for (int q0 = 0; q0 < 8; q0 += NQ) {
                double qv[NQ], t1[NQ], t2[NQ], e1[NQ], e2[NQ];
                int ix[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    qv[q] = a.tpred[j0 + jh + 2 * (q0 + q)] - tf;
                    ix[q] = eta_guess(a.ex, qv[q]);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    t1[q] = a.ex.eta_t[ix[q]];
                    t2[q] = a.ex.eta_t[ix[q] + 1];
                    e1[q] = a.ex.eta[ix[q]];
                    e2[q] = a.ex.eta[ix[q] + 1];
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    double val = 0.0;
                    if (in) val = eta_guess_ok(a.ex, ix[q], qv[q], t1[q], t2[q]) ? eta_interp(qv[q], t1[q], t2[q], e1[q], e2[q], a.error_flag)
                                                                                  : eta_search(a.ex, ix[q], qv[q], a.error_flag);
                    Us[(jh + 2 * (q0 + q)) * kUStride + sit * 8 + c8] = val * wf;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private LDS tile: no barrier needed
        // 3. 2 MFMAs per streamed 16-byte word and row tile; consecutive MFMAs use different accumulators
#pragma unroll
        for (int it = 0; it < kWaveGp; ++it) {
            const double u0 = Us[jstep * kUStride + it * 8 + kk];
            const double u1 = Us[jstep * kUStride + it * 8 + 4 + kk];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[it][m].x, u0, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[it][m].y, u1, acc[m], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // U reads done before the next sub-tile overwrites it
    }
    __syncthreads();
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg.  red[wave][m][row*16 + j] aliases U.
    double* red = Uall;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((size_t)wave * MT + m) * 256 + (kk + 4 * r) * 16 + jstep] = acc[m][r];
    __syncthreads();
    const int out_chunk = a.nchunks + e;
    for (int idx = tid; idx < MT * 256; idx += kConvThreads) {
        const int m = idx >> 8, el = idx & 255, row = el >> 4, j = el & 15;
        const double v = ((red[(0 * MT + m) * 256 + el] + red[(1 * MT + m) * 256 + el]) + red[(2 * MT + m) * 256 + el]) + red[(3 * MT + m) * 256 + el];
        a.partials[((size_t)out_chunk * a.depth + j0 + j) * a.Dpad + (grp * MT + m) * 16 + row] = v;
    }
}

// Bracket table of a look-ahead work item: entry k*L + j = bracket of (IRF sample s0 + k, block step j), weights already times the
// trapezoid width.  Every workgroup of the launch builds its table at the same moment, with HBM idle (4 us at C3: kernel arguments ->
// tau / width / plan rows -> sample times -> LDS, three dependent round trips), so a thread keeps its up to three entries in flight
// together -- tau / width first, then the two sample times the step-size hint points at; an entry whose hint misses (irregular
// history) falls back to the search of find_bracket, which gives the same bracket.
// BYTES: the slot of the older sample as a byte offset inside a row of the per-DoF ring (its newer neighbour follows it), no t_on.
// The table of a SHORT pass (two-level form, BlockArgs::mini_kw > 0): the same entries, but only brackets that touch the samples of
// the sub-block that has just ended count, and their times are the plan's predicted grid times handed over in the argument block
// (mini_bracket, hc_limits.hpp -- the host-side planner test runs the same function).  No ring_t reads at all.
template <int L, bool BYTES>
__device__ __forceinline__ void build_mini_table(const BlockArgs& a, const int chunk, const int s0, const int ns, const int s_live, double* t_wo,
                                                 double* t_wn, int* t_oo, int* t_on) {
    const HistoryView& h = a.hist;
    const int n  = ns * L;
    const int jb = a.mini_narrow ? (int)a.mini_jbase[chunk] : 0;  // narrow form: column jj of this chunk is step jb + jj
    for (int idx = threadIdx.x; idx < n; idx += kConvThreads) {
        const int k = idx / L, j = jb + (idx - k * L), s = s0 + k;
        double wo = 0.0, wn = 0.0;
        int lo = 0;
        if (j < a.mini_steps && s >= a.s_cut[j] && s != a.s_defer[j] && s < s_live) {
            if (!mini_bracket(a.mini_time, a.mini_kw, a.tpred[j] - a.tau[s], &wo, &wn, &lo)) *a.error_flag = 1;
            const double w = a.width[s];
            wo *= w;
            wn *= w;
        }
        // ring slots: history index k of the view lives in slot head - k (index 0 = the not-yet-known sample, weight 0)
        const int slot_o = (h.head - (lo + 1) + 2 * h.Hcap) % h.Hcap, slot_n = (h.head - lo + 2 * h.Hcap) % h.Hcap;
        const bool any   = wo != 0.0 || wn != 0.0;
        t_wo[idx] = wo;
        t_wn[idx] = wn;
        if constexpr (BYTES) {
            t_oo[idx] = any ? slot_o * 8 : 0;
        } else {
            t_oo[idx] = any ? slot_o : 0;
            t_on[idx] = (any && lo >= 1) ? slot_n : 0;
        }
    }
}

template <int L, bool BYTES>
__device__ __forceinline__ void build_block_table(const BlockArgs& a, const int chunk, const int s0, const int ns, const int s_live, double* t_wo,
                                                  double* t_wn, int* t_oo, int* t_on) {
    if (a.mini_kw > 0) {
        build_mini_table<L, BYTES>(a, chunk, s0, ns, s_live, t_wo, t_wn, t_oo, t_on);
        return;
    }
    constexpr int NE = 3;
    const HistoryView& h = a.hist;
    const int n = ns * L, D = h.D;
    for (int base = threadIdx.x; base < n; base += NE * kConvThreads) {
        int gi[NE];
        bool valid[NE], act[NE];
        double q[NE], w[NE], t_o[NE], t_n[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = base + e * kConvThreads;
            valid[e]      = idx < n;
            const int k = valid[e] ? idx / L : 0, j = valid[e] ? idx - k * L : 0, s = s0 + k;
            act[e] = valid[e] && s >= a.s_cut[j] && s != a.s_defer[j] && s < s_live;
            q[e]   = a.tpred[j] - a.tau[act[e] ? s : 0];
            w[e]   = a.width[s < s_live ? s : 0];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            int g = (int)((h.t - q[e]) / h.dt_hint) - 1;
            g     = max(0, min(g, h.H - 2));
            gi[e] = g;
            t_o[e] = hist_time(h, g + 1);
            t_n[e] = hist_time(h, g);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (!valid[e]) continue;
            Bracket b;
            b.wo = 0.0; b.wn = 0.0; b.off_older = 0; b.off_newer = 0;
            if (act[e]) {
                const bool hit = h.H >= 3 && t_o[e] <= q[e] && (gi[e] == 0 || t_n[e] > q[e]);
                b = hit ? bracket_weights(h, q[e], gi[e], t_n[e], t_o[e], a.error_flag) : find_bracket(h, q[e], a.error_flag);
            }
            const int idx = base + e * kConvThreads;
            t_wo[idx] = b.wo * w[e];
            t_wn[idx] = (b.off_newer >= 0) ? b.wn * w[e] : 0.0;
            if constexpr (BYTES) {
                t_oo[idx] = (b.off_older / D) * 8;
            } else {
                t_oo[idx] = b.off_older / D;          // ring slot of the older sample
                t_on[idx] = max(b.off_newer, 0) / D;  // ... of the newer one
            }
        }
    }
}

template <int MT, int R, int NB, bool UNI>
__device__ __forceinline__ void block_rad_stream(const BlockArgs& a, const int chunk, const int grp, double* red, double* t_wo, double* t_wn,
                                                 int* t_oo, int* t_on) {
    constexpr int L = 16 * NB;  // steps of the pass
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, jstep = lane & 15;
    const int D = a.hist.D;
    const int gp0 = chunk * a.chunk_gp;
    const int gp1 = min((a.F + 7) >> 3, gp0 + a.chunk_gp);
    const int s0  = (gp0 * 8) / D;
    const int ns  = (min(a.F, gp1 * 8) - 1) / D - s0 + 1;
    const int s_live = a.F / D;  // F is a whole number of samples
    build_block_table<L, false>(a, chunk, s0, ns, s_live, t_wo, t_wn, t_oo, t_on);
    __syncthreads();

    dvec4 acc[NB][MT];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[tb][m] = dvec4{0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ kbase = a.K.base + ((size_t)(grp * MT) * a.K.ngp) * 128 + lane * 2;
    const size_t tile_stride         = (size_t)a.K.ngp * 128;
    // B operands are gathered from the per-DoF copy of the ring: for consecutive steps j the query times move by one
    // history step, so the 16 lanes of one column read (up to a wrap of the ring) 16 consecutive doubles -- one or two
    // cache lines per column instead of one line per step from the sample-major ring.
    const double* __restrict__ ring = a.hist.ring_vT;
    const int Hc                    = a.hist.HcapT;  // row length

    // pipeline registers (UNI: the two columns of a lane lie in the same IRF sample and share bracket and weights)
    constexpr int HB = UNI ? 1 : 2;
    dvec2 kv[R][MT];
    double vo[R][2][NB], vn[R][2][NB], wo_[R][HB][NB], wn_[R][HB][NB];
    // issue-side position: column group gp_i of this wave
    int gp_i = gp0 + wave;
    // trackers of the two columns of a lane (half 0: 8gp + kk, half 1: 8gp + 4 + kk): IRF sample and column inside it.
    // UNI: both halves of all lanes are in sample s_u, the group's first column is column cb_u of it (wave-uniform).
    int s_t[2], c_t[2];
    int s_u = (gp_i * 8) / D, cb_u = gp_i * 8 - s_u * D;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int f = gp_i * 8 + 4 * h + kk;
        s_t[h] = UNI ? s_u : f / D;
        c_t[h] = f - s_t[h] * D;
    }
    double bwo[HB][NB], bwn[HB][NB];
    int boo[HB][NB], bon[HB][NB];
    auto load_bracket = [&](int h) {
        // columns past the chunk's last live sample (the rest of its last column group, or the groups the wave re-reads
        // past the end) get zero weights
        const int ks = s_t[h] - s0;
        const bool in = ks >= 0 && ks < ns;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
            const int k = (in ? ks : 0) * L + 16 * tb + jstep;
            bwo[h][tb] = in ? t_wo[k] : 0.0;
            bwn[h][tb] = in ? t_wn[k] : 0.0;
            boo[h][tb] = in ? t_oo[k] : 0;
            bon[h][tb] = in ? t_on[k] : 0;
        }
    };
    load_bracket(0);
    if constexpr (!UNI) load_bracket(1);

    auto issue = [&](const int slot) {
        // Unconditional: every call issues the same MT + 4*NB loads, so the compiler's vmcnt bookkeeping stays exact (a
        // branch around the loads degrades every wait in the loop to "all but a few").  Past the end of the chunk the
        // wave re-reads its last column group with zero weights.
        const bool live = gp_i < gp1;
        const int gpc   = live ? gp_i : gp1 - 1;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            kv[slot][m] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kbase + (size_t)m * tile_stride + (size_t)gpc * 128));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) {
                vo[slot][h][tb] = ring[c_t[h] * Hc + boo[h % HB][tb]];
                vn[slot][h][tb] = ring[c_t[h] * Hc + bon[h % HB][tb]];
            }
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) {
                wo_[slot][h][tb] = live ? bwo[h][tb] : 0.0;
                wn_[slot][h][tb] = live ? bwn[h][tb] : 0.0;
            }
        // advance to this wave's next column group (32 columns on)
        gp_i += 4;
        if constexpr (UNI) {
            cb_u += 32;
            if (cb_u >= D) {  // scalar branch
                do {
                    cb_u -= D;
                    ++s_u;
                } while (cb_u >= D);
                s_t[0] = s_u;
                load_bracket(0);
            }
            c_t[0] = cb_u + kk;
            c_t[1] = cb_u + 4 + kk;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                c_t[h] += 32;
                while (c_t[h] >= D) {
                    c_t[h] -= D;
                    ++s_t[h];
                }
                load_bracket(h);
            }
        }
    };
    auto consume = [&](const int slot) {
        double u[2][NB];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) u[h][tb] = fma(wo_[slot][h % HB][tb], vo[slot][h][tb], wn_[slot][h % HB][tb] * vn[slot][h][tb]);  // explicit: every instantiation rounds alike
        // consecutive MFMAs use different accumulators
#pragma unroll
        for (int tb = 0; tb < NB; ++tb)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[slot][m].x, u[0][tb], acc[tb][m], 0, 0, 0);
#pragma unroll
        for (int tb = 0; tb < NB; ++tb)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[slot][m].y, u[1][tb], acc[tb][m], 0, 0, 0);
    };

    const int nfrag = (gp1 - gp0 - wave + 3) / 4;  // column groups of this wave
#pragma unroll
    for (int r = 0; r < R; ++r) issue(r);
    for (int i = 0; i < nfrag; i += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            consume(r);
            issue(r);
        }
    }

    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg.  red[wave][m][row*16 + j], one block of
    // 16 steps at a time (the buffer holds one).
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * MT + m) * 256 + (kk + 4 * r) * 16 + jstep] = acc[tb][m][r];
        __syncthreads();
        for (int idx = tid; idx < MT * 256; idx += kConvThreads) {
            const int m = idx >> 8, el = idx & 255, row = el >> 4, j = el & 15;
            const double v = ((red[(0 * MT + m) * 256 + el] + red[(1 * MT + m) * 256 + el]) + red[(2 * MT + m) * 256 + el]) + red[(3 * MT + m) * 256 + el];
            a.partials[((size_t)chunk * L + 16 * tb + j) * a.Dpad + (grp * MT + m) * 16 + row] = v;
        }
    }
}

// The same work item for D % 8 == 0 (every multiple-of-4-bodies system, C3 / C4 among them), written for the instruction
// issue budget: with the K stream at the HBM rate and the B operands coming from the per-DoF ring, what bounds the pass
// is the vector-instruction count beside the MFMAs (one wave per SIMD at depth 32).  Differences from the general form:
//   * all samples / columns of a column group are wave-uniform, so the trackers are scalars (SALU);
//   * K and ring addresses are a uniform base (SGPR pair) + a 32-bit per-lane byte offset -- one v_add per gather, none per K load;
//   * the weights are not carried through the pipeline: a second, consume-side tracker R fragments behind the issue side
//     re-reads them from LDS when its sample changes (once per D/32 fragments), one fragment ahead of their use;
//   * fragments past the end of the chunk skip their MFMAs by a scalar branch instead of zeroed weights.
template <int MT, int R, int NB>
__device__ __forceinline__ void block_rad_stream_uni(const BlockArgs& a, const int chunk, const int grp, double* red, double* t_wo, double* t_wn,
                                                     int* t_oo, int* t_on) {
    constexpr int L = 16 * NB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, jstep = lane & 15;
    const int D = a.hist.D;
    const int gp0 = chunk * a.chunk_gp;
    const int gp1 = min((a.F + 7) >> 3, gp0 + a.chunk_gp);
    const int s0  = (gp0 * 8) / D;
    const int ns  = (min(a.F, gp1 * 8) - 1) / D - s0 + 1;
    const int s_live = a.F / D;
    const int Hc     = a.hist.HcapT;  // row length of the per-DoF ring

    dvec4 acc[NB][MT];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[tb][m] = dvec4{0.0, 0.0, 0.0, 0.0};
    const char* __restrict__ kb    = reinterpret_cast<const char*>(a.K.base) + ((size_t)(grp * MT) * a.K.ngp) * 1024;  // uniform
    const size_t tile_bytes        = (size_t)a.K.ngp * 1024;
    const unsigned lane16          = (unsigned)lane * 16u;
    const char* __restrict__ ringb = reinterpret_cast<const char*>(a.hist.ring_vT);  // uniform
    const unsigned col_bytes       = (unsigned)Hc * 8u;

    dvec2 kv[R][MT];
    dvec2u von[R][2][NB];  // {older, newer} sample of the bracket: adjacent entries of the lane's column (the ring row is mirrored past its end)

    build_block_table<L, true>(a, chunk, s0, ns, s_live, t_wo, t_wn, t_oo, t_on);  // t_oo: byte offset of the older sample in a ring row; the newer one (slot + 1 mod Hcap, or the not-yet-known sample, whose weight is 0) follows it
    __syncthreads();

    // ---- issue side ----
    int gp_i = gp0 + wave;
    int s_i = (gp_i * 8) / D, cb_i = gp_i * 8 - s_i * D;          // scalars
    unsigned cby = (unsigned)(cb_i + kk) * col_bytes;              // byte offset of the lane's first column (second: + 4 columns)
    unsigned boo[NB];
    auto load_offsets = [&]() {
        const int ks  = s_i - s0;
        const bool in = ks >= 0 && ks < ns;  // past the chunk's samples: any valid address will do (those fragments are skipped or weigh 0)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) boo[tb] = (unsigned)t_oo[(in ? ks : 0) * L + 16 * tb + jstep];
    };
    load_offsets();
    auto issue = [&](const int slot) {
        // unconditional (exact vmcnt bookkeeping); past the end of the chunk the wave re-reads its last column group
        const int gpc = min(gp_i, gp1 - 1);
        const char* __restrict__ kg = kb + (size_t)gpc * 1024;  // uniform
#pragma unroll
        for (int m = 0; m < MT; ++m) kv[slot][m] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kg + m * tile_bytes + lane16));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) {
                const unsigned cb = cby + (unsigned)(4 * h) * col_bytes;
                von[slot][h][tb]  = *reinterpret_cast<const dvec2u*>(ringb + (cb + boo[tb]));
            }
        gp_i += 4;
        cb_i += 32;
        cby += 32u * col_bytes;
        if (cb_i >= D) {  // scalar branch
            do {
                cb_i -= D;
                cby -= (unsigned)D * col_bytes;
                ++s_i;
            } while (cb_i >= D);
            load_offsets();
        }
    };

    // ---- consume side: the same walk, R fragments behind ----
    int gp_c = gp0 + wave;
    int s_c = s_i - 0, cb_c = gp_c * 8 - ((gp_c * 8) / D) * D;
    s_c     = (gp_c * 8) / D;
    double cwo[NB], cwn[NB];
    auto load_weights = [&]() {
        const int ks  = s_c - s0;
        const bool in = ks >= 0 && ks < ns;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
            const int k = (in ? ks : 0) * L + 16 * tb + jstep;
            cwo[tb]     = in ? t_wo[k] : 0.0;
            cwn[tb]     = in ? t_wn[k] : 0.0;
        }
    };
    load_weights();
    auto consume = [&](const int slot) {
        if (gp_c < gp1) {  // scalar branch around the matrix work of fragments past the end (no loads inside)
            if constexpr (NB > 2) {
                // Depth 64 is bound by the matrix pipe, not by HBM, and a wave alone on its SIMD issues in order: with all B operands
                // formed first (what the scheduler makes of the form below) the pipe runs dry during those 16 vector instructions of
                // every fragment (SQ counters, profiles/r05: matrix pipe busy 0.77).  So each B operand is formed right behind the
                // MFMAs of the group before -- in their shadow -- and nothing may move across the group boundaries.
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int tb = 0; tb < NB; ++tb) {
                        const double u = fma(cwo[tb], von[slot][h][tb].x, cwn[tb] * von[slot][h][tb].y);  // (the expression of the other depths: rounds alike)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(h == 0 ? kv[slot][m].x : kv[slot][m].y, u, acc[tb][m], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
            double u[2][NB];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int tb = 0; tb < NB; ++tb) u[h][tb] = fma(cwo[tb], von[slot][h][tb].x, cwn[tb] * von[slot][h][tb].y);  // explicit: every instantiation rounds alike
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[slot][m].x, u[0][tb], acc[tb][m], 0, 0, 0);
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[slot][m].y, u[1][tb], acc[tb][m], 0, 0, 0);
            }
        }
        gp_c += 4;
        cb_c += 32;
        if (cb_c >= D) {
            do {
                cb_c -= D;
                ++s_c;
            } while (cb_c >= D);
            load_weights();  // for the next fragment; the LDS latency hides behind the MFMAs just issued
        }
    };

    const int nfrag = (gp1 - gp0 - wave + 3) / 4;
#pragma unroll
    for (int r = 0; r < R; ++r) issue(r);
    for (int i = 0; i < nfrag; i += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            consume(r);
            issue(r);
        }
    }

#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * MT + m) * 256 + (kk + 4 * r) * 16 + jstep] = acc[tb][m][r];
        __syncthreads();
        for (int idx = tid; idx < MT * 256; idx += kConvThreads) {
            const int m = idx >> 8, el = idx & 255, row = el >> 4, j = el & 15;
            const double v = ((red[(0 * MT + m) * 256 + el] + red[(1 * MT + m) * 256 + el]) + red[(2 * MT + m) * 256 + el]) + red[(3 * MT + m) * 256 + el];
            a.partials[((size_t)chunk * L + 16 * tb + j) * a.Dpad + (grp * MT + m) * 16 + row] = v;
        }
    }
}

#ifdef HC_TUNING
// Depth 64, second form (tuning build; EXPERIMENTS.md): the first form above issues the MT + 2*NB loads of a fragment and their address
// arithmetic in one run behind the fragment's 8*MT MFMAs, and a wave alone on its SIMD issues in order -- the counters of profiles/r05 show
// that run (about 600 cycles per fragment of 3072 matrix-pipe cycles) exposed, not hidden: only what is issued within the 64 cycles of
// the last MFMA runs in its shadow.  Here the loads of the fragment NS - 1 ahead go out one at a time BETWEEN the MFMAs of the fragment
// being consumed (one load behind every MT/2 MFMAs), into a register slot that is not being read (NS slots, NS - 1 fragments in flight).
template <int MT, int NS, int NB, int VAR = 0>
__device__ __forceinline__ void block_rad_stream_il(const BlockArgs& a, const int chunk, const int grp, double* red, double* t_wo, double* t_wn,
                                                    int* t_oo, int* t_on) {
    constexpr int L = 16 * NB;
    static_assert(MT % 2 == 0 && MT + 2 * NB <= 4 * NB, "one load behind every half group of MFMAs");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, jstep = lane & 15;
    const int D = a.hist.D;
    const int gp0 = chunk * a.chunk_gp;
    const int gp1 = min((a.F + 7) >> 3, gp0 + a.chunk_gp);
    const int s0  = (gp0 * 8) / D;
    const int ns  = (min(a.F, gp1 * 8) - 1) / D - s0 + 1;
    const int s_live = a.F / D;
    const int Hc     = a.hist.HcapT;

    dvec4 acc[NB][MT];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[tb][m] = dvec4{0.0, 0.0, 0.0, 0.0};
    const char* __restrict__ kb    = reinterpret_cast<const char*>(a.K.base) + ((size_t)(grp * MT) * a.K.ngp) * 1024;
    const size_t tile_bytes        = (size_t)a.K.ngp * 1024;
    const unsigned lane16          = (unsigned)lane * 16u;
    const char* __restrict__ ringb = reinterpret_cast<const char*>(a.hist.ring_vT);
    const unsigned col_bytes       = (unsigned)Hc * 8u;

    dvec2 kv[NS][MT];
    dvec2u von[NS][2][NB];

    build_block_table<L, true>(a, chunk, s0, ns, s_live, t_wo, t_wn, t_oo, t_on);
    __syncthreads();

    // ---- issue side ----
    int gp_i = gp0 + wave;
    int s_i = (gp_i * 8) / D, cb_i = gp_i * 8 - s_i * D;
    unsigned cby = (unsigned)(cb_i + kk) * col_bytes;
    const unsigned cby0 = cby;
    unsigned boo[NB];
    auto load_offsets = [&]() {
        const int ks  = s_i - s0;
        const bool in = ks >= 0 && ks < ns;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) boo[tb] = (unsigned)t_oo[(in ? ks : 0) * L + 16 * tb + jstep];
    };
    load_offsets();
    // load number k of the fragment at the issue-side trackers: K tiles first (the long latency), then the gathers
    // VAR > 0 (timing bounds of EXPERIMENTS.md, results wrong): 1 = the loop issues no gathers, 2 = no loads at all, 3 = MFMAs only
    auto issue_part = [&](const int slot, const int k, const char* __restrict__ kg, const bool in_loop = false) {
        if (in_loop && (VAR == 2 || VAR == 3)) return;
        if (in_loop && VAR == 1 && k >= MT) return;
        // VAR 4: the gathers stay on the chunk's first 8 columns (cache hits); 5: K loads without the non-temporal hint; 6: the K loads of a
        // fragment from 6 KB in a row (one stream per wave instead of MT) -- timing bounds as well
        if (k < MT) {
            if constexpr (VAR == 5) kv[slot][k] = *reinterpret_cast<const dvec2*>(kg + k * tile_bytes + lane16);
            else if constexpr (VAR == 6) kv[slot][k] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kb + ((size_t)(kg - kb) * MT + k * 1024 + lane16)));
            else kv[slot][k] = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(kg + k * tile_bytes + lane16));
        } else if (k < MT + 2 * NB) {
            const int q = k - MT, h = q / NB, tb = q % NB;
            von[slot][h][tb] = *reinterpret_cast<const dvec2u*>(ringb + ((VAR == 4 ? cby0 : cby) + (unsigned)(4 * h) * col_bytes + boo[tb]));
        }
    };
    auto issue_advance = [&]() {
        gp_i += 4;
        cb_i += 32;
        cby += 32u * col_bytes;
        if (cb_i >= D) {
            do {
                cb_i -= D;
                cby -= (unsigned)D * col_bytes;
                ++s_i;
            } while (cb_i >= D);
            load_offsets();
        }
    };

    // ---- consume side ----
    int gp_c = gp0 + wave;
    int s_c = (gp_c * 8) / D, cb_c = gp_c * 8 - s_c * D;
    double cwo[NB], cwn[NB];
    auto load_weights = [&]() {
        const int ks  = s_c - s0;
        const bool in = ks >= 0 && ks < ns;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
            const int k = (in ? ks : 0) * L + 16 * tb + jstep;
            cwo[tb]     = in ? t_wo[k] : 0.0;
            cwn[tb]     = in ? t_wn[k] : 0.0;
        }
    };
    load_weights();

    const int nfrag = (gp1 - gp0 - wave + 3) / 4;
#pragma unroll
    for (int r = 0; r < NS - 1; ++r) {
        const char* __restrict__ kg = kb + (size_t)min(gp_i, gp1 - 1) * 1024;
#pragma unroll
        for (int k = 0; k < MT + 2 * NB; ++k) issue_part(r, k, kg);
        issue_advance();
    }
    for (int i = 0; i < nfrag; i += NS) {
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            constexpr int H = MT / 2;
            const int is = (r + NS - 1) % NS;  // free: consumed in the step before
            const char* __restrict__ kg = kb + (size_t)min(gp_i, gp1 - 1) * 1024;  // past the end of the chunk: the last column group again (never consumed)
            if (gp_c < gp1) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int tb = 0; tb < NB; ++tb) {
                        const double u = VAR == 3 ? cwo[tb] : fma(cwo[tb], von[r][h][tb].x, cwn[tb] * von[r][h][tb].y);  // (the expression of the other depths: rounds alike)
#pragma unroll
                        for (int m = 0; m < H; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(h == 0 ? kv[r][m].x : kv[r][m].y, u, acc[tb][m], 0, 0, 0);
                        issue_part(is, 2 * (h * NB + tb), kg, true);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = H; m < MT; ++m) acc[tb][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(h == 0 ? kv[r][m].x : kv[r][m].y, u, acc[tb][m], 0, 0, 0);
                        issue_part(is, 2 * (h * NB + tb) + 1, kg, true);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < MT + 2 * NB; ++k) issue_part(is, k, kg, true);
            }
            issue_advance();
            gp_c += 4;
            cb_c += 32;
            if (cb_c >= D) {
                do {
                    cb_c -= D;
                    ++s_c;
                } while (cb_c >= D);
                load_weights();
            }
        }
    }

#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * MT + m) * 256 + (kk + 4 * r) * 16 + jstep] = acc[tb][m][r];
        __syncthreads();
        for (int idx = tid; idx < MT * 256; idx += kConvThreads) {
            const int m = idx >> 8, el = idx & 255, row = el >> 4, j = el & 15;
            const double v = ((red[(0 * MT + m) * 256 + el] + red[(1 * MT + m) * 256 + el]) + red[(2 * MT + m) * 256 + el]) + red[(3 * MT + m) * 256 + el];
            a.partials[((size_t)chunk * L + 16 * tb + j) * a.Dpad + (grp * MT + m) * 16 + row] = v;
        }
    }
}
#endif

template <int MT, int R, int NB, int WPS = ((NB == 1 && MT <= 6) ? 2 : 1)>
__global__ void __launch_bounds__(kConvThreads, WPS) conv_block_kernel(BlockArgs a) {
    // dynamic LDS: [front: cross-wave reduction buffer / U tiles of the excitation items][bracket table, SoA: wo', wn', off_older, off_newer]
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* front = reinterpret_cast<double*>(smem_raw);
    const int nt  = a.max_steps_per_chunk * 16 * NB;
    double* t_wo  = front + a.lds_front_doubles;
    double* t_wn  = t_wo + nt;
    int* t_oo     = reinterpret_cast<int*>(t_wn + nt);
    int* t_on     = t_oo + nt;

    // Block index -> (chunk, row group).  Workgroups are dealt to the 8 XCDs round-robin by block index, and the row groups
    // of one chunk gather the same ring rows, so they get block indices that are congruent mod 8 and close together
    // ([octet of chunks][row group][chunk % 8]): the gathers of all but the first then hit that XCD's L2.
    const int r     = (int)blockIdx.x % (8 * a.ngroups);
    const int chunk = a.chunk_first + ((int)blockIdx.x / (8 * a.ngroups)) * 8 + (r & 7);
    const int grp   = r >> 3;
    if (chunk >= (a.chunk_last > 0 ? a.chunk_last : a.nchunks)) return;

    // the per-DoF ring must be addressable with 32-bit byte offsets for the scalar-base form (4 GB: far beyond any real history)
    if constexpr (NB > 2) {
        // depth 64 (experimental): the uniform form only -- the host selects this depth for D % 8 == 0 systems
#ifdef HC_TUNING
        if constexpr (R >= 13 && MT % 2 == 0) block_rad_stream_il<MT, 3, NB, R - 12>(a, chunk, grp, front, t_wo, t_wn, t_oo, t_on);  // R = 13 ... 18: timing bounds (wrong results)
        else if constexpr (R >= 10 && MT % 2 == 0) block_rad_stream_il<MT, R - 8, NB>(a, chunk, grp, front, t_wo, t_wn, t_oo, t_on);  // R = 11 / 12: the interleaved form with 3 / 4 slots
        else
#endif
        block_rad_stream_uni<MT, R, NB>(a, chunk, grp, front, t_wo, t_wn, t_oo, t_on);
    } else {
        if ((a.hist.D & 7) == 0 && (size_t)a.hist.D * a.hist.HcapT < ((size_t)1 << 28)) block_rad_stream_uni<MT, R, NB>(a, chunk, grp, front, t_wo, t_wn, t_oo, t_on);
        else block_rad_stream<MT, (NB > 1) ? 2 : R, NB, false>(a, chunk, grp, front, t_wo, t_wn, t_oo, t_on);
    }
    // The excitation force depends on time only: its work items over Kex (a fraction of a percent of K) for the predicted times ride
    // at the end of radiation workgroups, so the launch keeps its number of workgroups (grid rounds on the chip).  An item is
    // (excitation chunk, block of 16 steps, group of MTE row tiles) and takes about 8 us of dependent loads (free-surface table
    // -> LDS -> MFMA).  Workgroups take items from a counter as they finish their radiation chunk: the radiation chunks end
    // over a span of 10-15 us (HBM channels are not perfectly even), so the early finishers absorb the items and the launch
    // ends with its slowest radiation chunk.  (Stacked on the first 16 workgroups the items were a 28 us tail at C3; dealt one
    // per workgroup, 8 us.)  Every item writes its own partials, so the result does not depend on who computes it.
    constexpr int MTE = MT > 6 ? 6 : MT;  // row tiles per excitation work item (register budget of the LDS-staged form)
    const int RG      = a.ngroups * (MT / MTE);
    const int n_items = a.nchunks_ex * NB * RG;
    if (n_items > 0) {
        __shared__ int s_item;
        for (;;) {
            __syncthreads();  // the reduction buffer of the previous work item aliases the U tiles of the next one
            if (threadIdx.x == 0) s_item = atomicAdd(a.item_counter, 1);
            __syncthreads();
            const int it = s_item;
            if (it >= n_items) break;
            const int e = it / (NB * RG), rem = it - e * (NB * RG), tb = rem / RG, rg = rem - tb * RG;
            block_exc_work<MTE>(a, rg, e, 16 * tb, front);
        }
    }
}

template <class KernelT>
static void allow_dynamic_lds(KernelT kernel, size_t smem, size_t& granted) {
    if (smem > 64 * 1024 && smem > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        granted = smem;
    }
}

template <int MT, int R, int R32 = R + 1>
static void launch_conv_block_mt(const BlockArgs& b, int nblocks, size_t smem, hipStream_t stream) {
    static size_t granted16 = 0, granted32 = 0;
    if (b.depth == 32) {
#ifdef HC_TUNING  // variants of the depth-32 pass measured and not taken (EXPERIMENTS.md): the tuning build keeps them selectable
        static const int v32 = [] { const char* e = std::getenv("HC_BLOCK_V32"); return e ? std::atoi(e) : 0; }();
        if constexpr (MT == 4) {
            if (v32 == 1) { hipLaunchKernelGGL((conv_block_kernel<4, 3, 2, 2>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (v32 == 2) { hipLaunchKernelGGL((conv_block_kernel<4, 4, 2, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (v32 == 3) { hipLaunchKernelGGL((conv_block_kernel<4, 6, 2, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
        }
        if constexpr (MT == 6) {
            if (v32 == 4) { hipLaunchKernelGGL((conv_block_kernel<6, 5, 2, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (v32 == 5) { hipLaunchKernelGGL((conv_block_kernel<6, 3, 2, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (v32 == 6) { hipLaunchKernelGGL((conv_block_kernel<6, 6, 2, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
        }
#endif
        allow_dynamic_lds(conv_block_kernel<MT, R32, 2>, smem, granted32);
        hipLaunchKernelGGL((conv_block_kernel<MT, R32, 2>), dim3(nblocks), dim3(kConvThreads), smem, stream, b);
    } else {
        allow_dynamic_lds(conv_block_kernel<MT, R, 1>, smem, granted16);
        hipLaunchKernelGGL((conv_block_kernel<MT, R, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b);
    }
}

#ifdef HC_TUNING
// depth 64 (NB = 4; tuning build only -- measured in round 5 and not taken, EXPERIMENTS.md): MT row tiles per workgroup x R fragments in
// flight, both from the environment for the sweep of profiles/r05 (HC_BLOCK64_R; the tile count comes with the launch)
static int block64_R() {
    static const int r = [] { const char* e = std::getenv("HC_BLOCK64_R"); const int v = e ? std::atoi(e) : 4; return (v == 2 || v == 3 || v == 5 || (v >= 11 && v <= 18)) ? v : 4; }();
    return r;
}
template <int MT>
static void launch_conv_block64_mt(const BlockArgs& b, int nblocks, size_t smem, hipStream_t stream) {
    static size_t granted[5] = {0, 0, 0, 0, 0};
    const int R = block64_R();
    if constexpr (MT == 3) {
        // two workgroups per CU (two waves per SIMD, 256 registers each): the matrix pipe of a SIMD is fed by two instruction streams
        static size_t g3 = 0;
        if (R == 2) { allow_dynamic_lds(conv_block_kernel<3, 2, 4, 2>, smem, g3); hipLaunchKernelGGL((conv_block_kernel<3, 2, 4, 2>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
    }
    if constexpr (MT % 2 == 0) {
        if (R == 11) { allow_dynamic_lds(conv_block_kernel<MT, 11, 4, 1>, smem, granted[3]); hipLaunchKernelGGL((conv_block_kernel<MT, 11, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
        if (R == 12) { allow_dynamic_lds(conv_block_kernel<MT, 12, 4, 1>, smem, granted[4]); hipLaunchKernelGGL((conv_block_kernel<MT, 12, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
        if constexpr (MT == 6) {
            static size_t g2[3] = {0, 0, 0};
            if (R == 13) { allow_dynamic_lds(conv_block_kernel<MT, 13, 4, 1>, smem, g2[0]); hipLaunchKernelGGL((conv_block_kernel<MT, 13, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (R == 14) { allow_dynamic_lds(conv_block_kernel<MT, 14, 4, 1>, smem, g2[1]); hipLaunchKernelGGL((conv_block_kernel<MT, 14, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (R == 15) { allow_dynamic_lds(conv_block_kernel<MT, 15, 4, 1>, smem, g2[2]); hipLaunchKernelGGL((conv_block_kernel<MT, 15, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            static size_t g3[3] = {0, 0, 0};
            if (R == 16) { allow_dynamic_lds(conv_block_kernel<MT, 16, 4, 1>, smem, g3[0]); hipLaunchKernelGGL((conv_block_kernel<MT, 16, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (R == 17) { allow_dynamic_lds(conv_block_kernel<MT, 17, 4, 1>, smem, g3[1]); hipLaunchKernelGGL((conv_block_kernel<MT, 17, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
            if (R == 18) { allow_dynamic_lds(conv_block_kernel<MT, 18, 4, 1>, smem, g3[2]); hipLaunchKernelGGL((conv_block_kernel<MT, 18, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); return; }
        }
    }
    if (R == 3) { allow_dynamic_lds(conv_block_kernel<MT, 3, 4, 1>, smem, granted[0]); hipLaunchKernelGGL((conv_block_kernel<MT, 3, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); }
    else if (R == 5) { allow_dynamic_lds(conv_block_kernel<MT, 5, 4, 1>, smem, granted[2]); hipLaunchKernelGGL((conv_block_kernel<MT, 5, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); }
    else { allow_dynamic_lds(conv_block_kernel<MT, 4, 4, 1>, smem, granted[1]); hipLaunchKernelGGL((conv_block_kernel<MT, 4, 4, 1>), dim3(nblocks), dim3(kConvThreads), smem, stream, b); }
}
#endif

BlockLaunch block_launch_config(const BlockArgs& a, int mt, BlockArgs* b) {
    BlockLaunch l;
    const int nlaunch = (a.chunk_last > 0 ? a.chunk_last : a.nchunks) - a.chunk_first;  // chunks of this launch
    l.nblocks = ((nlaunch + 7) >> 3) * 8 * a.ngroups;  // octets of chunks, see the kernel's block mapping
    *b        = a;
    b->lds_front_doubles = max(4 * kUWave, 4 * mt * 256);  // [wave][tile][16x16] reduction buffer / per-wave U sub-tiles
    l.smem    = (size_t)b->lds_front_doubles * sizeof(double) + (size_t)max(1, a.max_steps_per_chunk) * b->depth * 24;
    // template arguments of the kernel this (mt, depth) runs: conv_block_kernel<MT, R, NB, WPS>
#ifdef HC_TUNING
    if (a.depth == 64) {
        l.MT  = (mt == 6 || mt == 4 || mt == 3) ? mt : 3;
        l.NB  = 4;
        l.R   = ((block64_R() >= 10 && l.MT % 2) || (block64_R() >= 13 && l.MT != 6) || (block64_R() == 2 && l.MT != 3)) ? 4 : block64_R();
        l.WPS = l.R == 2 ? 2 : 1;
        return l;
    }
    l.MT  = (mt == 12 || mt == 6 || mt == 4 || mt == 2) ? mt : 1;  // (12 tiles per workgroup: HC_BLOCK_MT=12, measured and not taken)
#else
    l.MT  = (mt == 6 || mt == 4 || mt == 2) ? mt : 1;
#endif
    const int R16 = (l.MT == 12 || l.MT == 6) ? 3 : 4;
    l.NB  = a.depth == 32 ? 2 : 1;
    l.R   = a.depth == 32 ? (l.MT == 12 ? 2 : R16 + 1) : R16;
    l.WPS = (l.NB == 1 && l.MT <= 6) ? 2 : 1;
    return l;
}

void launch_conv_block(const BlockArgs& a, int mt, hipStream_t stream) {
    BlockArgs b;
    const BlockLaunch l = block_launch_config(a, mt, &b);
    const int nblocks   = l.nblocks;
    const size_t smem   = l.smem;
    if (nblocks <= 0) return;
#ifdef HC_TUNING
    if (b.depth == 64) {
        if (mt == 6) launch_conv_block64_mt<6>(b, nblocks, smem, stream);
        else if (mt == 4) launch_conv_block64_mt<4>(b, nblocks, smem, stream);
        else launch_conv_block64_mt<3>(b, nblocks, smem, stream);
        return;
    }
    if (mt == 12) { launch_conv_block_mt<12, 3, 2>(b, nblocks, smem, stream); return; }
#endif
    if (mt == 6) launch_conv_block_mt<6, 3>(b, nblocks, smem, stream);
    else if (mt == 4) launch_conv_block_mt<4, 4>(b, nblocks, smem, stream);
    else if (mt == 2) launch_conv_block_mt<2, 4>(b, nblocks, smem, stream);
    else launch_conv_block_mt<1, 4>(b, nblocks, smem, stream);
}

__device__ __forceinline__ double lane16_sum(double v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}

// P[j][row] = sum over radiation chunks c of partials[c][j][row]; E[j][row] = the same over the excitation chunks.
// 16 lanes per output, chunks c = l, l+16, ... (8 loads in flight per lane, adds in ascending chunk order), then a 4-step
// xor tree.
__global__ void __launch_bounds__(256) reduce_block_kernel(ReduceArgs a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.item_counter = 0;  // the pass that has just finished counted its excitation items here
    const double* __restrict__ partials = a.partials;
    const int sub = threadIdx.x & 15;
    const int n   = a.depth * a.Dpad;
    int out       = blockIdx.x * 16 + (threadIdx.x >> 4);  // [segment][j*Dpad + row]
    const bool exc = out >= n;
    if (exc) out -= n;
    const int first = exc ? a.nchunks_rad : a.rad_first, count = exc ? a.nchunks_ex : a.nchunks_rad - a.rad_first;
    const int o     = out < n ? out : 0;
    double v = 0.0;
    if (a.narrow) {
        // narrow short pass: chunk c holds [16][Dpad] partials of steps jbase[c] .. jbase[c] + 15; the chunks that do not hold this
        // output's step add 0.0, as they do in the wide form -- same lanes, same order, bitwise the wide form's sum
        const int j = o / a.Dpad, row = o - j * a.Dpad;
        for (int c = sub; c < count; c += 8 * 16) {
            double w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cc = first + c + 16 * k;
                const int jj = (c + 16 * k) < count ? j - (int)a.jbase[cc] : -1;
                w[k]         = (jj >= 0 && jj < 16) ? partials[((size_t)cc * 16 + jj) * a.Dpad + row] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v += w[k];
        }
    } else {
        for (int c = sub; c < count; c += 8 * 16) {
            double w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = (c + 16 * k) < count ? partials[(size_t)(first + c + 16 * k) * n + o] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += w[k];
        }
    }
    v = lane16_sum(v);
    if (out < n && sub == 0) {
        if (a.accumulate) {
            const int j = out / a.Dpad;
            if (!exc && j < a.j_cnt) a.P[out + (size_t)a.j_off * a.Dpad] += v;
        } else {
            (exc ? a.E : a.P)[out] = v;
        }
    }
}

int reduce_block_grid(const ReduceArgs& r) {
    const int nblk = (r.depth * r.Dpad + 15) / 16;  // depth * Dpad is a multiple of 16, so the excitation segment starts on a workgroup boundary
    return r.nchunks_ex > 0 ? 2 * nblk : nblk;
}

void launch_reduce_block(const ReduceArgs& r, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_block_kernel, dim3(reduce_block_grid(r)), dim3(256), 0, stream, r);
}

// ------------------------------------------------------------------------------------------------
// finalize_kernel (the "step kernel"): one workgroup per tile of 16 owned output rows (+ one that stores this step's
// sample into the ring).  It finishes a step whichever way its radiation term was prepared:
//   plain step          16 lanes per row add the chunk partials of conv_step_kernel in a fixed order;
//   step inside a block the workgroup contracts the few IRF samples that involve this step's own sample itself (its 16
//                       rows x n_near * D columns of K, right-hand side staged in LDS once -- row-owned, so nothing is
//                       left to reduce across workgroups), adds the look-ahead part P and the scatter results of the
//                       earlier steps of the block: the step is this ONE launch;
// then hydrostatics / the regular-wave or spectral term, total = hydrostatic - radiation + waves (src/hydro_forces.cpp:
// 263-322,758-760; src/wave_types.cpp:315-327).  All sums run in a fixed order (bitwise reproducible).  For the host
// boundary the totals also leave as 16-byte {value, sequence} granules in mapped pinned memory.
// ------------------------------------------------------------------------------------------------
// Stage clock (tuning build): where a workgroup's time goes between kernel entry and the last store -- the 100 MHz constant clock
// (s_memrealtime) read at fixed points of the program; a wave passes a point once everything in front of it has been waited for, so
// the differences are the lengths of the dependent hops.  Kept in scalar registers until the workgroup's last store has been
// acknowledged, then stored by work-item 0 (nothing is added to the memory traffic of the stages themselves).  The release build
// compiles the marks away.
//   0 entry   1 arguments in registers   2 every load requested   3 right-hand side in LDS (state arrived)   4 barrier passed
//   5 contraction done (K arrived)   6 reduction visible (second barrier)   7 totals formed   8 stores issued   9 stores acknowledged
#ifdef HC_TUNING
struct StageClock {
    unsigned long long t[kStampStages] = {};
    template <int K>
    __device__ __forceinline__ void mark() {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
        t[K] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __device__ __forceinline__ void store(unsigned long long* stamps) const {
        if (stamps && threadIdx.x == 0 && blockIdx.x < kStampWGs) {
#pragma unroll
            for (int k = 0; k < kStampStages; ++k) stamps[blockIdx.x * kStampStages + k] = t[k];
        }
    }
};
#define HC_MARK(sc, K) (sc).template mark<K>()
#define HC_STAMPS(a) ((a).stamps)
#else
struct StageClock {
    __device__ __forceinline__ void store(unsigned long long*) const {}
};
#define HC_MARK(sc, K) ((void)0)
#define HC_STAMPS(a) (static_cast<unsigned long long*>(nullptr))
#endif

// NW = waves per workgroup (4).  Wide systems (near_slices_for(D) > 1) leave the own-sample part to near_split_kernel.
// the workgroup that hands back the state canary and stores this step's sample into ring slot `head` (both layouts)
// SLOT: the state lies behind the kernel's argument block (st, layout of hc_limits.hpp: kSlotArgBytes), not at a.state.
template <int NW, bool SLOT = false>
__device__ __forceinline__ void push_sample(const FinalizeArgs& a, const double* __restrict__ st = nullptr) {
    if (threadIdx.x == 64 && a.canary_out) {  // first thing this workgroup does: the word the host stored behind the state goes back, tagged
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const double word = SLOT ? st[12 * a.N] : *a.canary_in;
        *reinterpret_cast<u64x2*>(a.canary_out) = u64x2{(unsigned long long)__double_as_longlong(word), a.seq};
    }
    // nobody reads ring slot `head` during this step (the current sample is always taken from `state`)
    if (threadIdx.x == 0) a.ring_t[a.head] = a.t;
    double* slot = a.ring_v + (size_t)a.head * a.D;
    for (int c = threadIdx.x; c < a.D; c += 64 * NW) {
        const double v = SLOT ? st[c] : state_velocity(a.state, a.N, c);
        slot[c] = v;
        a.ring_vT[(size_t)c * a.HcapT + a.head] = v;  // per-DoF time series for the look-ahead pass
        if (a.head == 0) a.ring_vT[(size_t)c * a.HcapT + a.Hcap] = v;  // mirror of slot 0 behind the last slot
    }
}

// One tile of 16 owned rows (the whole workgroup); `tile` is the workgroup's block index in finalize_kernel.  `mid` runs after everything
// the tile needs has been REQUESTED and before anything is waited for; it returns whether this workgroup goes on (wide_step_kernel
// contracts its column slice of the own-sample part there and goes on only if it completed the tile).  COHERENT: the slice partials
// were written by other workgroups of this very launch, possibly on other XCDs (agent-scope atomic loads, see wide_step_kernel).
// SLOT: the state lies behind the argument block (st) and the velocities of columns tid, tid + 256, tid + 512 were requested before the
// first argument was looked at (ev0..ev3); 6N <= 1024 (kSlotStateMaxBodies).
// EARLY (finalize_pre_kernel): the first K words of the wave and the scatter results were requested at kernel entry, from addresses the
// packet processor had put into scalar registers (kernel-argument preload) -- `early` holds them.
struct EarlyLoads {
    dvec2 pre[12];
    double ypre[kTermMax / 16];
};
template <int NW, bool COHERENT, bool SLOT, bool EARLY = false, class Mid>
__device__ __forceinline__ void finalize_tile(const FinalizeArgs& a, const int tile, double* U, double (*red_near)[16], double (*red_term)[16], Mid&& mid,
                                              StageClock& sc, const double* __restrict__ st = nullptr, double ev0 = 0.0, double ev1 = 0.0, double ev2 = 0.0,
                                              double ev3 = 0.0, const EarlyLoads* early = nullptr) {
    static_assert(!SLOT || NW == 4, "the early loads assume 256 work-items");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub  = tid & 15;
    const bool rowthread = tid < 256;         // the first four waves own the rows; further waves only stream K
    const int rit  = (tid >> 4) & 15;         // row inside the tile
    const int row  = tile * 16 + rit;
    const bool live = rowthread && row < a.Dloc;
    const int rrow  = live ? row : 0;

    // lane `sub` adds chunks sub, sub+16, ... in ascending order.  The loads are issued 8 at a time, the tail batch too
    // (masked slots load nothing and add +0.0): a chunk list costs one round trip per 8 entries instead of one per entry
    // through the loop-carried add, and the order of the adds is unchanged.
    auto lane_sum = [&](int first, int count) {
        double acc = 0.0;
        for (int c = sub; c < count; c += 8 * 16) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = c + 16 * k;
                v[k] = idx < count ? a.partials[(size_t)(first + idx) * a.Dpad + rrow] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        return acc;
    };
    // what lane 0 of the row needs besides the sums is requested first, so that those loads are in flight together with
    // everything else instead of after it
    const int bl = rrow / 6, i = rrow - 6 * bl;  // local body, DoF
    const int b  = a.b0 + bl;                    // global body
    const bool finisher = live && sub == 0;
    double p_row = 0.0, e_row = 0.0, dq[6] = {0, 0, 0, 0, 0, 0}, krow[6] = {0, 0, 0, 0, 0, 0}, V = 0.0, r[3] = {0, 0, 0};
    if (finisher) {
        if (a.do_rad && a.P) p_row = a.P[rrow];
        if (a.do_waves && a.wave_mode == 2 && a.E) e_row = a.E[rrow];
        if (a.do_hs) {
            const double* pos = SLOT ? st + 6 * a.N + 3 * b : a.state + 3 * b;
            const double* rpy = SLOT ? st + 9 * a.N + 3 * b : a.state + 3 * a.N + 3 * b;
            const double* cg  = a.cg + 3 * bl;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dq[j]     = pos[j] - cg[j];
                dq[3 + j] = rpy[j] - 0.0;  // equilibrium rotations are zero (src/hydro_forces.cpp:208-216)
                r[j]      = a.cb_m_cg[3 * bl + j];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) krow[j] = a.lin[36 * bl + 6 * i + j];
            V = a.disp_vol[bl];
        }
    }

    const bool near_on = a.do_rad && a.n_near > 0, term_on = a.do_rad && a.n_terms > 0;
    // Loads first, uses later: the step is a chain of dependent round trips (state -> LDS -> K x u; table -> Y), so everything
    // that can be requested up front is -- the first K words of the wave, the scatter results it will add -- before the
    // first wait.
    const int kk = lane >> 4;
    constexpr int PRE = 12;  // C3: all 12 column groups a wave owns of one IRF sample
    static_assert(PRE == sizeof(EarlyLoads::pre) / sizeof(dvec2), "EarlyLoads::pre");
    dvec2 pre[PRE];
    const double* __restrict__ kbase = a.nearK.base + ((size_t)tile * a.nearK.ngp) * 128 + lane * 2;
    if constexpr (EARLY) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) pre[q] = early->pre[q];
    } else if (near_on) {
        const int f0_0 = a.near[0].s * a.D, g0_0 = f0_0 >> 3, g1_0 = (f0_0 + a.D + 7) >> 3;
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int gp = g0_0 + wave + NW * q;
            pre[q] = gp < g1_0 ? *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128) : dvec2{0.0, 0.0};
        }
    }
    // scatter results of the earlier steps of the block: the scatter launches have left them weighted and contiguous
    // (Yc[k][row]); thread (slice = tid >> 4, row = tid & 15) requests terms slice, slice + 16, ... here, before the first wait
    constexpr int TPRE = kTermMax / 16;
    double ypre[TPRE];
    if constexpr (EARLY) {
#pragma unroll
        for (int q = 0; q < TPRE; ++q) ypre[q] = early->ypre[q];
    } else if (term_on) {
        const double* __restrict__ yc = a.Yc + tile * 16 + (tid & 15);
#pragma unroll
        for (int q = 0; q < TPRE; ++q) {
            const int k = (tid >> 4) + 16 * q;
            ypre[q]     = (rowthread && k < a.n_terms) ? yc[(size_t)k * a.Dpad] : 0.0;
        }
    }
    HC_MARK(sc, 2);
    if (!mid()) return;
    if (near_on) {
        // ---- the IRF samples this step contracts itself: rows of this tile x [s*D, (s+1)*D) ----
        const int D = a.D;
        for (int e = 0; e < a.n_near; ++e) {
            const NearEntry& ne = a.near[e];
            int k = 0;
            for (int col = tid; col < D; col += 64 * NW, ++k) {
                double u = 0.0;
                if (ne.a != 0.0) u = ne.a * (SLOT ? (k == 0 ? ev0 : (k == 1 ? ev1 : (k == 2 ? ev2 : ev3))) : state_velocity(a.state, a.N, col));
                if (ne.b != 0.0) u = fma(ne.b, a.ring_v_ro[ne.off_b + col], u);
                if (ne.c != 0.0) u = fma(ne.c, a.ring_v_ro[ne.off_c + col], u);
                U[e * D + col] = u;
            }
        }
        HC_MARK(sc, 3);
        __syncthreads();
        HC_MARK(sc, 4);
        double acc = 0.0;
        for (int e = 0; e < a.n_near; ++e) {
            const int f0 = a.near[e].s * D, f1 = f0 + D;
            const int g0 = f0 >> 3, g1 = (f1 + 7) >> 3;
            const double* __restrict__ u = U + e * D - f0;
            int gp = g0 + wave;
            if (e == 0) {
#pragma unroll
                for (int q = 0; q < PRE; ++q, gp += NW) {
                    if (gp < g1) {
                        const int fa = gp * 8 + kk, fb = fa + 4;
                        const double u0 = (fa >= f0 && fa < f1) ? u[fa] : 0.0, u1 = (fb >= f0 && fb < f1) ? u[fb] : 0.0;
                        acc = fma(pre[q].x, u0, acc);
                        acc = fma(pre[q].y, u1, acc);
                    }
                }
            }
#pragma unroll 4
            for (; gp < g1; gp += NW) {
                const dvec2 kv = *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128);
                const int fa = gp * 8 + kk, fb = fa + 4;
                const double u0 = (fa >= f0 && fa < f1) ? u[fa] : 0.0, u1 = (fb >= f0 && fb < f1) ? u[fb] : 0.0;
                acc = fma(kv.x, u0, acc);
                acc = fma(kv.y, u1, acc);
            }
        }
        acc += __shfl_xor(acc, 16, kWave);
        acc += __shfl_xor(acc, 32, kWave);
#ifdef HC_TUNING
        asm volatile("" ::"v"(acc));  // (the mark below stands behind the contraction's last use of a K word)
#endif
        HC_MARK(sc, 5);
        if (lane < 16) red_near[wave][lane] = acc;
    }
    if (term_on) {
        double tacc = 0.0;
#pragma unroll
        for (int q = 0; q < TPRE; ++q) tacc += ypre[q];  // ascending term index within the slice
        if (rowthread) red_term[tid >> 4][tid & 15] = tacc;
    }
    if (near_on || term_on) __syncthreads();
    HC_MARK(sc, 6);

    double rad = 0.0, wav = 0.0;
    if (a.do_rad) {
        if (a.nchunks_rad > 0) rad = lane16_sum(lane_sum(0, a.nchunks_rad));
        if (a.P) rad = p_row + rad;
        if (a.n_near_slices > 0)  // own-sample part of a wide system, computed by near_split_kernel: slice `sub`, then the fixed xor tree
        {
            double part = 0.0;
            if (sub < a.n_near_slices) {
                const double* q = a.near_partials + (size_t)sub * a.Dpad + rrow;
                if constexpr (COHERENT) part = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else part = *q;
            }
            rad += lane16_sum(part);
        }
        if (term_on) {
            double ts = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) ts += red_term[q][rit];  // fixed order
            rad += ts;
        }
        if (near_on) {
            double ns_ = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) ns_ += red_near[w][rit];  // fixed order
            rad += ns_;
        }
    }
    if (a.do_waves && a.wave_mode == 2) wav = a.E ? e_row : lane16_sum(lane_sum(a.nchunks_rad, a.nchunks_ex));
    if (a.do_waves && a.wave_mode == 3) {
        // component sum over the spectrum (the north_star's literal wording; not the reference's IRF convolution):
        // eta(t) = sum a_i cos(w_i t - phi_i)  ->  f = sum |X(w_i)| a_i cos(w_i t - phi_i + arg X(w_i)); 16 lanes per row
        const double* __restrict__ mg = a.spec_mag + (size_t)rrow * a.spec_nf;
        const double* __restrict__ pg = a.spec_phase + (size_t)rrow * a.spec_nf;
        for (int k = sub; k < a.spec_nf; k += 16) wav += mg[k] * a.spec_amp[k] * cos(a.spec_omega[k] * a.t - a.spec_phi[k] + pg[k]);
        wav = lane16_sum(wav);
        if (a.spec_ramp > 0.0 && a.t < a.spec_ramp) wav *= (a.t <= 0.0) ? 0.0 : a.t / a.spec_ramp;
    }
    if (!finisher) return;

    double hs = 0.0;
    if (a.do_waves && a.wave_mode == 1) {
        // RegularWave::GetForceAtTime (src/wave_types.cpp:315-327)
        wav = a.reg_mag[row] * a.reg_amplitude * cos(a.reg_omega * a.t + a.reg_phase[i]);
    }
    if (a.do_hs) {
        // ComputeForceHydrostatics (src/hydro_forces.cpp:263-322)
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) s += krow[j] * dq[j];
        const double glen = sqrt(a.gx * a.gx + a.gy * a.gy + a.gz * a.gz);
        hs                = -(a.rho * glen) * s;
        const double fbx = a.rho * (-a.gx) * V, fby = a.rho * (-a.gy) * V, fbz = a.rho * (-a.gz) * V;
        double add;
        switch (i) {
            case 0: add = fbx; break;
            case 1: add = fby; break;
            case 2: add = fbz; break;
            case 3: add = r[1] * fbz - r[2] * fby; break;
            case 4: add = r[2] * fbx - r[0] * fbz; break;
            default: add = r[0] * fby - r[1] * fbx; break;
        }
        hs += add;
    }
    const double total = hs - rad + wav;  // src/hydro_forces.cpp:758-760
#ifdef HC_TUNING
    asm volatile("" ::"v"(total));
#endif
    HC_MARK(sc, 7);
    a.hs[row]    = hs;
    a.rad[row]   = rad;
    a.waves[row] = wav;
    a.total[row] = total;
    if (a.user_out) a.user_out[row] = total;
    if (a.host_tagged) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u64x2*>(a.host_tagged + 2 * (size_t)row) = u64x2{(unsigned long long)__double_as_longlong(total), a.seq};
    }
#ifdef HC_TUNING
    if (HC_STAMPS(a)) {
        HC_MARK(sc, 8);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (stores count in vmcnt on gfx9: the totals have been acknowledged)
#endif
        HC_MARK(sc, 9);
        sc.store(HC_STAMPS(a));
    }
#endif
}

// SLOT (direct dispatch of hc_step for systems of up to kSlotStateMaxBodies bodies): the host has stored the body state behind the
// argument block, so its address is known from the kernarg segment pointer alone and the velocities every work-item stages are
// requested at once -- in flight together with the argument loads instead of one memory round trip behind them (the state and the
// arguments both live in fine-grained device memory the host writes through the BAR: uncached reads of about a microsecond each).
template <int NW, bool SLOT>
__global__ void __launch_bounds__(64 * NW) finalize_kernel(FinalizeArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* U = reinterpret_cast<double*>(smem_raw);  // [n_near][D] right-hand sides of the near samples
    __shared__ double red_near[NW][16];
    __shared__ double red_term[16][16];  // [term slice][row]
    const double* __restrict__ st = nullptr;
    double ev0 = 0.0, ev1 = 0.0, ev2 = 0.0, ev3 = 0.0;
    StageClock sc;
    HC_MARK(sc, 0);
    if constexpr (SLOT) {
#if defined(__HIP_DEVICE_COMPILE__)
        st = (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kSlotArgBytes);
#endif
        ev0 = st[threadIdx.x];  // (always inside the slot's kSlotStateDoubles, whatever D is)
        ev1 = st[threadIdx.x + 256];
        ev2 = st[threadIdx.x + 512];
        ev3 = st[threadIdx.x + 768];
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);  // (the requests go out BEFORE the first wait for an argument, not behind it)
#endif
    }
    touch_args<sizeof(FinalizeArgs)>();
    HC_MARK(sc, 1);
    if (a.do_push && (int)blockIdx.x == a.nblocks - 1) {  // (a.nblocks, not gridDim: the kernel takes no hidden arguments, see hc_direct.hpp)
        push_sample<NW, SLOT>(a, st);
#ifdef HC_TUNING
        if (HC_STAMPS(a)) {  // the workgroup that stores the sample: entry, arguments, stores issued, stores acknowledged
            HC_MARK(sc, 8);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HC_MARK(sc, 9);
            sc.store(HC_STAMPS(a));
        }
#endif
        return;
    }
    finalize_tile<NW, false, SLOT>(a, (int)blockIdx.x, U, red_near, red_term, [] { return true; }, sc, st, ev0, ev1, ev2, ev3);
}
template __global__ void finalize_kernel<4, true>(FinalizeArgs);

#ifdef HC_TUNING  // (measured in round 6 and not taken, EXPERIMENTS.md: the tuning build keeps it selectable with HC_STEP_PRELOAD=1)
// finalize_pre_kernel: finalize_kernel<4, true> whose first dependent hop is gone.  The step kernel's chain is  arguments -> K words /
// scatter results -> right-hand side -> contraction -> totals: two memory round trips before the first multiply, the first of them to
// uncached memory (the argument slot the host has just written through the BAR).  The ten leading kernel arguments below are
// PRELOADED -- the packet processor reads them from the argument block while it sets the dispatch up and the waves start with them in
// scalar registers (-mllvm -amdgpu-kernarg-preload-count, kernel descriptor field kernarg_preload_length; a firmware that does not
// preload runs the compiler's compatibility prologue, which loads them first) -- so the K words of the wave, the scatter results and
// the body state are all requested by the kernel's first instructions, together with the rest of the argument block:
//   kfirst = K's panel base + the first column group of the step's own IRF sample (near[0]); ngroups = its column groups (0: none);
//   yc / n_terms / dpad = FinalizeArgs::Yc, n_terms (0 when the step has no radiation term), Dpad; ngp = row-tile stride in groups;
//   ntiles = row tiles of the context (the workgroup behind them stores the sample and requests nothing).
template <int NW>
__global__ void __launch_bounds__(64 * NW) finalize_pre_kernel(const double* __restrict__ kfirst, const double* __restrict__ yc, int ngp, int ngroups, int n_terms,
                                                                int dpad, int ntiles, int pad_, FinalizeArgs a) {
    static_assert(NW == 4, "the early loads assume 256 work-items");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* U = reinterpret_cast<double*>(smem_raw);
    __shared__ double red_near[NW][16];
    __shared__ double red_term[16][16];
    StageClock sc;
    HC_MARK(sc, 0);
    const double* __restrict__ st = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    st = (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kSlotArgBytes);
#endif
    const double ev0 = st[threadIdx.x], ev1 = st[threadIdx.x + 256], ev2 = st[threadIdx.x + 512], ev3 = st[threadIdx.x + 768];
    EarlyLoads el;
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = (int)blockIdx.x;
        const bool tile_wg = tile < ntiles;
        const double* __restrict__ kb = kfirst + ((size_t)(tile_wg ? tile : 0) * ngp) * 128 + lane * 2;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int gp = wave + NW * q;
            el.pre[q]    = (tile_wg && gp < ngroups) ? *reinterpret_cast<const dvec2*>(kb + (size_t)gp * 128) : dvec2{0.0, 0.0};
        }
        const double* __restrict__ y = yc + (tile_wg ? tile : 0) * 16 + (tid & 15);
#pragma unroll
        for (int q = 0; q < kTermMax / 16; ++q) {
            const int k = (tid >> 4) + 16 * q;
            el.ypre[q]  = (tile_wg && k < n_terms) ? y[(size_t)k * dpad] : 0.0;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);  // (all of the above goes out BEFORE the first wait for an argument)
#endif
    touch_args<sizeof(FinalizeArgs) + 40>();
    HC_MARK(sc, 1);
    (void)pad_;
    if (a.do_push && (int)blockIdx.x == a.nblocks - 1) {
        push_sample<NW, true>(a, st);
#ifdef HC_TUNING
        if (HC_STAMPS(a)) {
            HC_MARK(sc, 8);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HC_MARK(sc, 9);
            sc.store(HC_STAMPS(a));
        }
#endif
        return;
    }
    finalize_tile<NW, false, true, true>(a, (int)blockIdx.x, U, red_near, red_term, [] { return true; }, sc, st, ev0, ev1, ev2, ev3, &el);
}
template __global__ void finalize_pre_kernel<4>(const double*, const double*, int, int, int, int, int, int, FinalizeArgs);
#endif

// ------------------------------------------------------------------------------------------------
// step_hot_kernel<NE>: the step kernel of the common block step (StepHotArgs, hc_kernels.hpp), written around what the stage clock of
// finalize_kernel<4, true> showed (profiles/r06/step_stage_clock*.txt).  First pass: of the 4.8 us a tile workgroup
// lived, 0.3 were the argument block, 1.2 the one memory round trip the step needs -- and 3.3 were round trips the CODE made one after
// the other (scalar loads of single arguments with a wait each, a wait for the positions in front of the K requests, a full drain in
// the middle of the term loads).  Second pass, on the first form of this kernel (4.0 us): halving the bytes per compute unit changed
// nothing -- a workgroup is ONE wave per SIMD, nothing hides a latency, and what was left were requests (16 cycles of the workgroup's
// one address unit each, whatever their width and mask) and dependent operations (24 LDS round trips inside the product loop).  Here
//   * every argument is a scalar register after ONE wait (the compact block is requested whole, then pinned: no argument is loaded twice);
//   * every global load is an asm request in program order, waited for by hand; the ones that are not always made (tables: the 16
//     finishing lanes of wave 0 only; scatter results: only as far as the step has terms) carry their condition INSIDE the statement;
//   * the right-hand sides sit in LDS between zero pads and are read in one batch; the products are a straight line;
//   * the hydrostatic / wave terms of the finishing lanes are formed between those LDS requests and the wait for the K words;
//   * nothing else changes: the same products and sums in the same order as finalize_kernel (bitwise the same forces; the tuning
//     build keeps the A/B switch HC_STEP_HOT, tests/test_gpu_boundary.py, 14 shapes).
// 2.5 us per workgroup.  One workgroup per tile of 16 rows + one that stores the sample; 256 work-items; dynamic LDS NE * (D + 16) doubles.
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define HC_PIN_S(x) asm volatile("" : "+s"(x))
#else
#define HC_PIN_S(x) ((void)0)
#endif
// Loads the compiler neither moves nor counts: the step kernel's requests go out in exactly this order, at exactly this place, and
// are waited for by hand (hot_wait<N>: at most N of them still in flight, oldest first; hot_pin ties the values to the wait).  The
// optimiser would otherwise sink a load into the branch that uses it -- behind a barrier, one memory round trip later.
__device__ __forceinline__ void hot_ld(double& d, const double* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
#else
    d = *p;
#endif
}
__device__ __forceinline__ void hot_ld2(dvec2& d, const double* p) {  // 16 bytes at 8-byte alignment
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
#else
    d = dvec2{p[0], p[1]};
#endif
}
// ... and their scalar-base forms: address = base (scalar register pair) + byte offset (one 32-bit vector register) + immediate
template <int IMM = 0>
__device__ __forceinline__ void hot_lds(double& d, const double* sbase, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
#else
    d = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + voff + IMM);
#endif
}
template <int IMM = 0>
__device__ __forceinline__ void hot_lds2(dvec2& d, const double* sbase, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
#else
    const double* q = reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + voff + IMM);
    d               = dvec2{q[0], q[1]};
#endif
}
template <int N>
__device__ __forceinline__ void hot_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <class T>
__device__ __forceinline__ void hot_pin(T& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}
// Requests that are not always made must not become branches the compiler can see: a value that is "either 0 or what the load will
// bring" is a register copy waiting to be placed in front of the wait (it happened: the release build of one revision copied a scatter
// result before it had arrived -- the sphere-decay golden caught it, the tuning build of the same source was fine).  So the condition
// lives INSIDE the statement: for the compiler the register is written either way, and what it holds when the request was skipped is
// masked where it is used.
// hot_lds_if_at_least<LIMIT>: the request goes out if n >= LIMIT (uniform).
template <int LIMIT>
__device__ __forceinline__ void hot_lds_if_at_least(double& d, const double* sbase, unsigned voff, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_cmp_lt_i32 %3, %4\n\ts_cbranch_scc1 .Lhot_skip_%=\n\tglobal_load_dwordx2 %0, %1, %2\n.Lhot_skip_%=:"
                 : "=v"(d)
                 : "v"(voff), "s"(sbase), "s"(n), "n"(LIMIT)
                 : "memory", "scc");
#else
    if (n >= LIMIT) d = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + voff);
#endif
}
// hot_tables_a / _b: what the finishing lanes (mask) need besides the sums -- state and tables of the row's body, the row's look-ahead
// and excitation values -- requested by those lanes only, and by no lane of a wave that has none: the execution mask is narrowed,
// the requests made and the mask restored INSIDE one statement each (nothing the compiler schedules can land under the narrow mask).
__device__ __forceinline__ void hot_tables_a(dvec2& pos01, double& pos2, dvec2& rpy01, double& rpy2, dvec2& cg01, double& cg2, dvec2& r01, double& r2,
                                             const double* st_pos, const double* st_rpy, const double* cg, const double* cbm, unsigned b3_off, unsigned bl3_off,
                                             unsigned long long mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long sv;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "s_cbranch_execz .Lhot_ta_%=\n\t"
                 "global_load_dwordx4 %[o0], %[b3], %[ppos]\n\t"
                 "global_load_dwordx2 %[o1], %[b3], %[ppos] offset:16\n\t"
                 "global_load_dwordx4 %[o2], %[b3], %[prpy]\n\t"
                 "global_load_dwordx2 %[o3], %[b3], %[prpy] offset:16\n\t"
                 "global_load_dwordx4 %[o4], %[bl3], %[pcg]\n\t"
                 "global_load_dwordx2 %[o5], %[bl3], %[pcg] offset:16\n\t"
                 "global_load_dwordx4 %[o6], %[bl3], %[pcb]\n\t"
                 "global_load_dwordx2 %[o7], %[bl3], %[pcb] offset:16\n"
                 ".Lhot_ta_%=:\n\t"
                 "s_or_b64 exec, exec, %[sv]"
                 : [sv] "=&s"(sv), [o0] "=&v"(pos01), [o1] "=&v"(pos2), [o2] "=&v"(rpy01), [o3] "=&v"(rpy2), [o4] "=&v"(cg01), [o5] "=&v"(cg2), [o6] "=&v"(r01),
                   [o7] "=&v"(r2)
                 : [m] "s"(mask), [b3] "v"(b3_off), [bl3] "v"(bl3_off), [ppos] "s"(st_pos), [prpy] "s"(st_rpy), [pcg] "s"(cg), [pcb] "s"(cbm)
                 : "memory", "scc");
#else
    (void)mask;
    const char *a = reinterpret_cast<const char*>(st_pos) + b3_off, *b = reinterpret_cast<const char*>(st_rpy) + b3_off;
    const char *c = reinterpret_cast<const char*>(cg) + bl3_off, *d = reinterpret_cast<const char*>(cbm) + bl3_off;
    auto at = [](const char* q, int k) { return reinterpret_cast<const double*>(q)[k]; };
    pos01 = dvec2{at(a, 0), at(a, 1)}; pos2 = at(a, 2); rpy01 = dvec2{at(b, 0), at(b, 1)}; rpy2 = at(b, 2);
    cg01 = dvec2{at(c, 0), at(c, 1)}; cg2 = at(c, 2); r01 = dvec2{at(d, 0), at(d, 1)}; r2 = at(d, 2);
#endif
}
__device__ __forceinline__ void hot_tables_b(dvec2& k01, dvec2& k23, dvec2& k45, double& V, double& rmag, double& p_row, double& e_raw, const double* lin,
                                             const double* vol, const double* regm, const double* P, const double* E, unsigned lin_off, unsigned bl_off,
                                             unsigned row_off, unsigned long long mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long sv;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "s_cbranch_execz .Lhot_tb_%=\n\t"
                 "global_load_dwordx4 %[o0], %[lo], %[plin]\n\t"
                 "global_load_dwordx4 %[o1], %[lo], %[plin] offset:16\n\t"
                 "global_load_dwordx4 %[o2], %[lo], %[plin] offset:32\n\t"
                 "global_load_dwordx2 %[o3], %[bo], %[pvol]\n\t"
                 "global_load_dwordx2 %[o4], %[ro], %[prm]\n\t"
                 "global_load_dwordx2 %[o5], %[ro], %[pp]\n\t"
                 "global_load_dwordx2 %[o6], %[ro], %[pe]\n"
                 ".Lhot_tb_%=:\n\t"
                 "s_or_b64 exec, exec, %[sv]"
                 : [sv] "=&s"(sv), [o0] "=&v"(k01), [o1] "=&v"(k23), [o2] "=&v"(k45), [o3] "=&v"(V), [o4] "=&v"(rmag), [o5] "=&v"(p_row), [o6] "=&v"(e_raw)
                 : [m] "s"(mask), [lo] "v"(lin_off), [bo] "v"(bl_off), [ro] "v"(row_off), [plin] "s"(lin), [pvol] "s"(vol), [prm] "s"(regm), [pp] "s"(P), [pe] "s"(E)
                 : "memory", "scc");
#else
    (void)mask;
    const double* l = reinterpret_cast<const double*>(reinterpret_cast<const char*>(lin) + lin_off);
    k01 = dvec2{l[0], l[1]}; k23 = dvec2{l[2], l[3]}; k45 = dvec2{l[4], l[5]};
    V     = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(vol) + bl_off);
    rmag  = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(regm) + row_off);
    p_row = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(P) + row_off);
    e_raw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(E) + row_off);
#endif
}

// HOT_MARK: a stamp of the stage clock in the tuning build; in the shipped build the same scheduling fence WITHOUT the stamp, at the same
// places -- the requests and waits of this kernel are asm statements the compiler schedules around, and the build that is compared
// bitwise with the general kernel (the tuning build, HC_STEP_HOT) should be the instruction order that ships.
#ifdef HC_TUNING
#define HOT_MARK(sc, K) HC_MARK(sc, K)
#elif defined(__HIP_DEVICE_COMPILE__)
#define HOT_MARK(sc, K) __builtin_amdgcn_sched_barrier(0)
#else
#define HOT_MARK(sc, K) ((void)0)
#endif

template <int NE>
__global__ void __launch_bounds__(256) step_hot_kernel(StepHotArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* U = reinterpret_cast<double*>(smem_raw);  // [NE][D]
    __shared__ double red_near[4][16];
    __shared__ double red_term[16][16];
    StageClock sc;
    (void)sc;
    HOT_MARK(sc, 0);
    const double* __restrict__ st = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    st = (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kSlotArgBytes);
#endif
    const int tid = threadIdx.x;
    double ev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hot_ld(ev[k], st + tid + 256 * k);  // (inside the slot whatever D is)
    // ---- the arguments the loads need: the first part of the block, whole, ONE wait ----
    const double* kf0 = a.kfirst[0];
    const double* kf1 = a.kfirst[NE - 1];
    const double *Yc = a.Yc, *P = a.P, *E = a.E, *lin = a.lin, *cg = a.cg, *cbm = a.cb_m_cg, *vol = a.disp_vol, *regm = a.reg_mag;
    int ngp = a.ngp, ng0 = a.ng[0], ng1 = a.ng[NE - 1], n_terms = a.n_terms, Dpad = a.Dpad, Dloc = a.Dloc, N = a.N, b0 = a.b0, ntiles = a.ntiles, halves = a.halves;
    HC_PIN_S(kf0); HC_PIN_S(kf1); HC_PIN_S(Yc); HC_PIN_S(P); HC_PIN_S(E); HC_PIN_S(lin); HC_PIN_S(cg); HC_PIN_S(cbm); HC_PIN_S(vol); HC_PIN_S(regm);
    HC_PIN_S(ngp); HC_PIN_S(ng0); HC_PIN_S(ng1); HC_PIN_S(n_terms); HC_PIN_S(Dpad); HC_PIN_S(Dloc); HC_PIN_S(N); HC_PIN_S(b0); HC_PIN_S(ntiles); HC_PIN_S(halves);
    HOT_MARK(sc, 1);

    if ((int)blockIdx.x >= ntiles) {
        // ---- the workgroup that hands back the state canary and stores this step's sample into ring slot `head` (push_sample) ----
        const int D = a.D;
        const double t = a.t;
        const unsigned long long seq = a.seq;
        unsigned long long *canary_out = a.canary_out, *stamps = a.stamps;
        (void)stamps;
        double word;
        hot_ld(word, st + 12 * N);  // (requested whether or not it is handed back: no branch around a request, see hot_lds_if_at_least)
        hot_wait<0>();
#pragma unroll
        for (int k = 0; k < 4; ++k) hot_pin(ev[k]);
        hot_pin(word);
        if (tid == 64 && canary_out) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u64x2*>(canary_out) = u64x2{(unsigned long long)__double_as_longlong(word), seq};
        }
        const int head = a.head, Hcap = a.Hcap, HcapT = a.HcapT;
        if (tid == 0) a.ring_t[head] = t;
        double* slot = a.ring_v + (size_t)head * D;
        double* vT   = a.ring_vT;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tid + 256 * k;
            if (c < D) {
                slot[c] = ev[k];
                vT[(size_t)c * HcapT + head] = ev[k];
                if (head == 0) vT[(size_t)c * HcapT + Hcap] = ev[k];
            }
        }
#ifdef HC_TUNING
        if (stamps) {
            HOT_MARK(sc, 8);
            hot_wait<0>();
            HOT_MARK(sc, 9);
            sc.store(stamps);
        }
#endif
        return;
    }

    // rows are FINISHED by the first 16 lanes of wave 0 (lane r: row r of the tile): they alone request the tables, form the hydrostatic
    // and wave terms and store the row -- one wave's 15 requests instead of four waves' (a request costs the workgroup's one address
    // unit 16 cycles whatever its width and however many lanes are live: 39 requests per wave were the 1.1 us in front of the K words)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), rit = tid & 15, kk = lane >> 4;
    const bool fin = tid < 16;
    // halves == 2 (tuning experiment HC_STEP_HALVES, EXPERIMENTS.md round 6; the shipped library always passes 1): two workgroups per tile
    // of 16 rows, each finishing 8 of them.  The lanes of the other half's rows ask for THIS half's words a second time (the same cache
    // lines) and compute them a second time; every sum keeps its order.  Measured: the bytes per compute unit were not the bound.
    const int wg = (int)blockIdx.x, tile = halves == 2 ? wg >> 1 : wg, half = halves == 2 ? wg & 1 : 0;
    const int row = tile * 16 + rit;
    const bool live = row < Dloc && (halves != 2 || (rit >> 3) == half);
    const int row_h = tile * 16 + 8 * half, rrow = live ? row : (row_h < Dloc ? row_h : 0), bl = rrow / 6, i = rrow - 6 * bl, b = b0 + bl;
    const int lane_k = halves == 2 ? ((lane & ~8) | (half << 3)) : lane;  // the lane whose K words this lane asks for
    // ---- the requests of the step, in the order their values are needed: the finishing lanes' tables, the scatter results, the K words ----
    constexpr int PRE  = 12;  // C3: all 12 column groups a wave owns of one IRF sample
    constexpr int TPRE = kTermMax / 16;
    constexpr int kLoadsBehindTerms1 = NE * PRE;  // the K words: requested last, waited for last
    double p_row, e_raw, V, rmag, pos2, rpy2, cg2, r2;
    dvec2 pos01, rpy01, cg01, r01, k01, k23, k45;
    {
        const unsigned row_off = (unsigned)rrow * 8u, b3_off = (unsigned)b * 24u, bl3_off = (unsigned)bl * 24u, lin_off = (unsigned)(36 * bl + 6 * i) * 8u,
                       bl_off = (unsigned)bl * 8u;
        const unsigned long long finishing = wave == 0 ? 0xFFFFull : 0ull;  // (in every other lane these registers keep what they held: never used there)
        hot_tables_a(pos01, pos2, rpy01, rpy2, cg01, cg2, r01, r2, st + 6 * N, st + 9 * N, cg, cbm, b3_off, bl3_off, finishing);
        hot_tables_b(k01, k23, k45, V, rmag, p_row, e_raw, lin, vol, regm, P, E, lin_off, bl_off, row_off, finishing);
    }
    // ... the term slots tid >> 4, + 16, ... of the step only as far as the step has terms (16 on average, 192 slots: one or two requests
    // instead of twelve; requested in front of the K words, whose number is fixed, so that the first wait below still counts exactly) ...
    double ypre[TPRE];
    {
        const unsigned y_off = ((unsigned)(tid & 15) + (unsigned)(tid >> 4) * (unsigned)Dpad) * 8u;
        const double* __restrict__ y = Yc + tile * 16;
        hot_lds_if_at_least<1>(ypre[0], y, y_off, n_terms);  // (slots past n_terms inside a group of 16 exist; all are masked where they are used)
        hot_lds_if_at_least<17>(ypre[1], y + (size_t)16 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<33>(ypre[2], y + (size_t)32 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<49>(ypre[3], y + (size_t)48 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<65>(ypre[4], y + (size_t)64 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<81>(ypre[5], y + (size_t)80 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<97>(ypre[6], y + (size_t)96 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<113>(ypre[7], y + (size_t)112 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<129>(ypre[8], y + (size_t)128 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<145>(ypre[9], y + (size_t)144 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<161>(ypre[10], y + (size_t)160 * Dpad, y_off, n_terms);
        hot_lds_if_at_least<177>(ypre[11], y + (size_t)176 * Dpad, y_off, n_terms);
        static_assert(TPRE == 12, "one request per group of 16 term slots");
    }
    dvec2 pre[NE][PRE];
    {
        // K words: the wave's column groups wave, wave + 4, ... of own sample e -- one scalar base per load, the lane's 16 bytes as the offset
        const unsigned lane_off = (unsigned)lane_k * 16u;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double* __restrict__ kb = (e == 0 ? kf0 : kf1) + ((size_t)tile * ngp) * 128;
            const int ng = e == 0 ? ng0 : ng1;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int gp = wave + 4 * q;
                hot_lds2(pre[e][q], kb + (size_t)(gp < ng ? gp : 0) * 128, lane_off);
            }
        }
    }
    HOT_MARK(sc, 2);
    // ---- the rest of the arguments, while the loads are in flight (read through a pointer the compiler cannot see through, so that
    //      these requests are not hoisted in front of the loads above, where they would compete for the scalar registers) ----
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) StepHotArgs* HotArgsPtr;  // (constant address space: scalar loads)
    HotArgsPtr a2 = (HotArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    HC_PIN_S(a2);
#else
    const StepHotArgs* a2 = &a;
#endif
    int off0 = a2->off[0], off1 = a2->off[NE - 1], D = a2->D, wave_mode = a2->wave_mode, has_E = a2->has_E;
    double a0 = a2->a[0], a1 = a2->a[NE - 1];
    double rho = a2->rho, gx = a2->gx, gy = a2->gy, gz = a2->gz, t = a2->t, reg_amp = a2->reg_amplitude, reg_omega = a2->reg_omega;
    double ph0 = a2->reg_phase[0], ph1 = a2->reg_phase[1], ph2 = a2->reg_phase[2], ph3 = a2->reg_phase[3], ph4 = a2->reg_phase[4], ph5 = a2->reg_phase[5];
    unsigned long long seq = a2->seq;
    double *o_hs = a2->hs, *o_rad = a2->rad, *o_waves = a2->waves, *o_total = a2->total;
    unsigned long long *tagged = a2->host_tagged, *stamps = a2->stamps;
    (void)stamps;
    HC_PIN_S(off0); HC_PIN_S(off1); HC_PIN_S(D); HC_PIN_S(wave_mode); HC_PIN_S(has_E); HC_PIN_S(a0); HC_PIN_S(a1);
    HC_PIN_S(rho); HC_PIN_S(gx); HC_PIN_S(gy); HC_PIN_S(gz); HC_PIN_S(t); HC_PIN_S(reg_amp); HC_PIN_S(reg_omega);
    HC_PIN_S(ph0); HC_PIN_S(ph1); HC_PIN_S(ph2); HC_PIN_S(ph3); HC_PIN_S(ph4); HC_PIN_S(ph5); HC_PIN_S(seq);
    HC_PIN_S(o_hs); HC_PIN_S(o_rad); HC_PIN_S(o_waves); HC_PIN_S(o_total); HC_PIN_S(tagged); HC_PIN_S(stamps);

    // ---- state and tables are in (requested first): right-hand sides of the own samples u = a_e * v_state into LDS, and the
    //      hydrostatic / wave terms of the row while the K words are still in flight ----
    hot_wait<kLoadsBehindTerms1>();
#pragma unroll
    for (int k = 0; k < 4; ++k) hot_pin(ev[k]);
    hot_pin(p_row); hot_pin(e_raw); hot_pin(V); hot_pin(rmag); hot_pin(pos2); hot_pin(rpy2); hot_pin(cg2); hot_pin(r2);
    hot_pin(pos01); hot_pin(rpy01); hot_pin(cg01); hot_pin(r01); hot_pin(k01); hot_pin(k23); hot_pin(k45);
    // (LDS: per own sample 8 zeros, the D right-hand-side values, 8 zeros -- a lane column in front of the sample's first or behind its last
    //  reads a zero instead of being masked: U_e = U + 8 + e * (D + 16))
    const int Dp = D + 16;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const double ae = e == 0 ? a0 : a1;
        double* __restrict__ ue = U + 8 + e * Dp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = tid + 256 * k;
            if (col < D) ue[col] = (ae != 0.0) ? ae * ev[k] : 0.0;
        }
        if (tid >= 240) ue[tid < 248 ? tid - 248 : D + tid - 248] = 0.0;  // (a wave whose lanes store no column of a 64-body system)
    }
    HOT_MARK(sc, 3);
    __syncthreads();
    HOT_MARK(sc, 4);
    // Every right-hand-side value the preloaded K words will meet, requested from LDS AT ONCE: written as `in range ? u[c] : 0` inside
    // the product loop these were 24 LDS round trips one after the other, each with its own wait -- 1.3 us of the workgroup's 4 (stage
    // clock: "contraction done"), more than the K words themselves took.  all_in[e]: every preloaded column group of this wave belongs
    // to own sample e (a system of 62 bodies or more) -- then the 24 addresses are one register + constants, and the products below a
    // straight line of 24 fused multiply-adds.
    bool all_in[NE];
    double u0v[NE][PRE], u1v[NE][PRE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int ng = e == 0 ? ng0 : ng1, off = e == 0 ? off0 : off1;
        all_in[e] = wave + 4 * (PRE - 1) < ng;
        const double* __restrict__ u = U + 8 + e * Dp;  // column c of the sample = lane column 8 * gp + kk (+ 4) - off, groups counted from the sample's first
        if (all_in[e]) {
            const double* __restrict__ ul = u + (wave * 8 + kk - off);
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                u0v[e][q] = ul[32 * q];
                u1v[e][q] = ul[32 * q + 4];
            }
        } else {
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int ca = (wave + 4 * q) * 8 + kk - off;  // (groups past the sample's last: any valid address, the value is not used)
                u0v[e][q] = u[min(ca, D + 3)];
                u1v[e][q] = u[min(ca, D + 3) + 4];
            }
        }
    }
    // ... and the hydrostatic and wave terms of the row formed in their shadow and in that of the K words
    double wav = 0.0, hs = 0.0;
    if (fin) {
        if (wave_mode == 2) wav = has_E ? e_raw : 0.0;
        if (wave_mode == 1) {
            const double ph = i == 0 ? ph0 : (i == 1 ? ph1 : (i == 2 ? ph2 : (i == 3 ? ph3 : (i == 4 ? ph4 : ph5))));
            wav = rmag * reg_amp * cos(reg_omega * t + ph);  // RegularWave::GetForceAtTime (src/wave_types.cpp:315-327)
        }
        // ComputeForceHydrostatics (src/hydro_forces.cpp:263-322)
        const double posv[3] = {pos01.x, pos01.y, pos2}, rpyv[3] = {rpy01.x, rpy01.y, rpy2}, cgv[3] = {cg01.x, cg01.y, cg2}, r[3] = {r01.x, r01.y, r2};
        const double krow[6] = {k01.x, k01.y, k23.x, k23.y, k45.x, k45.y};
        double dq[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            dq[j]     = posv[j] - cgv[j];
            dq[3 + j] = rpyv[j] - 0.0;
        }
        double ssum = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) ssum += krow[j] * dq[j];
        const double glen = sqrt(gx * gx + gy * gy + gz * gz);
        hs                = -(rho * glen) * ssum;
        const double fbx = rho * (-gx) * V, fby = rho * (-gy) * V, fbz = rho * (-gz) * V;
        double add;
        switch (i) {
            case 0: add = fbx; break;
            case 1: add = fby; break;
            case 2: add = fbz; break;
            case 3: add = r[1] * fbz - r[2] * fby; break;
            case 4: add = r[2] * fbx - r[0] * fbz; break;
            default: add = r[0] * fby - r[1] * fbx; break;
        }
        hs += add;
    }
    hot_pin(hs);
    hot_pin(wav);
    HOT_MARK(sc, 10);  // (hydrostatic / wave terms formed)
    hot_wait<0>();
    HOT_MARK(sc, 11);  // (every K word of this wave is in)
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int q = 0; q < PRE; ++q) hot_pin(pre[e][q]);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int ng = e == 0 ? ng0 : ng1, off = e == 0 ? off0 : off1;
        const double* __restrict__ u = U + 8 + e * Dp;
        int gp = wave + 4 * PRE;
        if (all_in[e]) {
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                acc = fma(pre[e][q].x, u0v[e][q], acc);
                acc = fma(pre[e][q].y, u1v[e][q], acc);
            }
        } else {
            const int nq = max(0, (ng - wave + 3) >> 2);  // the wave's groups inside the sample: a prefix (fewer than PRE here)
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const double t = fma(pre[e][q].y, u1v[e][q], fma(pre[e][q].x, u0v[e][q], acc));
                acc            = q < nq ? t : acc;
            }
        }
        const double* __restrict__ kb = (e == 0 ? kf0 : kf1) + ((size_t)tile * ngp) * 128 + lane_k * 2;
#pragma unroll 4
        for (; gp < ng; gp += 4) {  // (more than 384 columns)
            const dvec2 kv = *reinterpret_cast<const dvec2*>(kb + (size_t)gp * 128);
            const int ca = gp * 8 + kk - off;
            acc = fma(kv.x, u[ca], acc);
            acc = fma(kv.y, u[ca + 4], acc);
        }
    }
    acc += __shfl_xor(acc, 16, kWave);
    acc += __shfl_xor(acc, 32, kWave);
#ifdef HC_TUNING
    asm volatile("" ::"v"(acc));
#endif
    HOT_MARK(sc, 5);
    if (lane < 16) red_near[wave][lane] = acc;
    {
#pragma unroll
        for (int q = 0; q < TPRE; ++q) hot_pin(ypre[q]);
        double tacc = 0.0;
#pragma unroll
        for (int q = 0; q < TPRE; ++q) tacc += ((tid >> 4) + 16 * q < n_terms) ? ypre[q] : 0.0;  // ascending term index within the slice
        red_term[tid >> 4][tid & 15] = tacc;
    }
    __syncthreads();
    HOT_MARK(sc, 6);
#ifdef HC_TUNING
    if (stamps && tid == 0 && !live) {  // (halves == 2: the second workgroup of a tile finishes rows 8 .. 15; its clock row all the same)
        HOT_MARK(sc, 7);
        HOT_MARK(sc, 8);
        HOT_MARK(sc, 9);
        sc.store(stamps);
    }
#endif
    if (!(live && fin)) return;

    double rad = p_row + 0.0;
    if (n_terms > 0) {
        double ts = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) ts += red_term[q][rit];
        rad += ts;
    }
    {
        double ns_ = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) ns_ += red_near[w][rit];
        rad += ns_;
    }
    const double total = hs - rad + wav;  // src/hydro_forces.cpp:758-760
#ifdef HC_TUNING
    asm volatile("" ::"v"(total));
#endif
    HOT_MARK(sc, 7);
    o_hs[row]    = hs;
    o_rad[row]   = rad;
    o_waves[row] = wav;
    o_total[row] = total;
    if (tagged) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u64x2*>(tagged + 2 * (size_t)row) = u64x2{(unsigned long long)__double_as_longlong(total), seq};
    }
#ifdef HC_TUNING
    if (stamps) {
        HOT_MARK(sc, 8);
        hot_wait<0>();
        HOT_MARK(sc, 9);
        sc.store(stamps);
    }
#endif
}
template __global__ void step_hot_kernel<1>(StepHotArgs);
template __global__ void step_hot_kernel<2>(StepHotArgs);

FinalizeLaunch finalize_launch_config(FinalizeArgs& a) {
    FinalizeLaunch l;
    l.smem    = (size_t)max(0, a.n_near) * a.D * sizeof(double);
    l.grid    = (a.Dloc + 15) / 16 + (a.do_push ? 1 : 0);
    l.threads = 256;
    a.nblocks = l.grid;
    return l;
}

void launch_finalize(const FinalizeArgs& a0, hipStream_t stream) {
    FinalizeArgs a         = a0;
    const FinalizeLaunch l = finalize_launch_config(a);
    if (l.smem > 64 * 1024) {
        static size_t granted = 0;
        if (l.smem > granted) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.smem);
            granted = l.smem;
        }
    }
    hipLaunchKernelGGL((finalize_kernel<4, false>), dim3(l.grid), dim3(256), l.smem, stream, a);
}

// ------------------------------------------------------------------------------------------------
// near_split_kernel: the own-sample part of a step inside a look-ahead block for WIDE systems.  The step kernel alone would stream
// n_near x 16 x D x 8 bytes per row tile from one workgroup (393 KB per sample at D = 3072: 24 workgroups on a 256-CU chip, 19 us);
// here workgroup (row tile, slice) contracts the slice's columns of every near sample and leaves one partial per row; the step
// kernel adds the slices in a fixed order.  Same right-hand side arithmetic as finalize_kernel's near part.
// ------------------------------------------------------------------------------------------------
// slice `sl` of row tile `rt` (the whole workgroup); U: [n_near][8 * gps_per_slice] doubles of LDS.  COHERENT: the partial is read by
// another workgroup of the same launch (agent-scope atomic store: written through to where every XCD sees it).
template <bool COHERENT>
__device__ __forceinline__ void near_slice(const NearArgs& a, const int rt, const int sl, double* U, double (*red)[16]) {
    const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = a.D, W = 8 * a.gps_per_slice;
    const double* __restrict__ kbase = a.K.base + ((size_t)rt * a.K.ngp) * 128 + lane * 2;
    // the K words of the first near sample (13 column groups per wave at 384-column slices) are requested before the right-hand side is
    // staged: the kernel is on the critical path of every block step of a wide system, and its two memory round trips then overlap
    constexpr int PRE = 13;
    dvec2 pre[PRE];
    int pg0 = 0, pg1 = 0;
    if (a.n_near > 0) {
        const int f0 = a.near[0].s * D, f1 = f0 + D;
        pg0 = (f0 >> 3) + sl * a.gps_per_slice;
        pg1 = min((f1 + 7) >> 3, pg0 + a.gps_per_slice);
    }
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int gp = pg0 + wave + 4 * q;
        pre[q] = gp < pg1 ? *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128) : dvec2{0.0, 0.0};
    }
    // stage u for the slice's columns of every near sample (columns outside the sample -> 0)
    for (int e = 0; e < a.n_near; ++e) {
        const NearEntry& ne = a.near[e];
        const int f0 = ne.s * D, f1 = f0 + D;
        const int fs = ((f0 >> 3) + sl * a.gps_per_slice) * 8;  // first column of the slice (group aligned)
        for (int i = tid; i < W; i += 256) {
            const int f = fs + i;
            double u = 0.0;
            if (f >= f0 && f < f1) {
                const int col = f - f0;
                if (ne.a != 0.0) u = ne.a * state_velocity(a.state, a.N, col);
                if (ne.b != 0.0) u = fma(ne.b, a.ring_v[ne.off_b + col], u);
                if (ne.c != 0.0) u = fma(ne.c, a.ring_v[ne.off_c + col], u);
            }
            U[e * W + i] = u;
        }
    }
    __syncthreads();
    // the right-hand-side values the preloaded K words meet, requested from LDS at once (read inside the product loop they were one LDS
    // round trip per column group, each with its own wait: step_hot_kernel has the measurement); nq: the wave's preloaded groups that
    // lie inside the slice -- a prefix; the products below are a straight line, the ones past it computed and dropped
    const int nq = min(PRE, max(0, (pg1 - pg0 - wave + 3) >> 2));
    double u0v[PRE], u1v[PRE];
    {
        const double* __restrict__ u = U - pg0 * 8;
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int gpc = max(min(pg0 + wave + 4 * q, pg1 - 1), pg0);  // (past the slice: any valid address, the value is not used)
            u0v[q] = u[gpc * 8 + kk];
            u1v[q] = u[gpc * 8 + 4 + kk];
        }
    }
    double acc = 0.0;
    for (int e = 0; e < a.n_near; ++e) {
        const int f0 = a.near[e].s * D, f1 = f0 + D;
        const int g0 = (f0 >> 3) + sl * a.gps_per_slice, g1 = min((f1 + 7) >> 3, g0 + a.gps_per_slice);
        const double* __restrict__ u = U + e * W - g0 * 8;
        int gp = g0 + wave;
        if (e == 0) {
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const double t = fma(pre[q].y, u1v[q], fma(pre[q].x, u0v[q], acc));
                acc            = q < nq ? t : acc;
            }
            gp += 4 * PRE;
        }
#pragma unroll 4
        for (; gp < g1; gp += 4) {
            const dvec2 kv = *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128);
            acc = fma(kv.x, u[gp * 8 + kk], acc);
            acc = fma(kv.y, u[gp * 8 + 4 + kk], acc);
        }
    }
    acc += __shfl_xor(acc, 16, kWave);
    acc += __shfl_xor(acc, 32, kWave);
    if (lane < 16) red[wave][lane] = acc;
    __syncthreads();
    if (tid < 16) {
        const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        double* q      = a.partials + (size_t)sl * a.Dpad + rt * 16 + tid;
        if constexpr (COHERENT) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *q = v;
    }
}

__global__ void __launch_bounds__(256) near_split_kernel(NearArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double red[4][16];
    touch_args<sizeof(NearArgs)>();
    const int rt = (int)blockIdx.x / a.n_slices, sl = (int)blockIdx.x - rt * a.n_slices;
    near_slice<false>(a, rt, sl, reinterpret_cast<double*>(smem_raw), red);
}

// ------------------------------------------------------------------------------------------------
// wide_step_kernel: near_split_kernel and the step kernel of a wide system in ONE launch.  Workgroup (row tile, slice) first requests
// what the tile's step kernel needs (its rows of P, its scatter terms, the hydrostatics inputs), contracts its column slice of the
// own-sample part, leaves the partial and counts itself in at the tile's counter; the workgroup that finds itself LAST there is the
// tile's step kernel (finalize_tile: slice partials in slice order, look-ahead part, scatter terms, hydrostatics, total, tagged store).
// Which workgroup that is does not matter -- the slices are added in a fixed order from memory -- so the forces are bitwise those of the
// two-launch form; what is saved is a dispatch (packet, barrier, the 4-5 us floor of a kernel) on the critical path of every block step
// of a wide system.
// The hand-off crosses XCDs (their L2s are not coherent with each other), and an agent-scope FENCE costs an L2 write-back and
// invalidate per workgroup (33 us per step when it was tried, profiles/r04/ab_wide_kernels.txt).  So no fence: the 16 partials of a
// slice are agent-scope atomic stores (global_store sc1: written through), complete before the count (s_waitcnt vmcnt(0)); the count is
// an agent-scope atomic add; the last workgroup reads the partials with agent-scope atomic loads (global_load sc1) issued after its own
// add has returned.  Nothing else the step kernel reads was written in this launch.
// This is the fence-free hand-off MI355X_MICROARCH.md lists as measured on gfx950 ("ONE lane of each storing workgroup, for ALL that
// workgroup's stores: an agent-scope atomic add ... the workgroup whose add came last, told by the value its add returned"; stores and
// loads all sc1, every storing wave drained, a workgroup barrier on both sides) -- not an architectural guarantee of the memory model.
// The library runs on gfx950 only (hc_create refuses every other device), the suite keeps the bitwise A/B against the two-launch form
// with its ordinary kernel boundary (test_wide_step_in_one_launch_is_bitwise_the_two_launch_form, tuning build: HC_WIDE_FUSED=0), and
// profiles/soak_wide_fused.py soaks it.  The counters are re-zeroed by the last arriver; a launch that was cut short would leave
// them non-zero, so the host clears them before the next fused step whenever a step has failed (hc_step.cpp: tile_counter_suspect).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wide_step_kernel(WideStepArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double red[4][16];
    __shared__ double red_near[4][16];
    __shared__ double red_term[16][16];
    __shared__ int s_last;
    touch_args<sizeof(WideStepArgs)>();
    if (a.f.do_push && (int)blockIdx.x == a.f.nblocks - 1) {
        push_sample<4>(a.f);
        return;
    }
    const int rt = (int)blockIdx.x / a.n.n_slices, sl = (int)blockIdx.x - rt * a.n.n_slices;
    StageClock sc;  // (the host leaves FinalizeArgs::stamps null for this kernel)
    finalize_tile<4, true, false>(a.f, rt, nullptr, red_near, red_term, [&] {
        near_slice<true>(a.n, rt, sl, reinterpret_cast<double*>(smem_raw), red);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial has been written through
        __syncthreads();
        if (threadIdx.x == 0) {
            const int arrived = __hip_atomic_fetch_add(a.tile_counter + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last            = arrived == a.n.n_slices - 1;
            if (s_last) __hip_atomic_store(a.tile_counter + rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next step
        }
        __syncthreads();
        return s_last != 0;
    }, sc);
}

WideLaunch wide_launch_config(WideStepArgs& a) {
    WideLaunch l;
    const NearLaunch nl = near_launch_config(a.n);
    l.grid              = nl.grid + (a.f.do_push ? 1 : 0);
    l.smem              = nl.smem;
    a.f.nblocks         = l.grid;
    a.f.near_partials   = a.n.partials;
    a.f.n_near_slices   = a.n.n_slices;
    a.f.n_near          = 0;
    return l;
}

void launch_wide_step(const WideStepArgs& a0, hipStream_t stream) {
    WideStepArgs a     = a0;
    const WideLaunch l = wide_launch_config(a);
    hipLaunchKernelGGL(wide_step_kernel, dim3(l.grid), dim3(256), l.smem, stream, a);
}

int near_slices_for(int D) { return D >= 1024 ? min(16, max(2, D / 384)) : 1; }

NearLaunch near_launch_config(NearArgs& a) {
    NearLaunch l;
    a.n_slices = near_slices_for(a.D);
    const int groups = (a.D + 7) / 8 + 1;  // a sample's columns may straddle one more group when D % 8 != 0
    a.gps_per_slice = (((groups + a.n_slices - 1) / a.n_slices) + 3) & ~3;  // every wave of the workgroup gets work
    l.grid = a.K.ntiles * a.n_slices;
    l.smem = (size_t)max(1, a.n_near) * 8 * a.gps_per_slice * sizeof(double);
    return l;
}

void launch_near_split(const NearArgs& a0, hipStream_t stream) {
    NearArgs a         = a0;
    const NearLaunch l = near_launch_config(a);
    hipLaunchKernelGGL(near_split_kernel, dim3(l.grid), dim3(256), l.smem, stream, a);
}

// ------------------------------------------------------------------------------------------------
// scatter_kernel: what the sample that has just arrived contributes to the later steps of the look-ahead block.  For each
// IRF sample s in [s_lo, s_lo + ns):  y_s[row] = width_s * sum_col K[row, s*D + col] * v[col], stored times the interpolation
// weight of the sample into the term slot of every later block step it contributes to (at most kTargets), so that the
// step kernels read their terms contiguously and without a table.  One workgroup per (row tile, s): row-owned, no
// partials.  Runs after the step has delivered its forces -- off the caller's critical path.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) scatter_kernel(ScatterArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* vs = reinterpret_cast<double*>(smem_raw);  // [8 * gps_per_slice]: the slice's columns of the sample's velocities
    __shared__ double red[4][16];
    // (no touch_args here: a workgroup needs the head of the 2.6 KB block and ONE row of its target tables; requesting all 41 lines
    // cost 0.5-0.8 us per launch at C4/8, profiles/r04/c4_rank_share_kernel_stats_by_grid.csv before / after)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4;
    const int rt = (int)blockIdx.x % a.K.ntiles, rest = (int)blockIdx.x / a.K.ntiles;
    const int si = rest / a.n_slices, sl = rest - si * a.n_slices;
    const int s  = a.s_lo + si;
    const int D  = a.D;
    const int f0 = s * D, f1 = f0 + D;
    const int g0 = (f0 >> 3) + sl * a.gps_per_slice, g1 = min((f1 + 7) >> 3, g0 + a.gps_per_slice);
    const double* __restrict__ kbase = a.K.base + ((size_t)rt * a.K.ngp) * 128 + lane * 2;
    constexpr int PRE = 4;  // (all 13 column groups of a 384-column slice up front, or the vector straight from L2 without the LDS staging: 10.0-10.3 us against 9.4-9.6 at C4/8, round 4)
    dvec2 pre[PRE];
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int gp = g0 + wave + 4 * q;
        pre[q] = gp < g1 ? *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128) : dvec2{0.0, 0.0};
    }
    for (int i = tid; i < 8 * a.gps_per_slice; i += 256) {
        const int f = g0 * 8 + i;
        vs[i]       = (f >= f0 && f < f1) ? a.v[f - f0] : 0.0;
    }
    __syncthreads();
    const double* __restrict__ u = vs - g0 * 8;
    double acc = 0.0;
    int gp = g0 + wave;
#pragma unroll
    for (int q = 0; q < PRE; ++q, gp += 4) {
        if (gp < g1) {
            acc = fma(pre[q].x, u[gp * 8 + kk], acc);
            acc = fma(pre[q].y, u[gp * 8 + 4 + kk], acc);
        }
    }
#pragma unroll 4
    for (; gp < g1; gp += 4) {
        const dvec2 kv = *reinterpret_cast<const dvec2*>(kbase + (size_t)gp * 128);
        acc = fma(kv.x, u[gp * 8 + kk], acc);
        acc = fma(kv.y, u[gp * 8 + 4 + kk], acc);
    }
    acc += __shfl_xor(acc, 16, kWave);
    acc += __shfl_xor(acc, 32, kWave);
    if (lane < 16) red[wave][lane] = acc;
    __syncthreads();
    if (tid < 16) {
        const double y = (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]) * a.width[s];
        for (int t = 0; t < a.n_tgt[si]; ++t) a.Y[(size_t)a.tgt_off[si][t] + (size_t)sl * a.Dpad + rt * 16 + tid] = a.tgt_coef[si][t] * y;
    }
}

ScatterLaunch scatter_launch_config(ScatterArgs& a) {
    ScatterLaunch l;
    a.n_slices = near_slices_for(a.D);
    const int groups = (a.D + 7) / 8 + 1;  // a sample's columns may straddle one more group when D % 8 != 0
    a.gps_per_slice = (((groups + a.n_slices - 1) / a.n_slices) + 3) & ~3;
    l.grid = a.K.ntiles * a.ns * a.n_slices;
    l.smem = (size_t)8 * a.gps_per_slice * sizeof(double);
    return l;
}

void launch_scatter(const ScatterArgs& a0, hipStream_t stream) {
    if (a0.ns <= 0) return;
    ScatterArgs a         = a0;
    const ScatterLaunch l = scatter_launch_config(a);
    if (l.smem > 64 * 1024) {
        static size_t granted = 0;
        if (l.smem > granted) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.smem);
            granted = l.smem;
        }
    }
    hipLaunchKernelGGL(scatter_kernel, dim3(l.grid), dim3(256), l.smem, stream, a);
}

__global__ void __launch_bounds__(256) ring_transpose_kernel(const double* __restrict__ ring_v, int Hcap, int HcapT, int D, double* __restrict__ ring_vT) {
    const size_t n      = (size_t)Hcap * D;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int slot = (int)(i / D), col = (int)(i - (size_t)slot * D);
        const double v = ring_v[i];
        ring_vT[(size_t)col * HcapT + slot] = v;
        if (slot == 0) ring_vT[(size_t)col * HcapT + Hcap] = v;
    }
}

void launch_ring_transpose(const double* d_ring_v, int Hcap, int HcapT, int D, double* d_ring_vT, hipStream_t stream) {
    const size_t n = (size_t)Hcap * D;
    hipLaunchKernelGGL(ring_transpose_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, d_ring_v, Hcap, HcapT, D, d_ring_vT);
}

// ------------------------------------------------------------------------------------------------
// TaperedDirect preprocessing (TestHydro::EnsureProcessedRIRF, src/hydro_forces.cpp:385-535), once per
// option change: per (row, col) series along s -> truncate, smooth (SG-5 / moving average), half-cosine taper.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) taper_kernel(TaperArgs a) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)a.Dloc * a.D) return;
    const int row = (int)(gid / a.D), col = (int)(gid % a.D);
    const int ngp = a.Kraw.ngp;
    auto in = [&](int s) -> double { return a.Kraw.base[panel_offset(ngp, row, s * a.D + col)]; };
    const int E = a.effective_steps;
    const double sg0 = -3.0 / 35.0, sg1 = 12.0 / 35.0, sg2 = 17.0 / 35.0;
    const int taper_len = a.tc_end - a.tc_index;
    for (int s = 0; s < E; ++s) {
        double v;
        if (a.smoothing == 1) {
            const int half = a.window / 2;
            const int lo = max(0, s - half), hi = min(E - 1, s + half);
            double sum = 0.0;
            for (int k = lo; k <= hi; ++k) sum += in(k);
            const int cnt = hi - lo + 1;
            v = (cnt > 0) ? (sum / cnt) : in(s);
        } else if (E >= 5 && s >= 2 && s <= E - 3) {
            v = sg0 * in(s - 2) + sg1 * in(s - 1) + sg2 * in(s) + sg1 * in(s + 1) + sg0 * in(s + 2);
        } else {
            v = in(s);
        }
        if (s < a.tc_index) {
        } else if (s < a.tc_end && taper_len > 0) {
            const double tt = (double)(s - a.tc_index) / (double)taper_len;
            const double w  = a.final_amplitude + (1.0 - a.final_amplitude) * 0.5 * (1.0 + cos(3.14159265358979323846 * tt));
            v *= w;
        } else {
            v = 0.0;
        }
        a.Kproc[panel_offset(ngp, row, s * a.D + col)] = v;
    }
    for (int s = max(E, 0); s < a.S; ++s) a.Kproc[panel_offset(ngp, row, s * a.D + col)] = 0.0;
}

void launch_taper(const TaperArgs& a, hipStream_t stream) {
    const size_t n = (size_t)a.Dloc * a.D;
    hipLaunchKernelGGL(taper_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
}

// ------------------------------------------------------------------------------------------------
// eta(t) synthesis at init (GetEtaIrregularTimeSeries, src/wave_types.cpp:14-59 with x = 0, then the ramp of
// :759-769).  Thread = one time sample; components are summed in index order like the reference loop.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) eta_kernel(const double* __restrict__ t, int nt, const double* __restrict__ amp,
                                                   const double* __restrict__ omega, const double* __restrict__ phase, int nf,
                                                   double ramp, double* __restrict__ eta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nt) return;
    const double tj = t[j];
    double acc = 0.0;
    for (int i = 0; i < nf; ++i) acc += amp[i] * cos(0.0 - omega[i] * tj + phase[i]);
    if (ramp > 0.0 && tj < ramp) {
        if (tj <= 0.0) acc *= 0.0;
        else acc *= tj / ramp;
    }
    eta[j] = acc;
}

void launch_eta_synthesis(const double* d_t, int nt, const double* d_amp, const double* d_omega, const double* d_phase, int nf,
                          double ramp_duration, double* d_eta, hipStream_t stream) {
    hipLaunchKernelGGL(eta_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream, d_t, nt, d_amp, d_omega, d_phase, nf,
                       ramp_duration, d_eta);
}

// ------------------------------------------------------------------------------------------------
// Added-mass product (ChLoadAddedMass::LoadIntLoadResidual_Mv, src/chloadaddedmass.cpp:55-70): one wave per row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// row . w with 16-byte loads, four of them in flight per lane and four partial sums combined in a fixed order (a row of a wide
// system is 24 KB: with one 8-byte load per lane and iteration the product was latency-bound -- 70 us for 3072 x 3072)
__device__ __forceinline__ double row_dot(const double* __restrict__ m, const double* __restrict__ w, int cols, int lane) {
    const dvec2* __restrict__ m2 = reinterpret_cast<const dvec2*>(m);
    const dvec2* __restrict__ w2 = reinterpret_cast<const dvec2*>(w);
    const int n2 = cols >> 1;  // rows start 16-byte aligned when cols is even (cols = 6N)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    {
        for (int j = lane; j < n2; j += 4 * kWave) {
            dvec2 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = j + kWave * u;
                const bool in = idx < n2;
                a[u] = in ? m2[idx] : dvec2{0.0, 0.0};
                b[u] = in ? w2[idx] : dvec2{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = fma(a[u].x, b[u].x, acc[u]);
                acc[u] = fma(a[u].y, b[u].y, acc[u]);
            }
        }
    }
    double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if ((cols & 1) && lane == 0) s = fma(m[cols - 1], w[cols - 1], s);  // odd column count (not a 6N system: the 1 x 1 self-test)
    return wave_sum(s);
}

__global__ void __launch_bounds__(256) added_mass_mv_kernel(const double* __restrict__ M, int rows, int cols,
                                                             const double* __restrict__ w, double c, double* __restrict__ R) {
    const int row  = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const double acc = row_dot(M + (size_t)row * cols, w, cols, lane);
    if (lane == 0) R[row] += c * acc;
}

// The same product for the host boundary (hc_added_mass_mv): w and the incoming R come from a buffer the host wrote through
// the BAR (or mapped pinned memory), the result leaves as {value, sequence} granules in mapped pinned memory -- one launch,
// no copies, no stream synchronisation, like hc_step.
__global__ void __launch_bounds__(256) added_mass_mv_tagged_kernel(const double* __restrict__ M, int rows, int cols,
                                                                    const double* __restrict__ w, const double* __restrict__ R_in, double c,
                                                                    unsigned long long* __restrict__ tagged, unsigned long long seq,
                                                                    const double* __restrict__ canary_in, unsigned long long* __restrict__ canary_out) {
    const int row  = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 1 && canary_out) {  // the word the host stored behind w and R goes back, tagged (see finalize_kernel)
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u64x2*>(canary_out) = u64x2{(unsigned long long)__double_as_longlong(*canary_in), seq};
    }
    if (row >= rows) return;
    const double acc = row_dot(M + (size_t)row * cols, w, cols, lane);
    if (lane == 0) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const double r = R_in[row] + c * acc;  // the same two roundings as R[row] += c * acc
        *reinterpret_cast<u64x2*>(tagged + 2 * (size_t)row) = u64x2{(unsigned long long)__double_as_longlong(r), seq};
    }
}

void launch_added_mass_mv_tagged(const double* d_M, int rows, int cols, const double* d_w, const double* d_R_in, double c,
                                 unsigned long long* d_tagged, unsigned long long seq, const double* d_canary_in, unsigned long long* d_canary_out,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(added_mass_mv_tagged_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, d_M, rows, cols, d_w, d_R_in, c, d_tagged, seq, d_canary_in,
                       d_canary_out);
}

void launch_added_mass_mv(const double* d_M, int rows, int cols, const double* d_w, double c, double* d_R, hipStream_t stream) {
    hipLaunchKernelGGL(added_mass_mv_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, d_M, rows, cols, d_w, c, d_R);
}

// ------------------------------------------------------------------------------------------------
// Synthetic radiation kernels generated in HBM (benchmark inputs):
//   K[row][col][s] = a * exp(-tau_s / tau_d) * cos(om * tau_s),  tau_s = s*dt,
//   (a, tau_d, om) = per-(row,col) draws of a counter-based splitmix64 stream; same-body blocks x10.
// hydrochrono_amd/synthetic.py holds the identical formula for host-side generation.  Thread = one panel element
// (coalesced stores); padding rows / columns are written as zero.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

__global__ void __launch_bounds__(256) synth_rirf_kernel(double* __restrict__ K, int ntiles, int ngp, int Dloc, int D, int S, int row0, double dt,
                                                          unsigned long long seed, double rho) {
    // grid-stride: a C4-size matrix has more elements (9.7e9) than a launch can have work-items (2^32)
    const size_t n      = (size_t)ntiles * ngp * 128;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gid < n; gid += stride) {
        const int j    = (int)(gid & 1);
        const int lane = (int)((gid >> 1) & 63);
        const size_t blk = gid >> 7;
        const int gp = (int)(blk % ngp), rt = (int)(blk / ngp);
        const int row = rt * 16 + (lane & 15);
        const size_t f = (size_t)gp * 8 + 4 * j + (lane >> 4);
        double val = 0.0;
        if (row < Dloc && f < (size_t)S * D) {
            const int s = (int)(f / D), col = (int)(f - (size_t)s * D);
            const int grow = row0 + row;
            const uint64_t base = splitmix64(seed ^ (((uint64_t)grow << 32) | (uint64_t)col));
            const double ua = u01(splitmix64(base + 1)), ud = u01(splitmix64(base + 2)), uo = u01(splitmix64(base + 3));
            double amp = (2.0 * ua - 1.0);
            if (grow / 6 == col / 6) amp *= 10.0;
            const double tau_d = 1.0 + 3.0 * ud;
            const double om    = 0.5 + 2.5 * uo;
            const double tau   = s * dt;
            val = (amp * exp(-tau / tau_d) * cos(om * tau)) * rho;
        }
        K[gid] = val;
    }
}

void launch_synth_rirf(double* d_K, int ntiles, int ngp, int Dloc, int D, int S, int row0, double dt, unsigned long long seed, double rho,
                       hipStream_t stream) {
    const size_t n      = (size_t)ntiles * ngp * 128;
    const size_t blocks = std::min<size_t>((n + 255) / 256, (size_t)1 << 22);  // <= 2^30 work-items per launch
    hipLaunchKernelGGL(synth_rirf_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_K, ntiles, ngp, Dloc, D, S, row0, dt, seed, rho);
}

}  // namespace hc
