// hc_kernels.hpp -- launch interface of the gfx950 kernels of the hydro-force path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "hc_limits.hpp"

namespace hc {


// ------------------------------------------------------------------------------------------------------------------
// Panel layout of the convolution matrices in HBM (radiation K[D_loc x S*D], excitation Kex[D_loc x L]).
// Rows are grouped in tiles of 16, columns in groups of 8 ("gp").  One 16 x 8 block is 128 consecutive doubles:
//     element (row, f):  rt = row/16, ri = row%16, gp = f/8, j = (f%8)/4, kk = f%4
//     offset = ((rt*ngp + gp)*64 + (kk*16 + ri))*2 + j
// so that lane l = kk*16 + ri of a wave reads ONE 16-byte word per 16 x 8 block (a 1 KiB fully coalesced wave load) and
// holds exactly the A operands of two v_mfma_f64_16x16x4_f64 (A[row = l&15][k = l>>4]) for columns 8gp+kk and 8gp+4+kk.
// Rows >= D_loc and columns >= F are zero padding.
// ------------------------------------------------------------------------------------------------------------------
struct Panel {
    const double* base;
    int ntiles;  // ceil(D_loc / 16)
    int ngp;     // ceil(F / 8)
};
inline size_t panel_doubles(int ntiles, int ngp) { return static_cast<size_t>(ntiles) * ngp * 128; }

// Velocity-history ring in HBM: ring_v[Hcap][D], ring_t[Hcap]; sample k (0 = newest) lives in slot
// (head - k + Hcap) % Hcap.  The sample of the CURRENT step (k = 0) is read from `state`, never from the
// ring, so the workgroup that stores it into slot `head` (finalize_kernel) races with nobody.
struct HistoryView {
    const double* state;  // this step: pos[3N] | rpy[3N] | linvel[3N] | angvel[3N]
    int N, D;
    double t;
    const double* ring_t;
    const double* ring_v;
    const double* ring_vT;  // the same samples per DoF: ring_vT[col][HcapT] (each DoF's time series contiguous; look-ahead pass).
                            // HcapT = Hcap + 2: entry [Hcap] mirrors slot 0, so that the two samples of a bracket (ring slots
                            // k, k+1 mod Hcap) are always 16 contiguous bytes -- one load in the pass
    int head, H, Hcap;    // H counts the current sample
    int HcapT;            // row length of ring_vT
    double dt_hint;       // t - previous sample time (bracket-search hint only, > 0)
};

// Interpolation bracket of one IRF sample's query time in the history (AdvanceToBracket + InterpolateVelocity6D weights,
// src/hydro_forces.cpp:343-381).
struct Bracket {
    double wo, wn;   // weights of the older / newer sample (both 0: the sample contributes nothing)
    int off_older;   // element offset (slot * D) of the older sample's ring row
    int off_newer;   // element offset of the newer sample's ring row, -1: the newer sample is the current state
};

// ------------------------------------------------------------------------------------------------------------------
// Scatter-form look-ahead (hc_step.cpp: make_plan).  The interpolated history is linear in its samples,
//     v~(q) = sum_k phi_k(q) v_k      (phi_k = the reference's two interpolation weights of sample k, src/hydro_forces.cpp:343-371),
// so the radiation sum of a future step m splits by history sample: what the samples known when the block was planned
// contribute (the look-ahead pass, K read once for a block of 32 or 16 steps), what a sample that arrives at block step i < m contributes
// (scatter_kernel right after step i has delivered its forces, off the caller's critical path), and what step m's own
// sample contributes (the few IRF samples tau_s < dt, contracted by step_kernel itself).  A step inside a block is then
// ONE launch on the caller's critical path.
// ------------------------------------------------------------------------------------------------------------------
// capacities kLookahead, kNearMax, kTermMax, kScatterSamples, kTargets: hc_limits.hpp

// One IRF sample s contracted by the step itself: columns [s*D, (s+1)*D) of K against
//     u[col] = a * v_state[col] + b * ring_v[off_b + col] + c * ring_v[off_c + col]
// (a: this step's own sample, weight x trapezoid width; b, c: a complete bracket of stored samples, used for the one IRF
// sample per step whose "is there an older sample" test the pass could not decide ahead of time).
struct NearEntry {
    int s, off_b, off_c, pad;
    double a, b, c;
};

// Excitation side shared by the per-step and the look-ahead launch: e[j] = eta(t - ex_tau[j]) * ex_width[j], eta linearly
// interpolated in the precomputed free-surface table (src/wave_types.cpp:797-831).
struct EtaTable {
    int L;                   // excitation samples (columns of Kex)
    const double* ex_tau;    // [L]
    const double* ex_width;  // [L]
    const double* eta_t;     // [nt]
    const double* eta;       // [nt]
    int nt;
    double eta_dt;           // nominal spacing of eta_t (search hint only)
    double eta_t0;           // eta_t[0]
};

// Plain per-step launch: radiation columns [0, F_limit) of K (F_limit = live samples * D) and, for irregular waves, the
// excitation matrix, both as column chunks of a streamed FP64 GEMV.  A workgroup owns MT row tiles x one chunk and leaves
// one partial per row in partials[chunk][Dpad].
struct StepArgs {
    Panel K;
    int F_limit;          // radiation columns [0, F_limit) to contract (multiple of D)
    int chunk_gp;         // column groups per radiation chunk
    int nchunks_rad;      // chunks covering [0, F_limit)
    int max_steps_per_chunk;  // LDS bracket table entries
    int rhs_capacity;         // LDS right-hand-side entries: 8 * max(chunk_gp, chunk_gp_ex)
    HistoryView hist;
    const double* tau;    // [S] radiation IRF sample times
    const double* width;  // [S] trapezoid widths
    Panel Kex;
    EtaTable ex;
    int chunk_gp_ex;
    int nchunks_ex;
    double* partials;        // [(nchunks_rad + nchunks_ex)][Dpad]
    int Dpad;
    int ngroups;             // ntiles / MT
    int* error_flag;         // 1 / 2: a query time is not bracketed (reference: runtime_error)
};

// Look-ahead pass: for the `depth` predicted steps j of a block, the part of step j's radiation sum that depends only on
// history known when the block is planned,
//   P_j[row] = sum over s >= s_cut[j], col of K[row, s, col] * u_{n+j}(s, col),   t_{n+j} = t + j*dt,
// as one [D_loc x F] x [F x depth] FP64 GEMM on the matrix cores; K is read once for the whole block.
struct BlockArgs {
    Panel K;
    int F;                // S*D
    int chunk_gp;
    int nchunks;
    int max_steps_per_chunk;
    int lds_front_doubles;  // set by the launcher
    int depth;              // steps covered: 16 or 32
    HistoryView hist;
    double tpred[kLookahead];  // predicted step times, tpred[0] = hist.t
    int s_cut[kLookahead];
    int s_defer[kLookahead];   // IRF sample left to the step itself although it is >= s_cut (-1: none), see plan_step
    const double* tau;
    const double* width;
    // excitation for the 16 predicted times rides in the same launch as extra chunks (nchunks_ex may be 0):
    // E_j[row] = sum_l Kex[row, l] * eta(tpred[j] - ex_tau[l]) * ex_width[l]
    Panel Kex;
    EtaTable ex;
    int chunk_gp_ex;
    int nchunks_ex;
    double* partials;     // [nchunks + nchunks_ex][16][Dpad]
    int Dpad;
    int ngroups;
    int* error_flag;
    int* item_counter;    // excitation work items taken so far in this launch (zero at launch; reset by reduce_block_kernel)
    // Short pass of the two-level form (hc_plan.hpp: MiniPass; mini_kw == 0: the ordinary pass of a block).  Only the brackets that
    // touch the mini_kw samples of the sub-block that has just ended count; their times come from the plan's predicted grid
    // (mini_time[k], history index k = 0 .. mini_kw + 1, index 0 = hist.t), not from the ring, so the bracket of every
    // (IRF sample, step) is the planner's bit for bit.  Steps j >= mini_steps are empty.
    int mini_kw, mini_steps;
    double mini_time[kLookahead + 2];
    // A launch may cover a range of the radiation chunks only (chunk_last == 0: all of them): the pass of the NEXT block is issued in
    // slices between the steps of the current one (hc_step.cpp: pass schedule "one block ahead"), and a short pass towards the next
    // block starts at the first IRF sample it needs.  Partials are indexed by the absolute chunk number either way, so the sums
    // reduce_block_kernel forms do not depend on how the chunks were spread over launches.
    int chunk_first, chunk_last;
    // NARROW short pass (mini_narrow != 0; depth == 16): the sub-block's kw samples reach, through ONE IRF sample, a window of only
    // kw + 2 consecutive block steps, so a chunk (half an IRF sample) needs 16 step columns, not 32 -- the 16 columns of chunk c are the
    // steps mini_jbase[c] .. mini_jbase[c] + 15 (of the pass's own step numbering, index into tpred / s_cut / s_defer), and its
    // partials are [chunk][16][Dpad]; reduce_block_kernel adds them to the rows they belong to.  Half the matrix work, the
    // B-operand gathers and the partials of the wide form, and the 16-step kernel (two or three waves per SIMD instead of one).
    int mini_narrow;
    unsigned char mini_jbase[kMiniChunks];
};

// near_split_kernel (wide systems): the step's own-sample part  K[rows of a tile, columns of the near samples] x u  split over
// column slices so that hundreds of workgroups stream it instead of one per row tile; slice partials [slice][Dpad], added by the
// step kernel in a fixed order.  Slice geometry depends on D only (never on the rows a context owns): row shards stay bitwise.
struct NearArgs {
    Panel K;
    int D, Dpad, N;
    int n_near;
    NearEntry near[kNearMax];
    const double* state;
    const double* ring_v;
    int n_slices, gps_per_slice;  // column groups of one IRF sample's D columns (+ a straddled group) per slice
    double* partials;             // [n_slices][Dpad]
};

struct FinalizeArgs {
    const double* partials;
    int nchunks_rad, nchunks_ex;
    const double* near_partials;  // [n_near_slices][Dpad] from near_split_kernel (wide systems), or null
    int n_near_slices;
    const double* P;         // look-ahead part of this step's radiation sum, [Dpad] (may be null)
    const double* E;         // excitation force of this step precomputed by the look-ahead pass, [Dpad] (may be null)
    int Dloc, Dpad, N, b0;
    const double* state;
    // hydrostatics
    const double* lin;       // [nloc][36]
    const double* cg;        // [nloc][3]
    const double* cb_m_cg;   // [nloc][3]
    const double* disp_vol;  // [nloc]
    double rho;
    double gx, gy, gz;
    // waves
    int wave_mode;            // 0 none, 1 regular, 2 irregular (excitation-IRF convolution), 3 irregular (spectral component sum)
    const double* reg_mag;    // [Dloc]
    double reg_phase[6];      // body-0 phases (reference indexes the phase by DoF only, src/wave_types.cpp:323)
    double reg_amplitude, reg_omega, t;
    // spectral mode: f[row] = ramp(t) * sum_i |X_row(w_i)| a_i cos(w_i t - phi_i + arg X_row(w_i))
    int spec_nf;
    const double* spec_mag;    // [Dloc][nf]
    const double* spec_phase;  // [Dloc][nf]
    const double* spec_amp;    // [nf] sqrt(2 S df)
    const double* spec_omega;  // [nf]
    const double* spec_phi;    // [nf] random phases
    double spec_ramp;
    int do_hs, do_rad, do_waves;
    // outputs
    double* hs;
    double* rad;
    double* waves;
    double* total;
    double* user_out;  // may be null
    // Step inside a look-ahead block: the IRF samples this step contracts itself (row-owned, no partials) ...
    int n_near;
    NearEntry near[kNearMax];
    Panel nearK;
    const double* ring_v_ro;   // velocity ring (read side)
    // ... and the scatter results of the earlier steps of the block, already weighted and laid out for this step by the
    // scatter launches: rad += sum_{k < n_terms} Yc[k][row]  (one contiguous read, no table)
    int n_terms;
    const double* Yc;  // [n_terms][Dpad]
    // Host boundary: the totals also go to mapped pinned host memory as 16-byte granules {total, seq}; the host spins on
    // seq instead of synchronising the stream (one store carries value and sequence number, so no ordering is assumed).
    unsigned long long* host_tagged;  // [Dloc][2] or null
    unsigned long long seq;
    // Proof that the kernel read THIS step's state: the host stores the step's sequence number behind the state (same buffer, same
    // store path -- the PCIe BAR), the kernel hands the word back as one more tagged granule and the host compares (hc_step only;
    // both null otherwise).  A GPU that served a stale copy of host-rewritten memory would show here at the step it happens.
    const double* canary_in;
    unsigned long long* canary_out;   // [2] {canary bits, seq}
    // history push of this step's sample into ring slot `head` (src/hydro_forces.cpp:559-574)
    int do_push, head, D;
    int nblocks;      // workgroups of the launch (set by finalize_launch_config; the last one stores the sample)
    double* ring_t;
    double* ring_v;
    double* ring_vT;  // [D][HcapT] transposed copy (entry [Hcap] mirrors slot 0)
    int Hcap, HcapT;
#ifdef HC_TUNING
    // stage clock (profiles/step_stamps_probe.cpp): [workgroup][kStampStages] s_memrealtime values of this launch, or null
    unsigned long long* stamps;
#endif
};

// finalize_pre_kernel's arguments as the kernel lays them out: ten preloaded words in front of the step kernel's argument block
// (hc_kernels.hip; filled by hc_step.cpp: enqueue_step)
struct FinalizePreArgs {
    const double* kfirst;
    const double* yc;
    int ngp, ngroups, n_terms, dpad, ntiles, pad_;
    FinalizeArgs a;
};
static_assert(offsetof(FinalizePreArgs, a) == 40, "kernarg layout of finalize_pre_kernel");

// step_hot_kernel: the step kernel of the COMMON block step -- a step inside a look-ahead block of a system that is not wide, direct
// dispatch with the body state behind the arguments, the step's own IRF samples weighted against its own velocity only (NE = 1 or 2 of
// them), no plain partials, no spectral wave mode -- with a compact argument block of its own.  Everything finalize_kernel<4, true>
// does for such a step, bit for bit (hc_step.cpp: enqueue_step decides; HC_STEP_HOT=0 in the tuning build keeps the general kernel).
struct StepHotArgs {
    // ---- first part: what the kernel needs to issue its loads (requested whole, one wait) ----
    const double* kfirst[2];  // K's panel base + the first column group of own IRF sample e (row tile 0)
    const double* Yc;         // [n_terms][Dpad] scatter results for this step
    const double* P;          // look-ahead row of this step
    const double* E;          // excitation row of this step (has_E), else any valid address
    const double *lin, *cg, *cb_m_cg, *disp_vol, *reg_mag;
    int ngp;                  // column groups per row tile of the panel
    int ng[2];                // column groups of own sample e
    int n_terms, Dpad, Dloc, N, b0;
    int ntiles;               // workgroups that finish rows: row tiles x halves (the workgroup behind them stores the sample)
    int halves;               // 1: a workgroup finishes the 16 rows of a tile; 2: 8 of them (twice the compute units share the K words)
    // ---- second part: requested while the loads are in flight ----
    int off[2];               // first column of own sample e inside its first group (s * D - 8 * (s * D / 8))
    int D, wave_mode, has_E, pad1_;
    double a[2];              // weight x trapezoid width of own sample e
    double rho, gx, gy, gz, t, reg_amplitude, reg_omega;
    double reg_phase[6];
    unsigned long long seq;
    double *hs, *rad, *waves, *total;
    unsigned long long* host_tagged;
    unsigned long long* canary_out;
    unsigned long long* stamps;  // stage clock rows (tuning build), else null
    // the workgroup behind the row tiles stores the sample (ring slot `head`)
    double *ring_t, *ring_v, *ring_vT;
    int head, Hcap, HcapT, pad2_;
};

// wide_step_kernel: near_split_kernel + the step kernel of a wide system in one launch (the workgroup that completes a row tile's
// slices finishes the tile).  tile_counter: [ntiles] ints, zero between launches.
struct WideStepArgs {
    NearArgs n;
    FinalizeArgs f;
    int* tile_counter;
};

// scatter_kernel: y_s[row] = width[s] * sum_col K[row, s*D + col] * v[col] for s in [s_lo, s_lo + ns); one workgroup per
// (row tile, sample), so nothing is left to reduce across workgroups.  Each result goes, times the interpolation weight of
// the sample, straight into the term slots of the later block steps it contributes to: Y[tgt_off + row] = tgt_coef * y_s.
struct ScatterArgs {
    Panel K;
    int D, Dpad;
    int s_lo, ns;
    const double* v;      // [D] the sample's velocities (its ring row)
    const double* width;  // [S]
    double* Y;            // term slots of the block: [step][kTermMax][Dpad]
    // wide systems: a sample's D columns are split over n_slices workgroups per row tile (same slice geometry as near_split_kernel);
    // slice sl leaves its partial in term slot tgt_off + sl * Dpad -- the step kernel adds the slices like any other terms
    int n_slices, gps_per_slice;
    int n_tgt[kScatterSamples];               // per s - s_lo
    int tgt_off[kScatterSamples][kTargets];   // (step * kTermMax + term index) * Dpad
    double tgt_coef[kScatterSamples][kTargets];
};

// added_mass_mv_tagged_kernel's arguments as the kernel lays them out (direct dispatch, hc_step.cpp)
struct AddedMassArgs {
    const double* M;
    int rows, cols;
    const double *w, *R_in;
    double c;
    unsigned long long* tagged;
    unsigned long long seq;
    const double* canary_in;          // the word the host stored behind w and R (null: none)
    unsigned long long* canary_out;   // [2] {canary bits, seq}
};
static_assert(sizeof(AddedMassArgs) == 72, "kernarg layout of added_mass_mv_tagged_kernel");

// reduce_block_kernel's arguments as the kernel lays them out.  accumulate != 0 (short pass of the two-level form): the chunk sum
// of step j is ADDED to row j_off + j of P for j < j_cnt (the rows of the block steps still to come), nothing else is touched.
struct ReduceArgs {
    const double* partials;
    int nchunks_rad, nchunks_ex, Dpad, depth;
    double *P, *E;
    int* item_counter;
    int accumulate, j_off, j_cnt;
    int rad_first;  // first radiation chunk to add (short passes that start past IRF sample 0)
    // narrow short pass (BlockArgs::mini_narrow): the partials of chunk c are [16][Dpad] and belong to steps jbase[c] .. jbase[c] + 15
    int narrow, pad_;
    unsigned char jbase[kMiniChunks];
};
static_assert(sizeof(ReduceArgs) == 72 + kMiniChunks, "kernarg layout of reduce_block_kernel");

struct TaperArgs {
    Panel Kraw;
    double* Kproc;  // same panel geometry
    int Dloc, D, S;
    int effective_steps;
    int smoothing;  // 0 sg5, 1 moving average
    int window;     // moving average window (already max(3, window_length))
    int tc_index, tc_end;
    double final_amplitude;
};

// ---- launch geometry shared by the HIP launchers below and the direct AQL dispatch of hc_step.cpp (hc_direct.hpp) ----
struct FinalizeLaunch {
    int grid = 0, threads = 256;
    size_t smem = 0;
};
struct NearLaunch {
    int grid = 0;
    size_t smem = 0;
};
// slices of the own-sample part for a system of D columns (1: the step kernel contracts it itself)
int near_slices_for(int D);
NearLaunch near_launch_config(NearArgs& a);  // fills n_slices / gps_per_slice
struct WideLaunch {
    int grid = 0;
    size_t smem = 0;
};
WideLaunch wide_launch_config(WideStepArgs& a);  // fills the slice geometry, f.nblocks and f.near_partials / n_near_slices (f.n_near = 0)
struct ScatterLaunch {
    int grid = 0;
    size_t smem = 0;
};
ScatterLaunch scatter_launch_config(ScatterArgs& a);  // fills n_slices / gps_per_slice
FinalizeLaunch finalize_launch_config(FinalizeArgs& a);  // also fills a.nblocks
struct StepLaunch {
    int nblocks = 0;
    size_t smem = 0;
    int MT = 0, U = 0;  // conv_step_kernel<MT, U>
};
StepLaunch step_launch_config(const StepArgs& a, int mt);
struct BlockLaunch {
    int nblocks = 0;
    size_t smem = 0;
    int MT = 0, R = 0, NB = 0, WPS = 0;  // conv_block_kernel<MT, R, NB, WPS>
};
BlockLaunch block_launch_config(const BlockArgs& a, int mt, BlockArgs* with_lds_layout);

// ---- launchers (all asynchronous on `stream`) ----
// staging Kb[6][D][S] (file order, unscaled) -> rows row0..row0+5 of the panel matrix, times `scale`
void launch_relayout_rirf(const double* d_Kb_6xDxS, double* d_K, int ngp, int D, int S, int row0, double scale, hipStream_t stream);
// row-major src[rows][cols] -> panel rows row0.. (used for the excitation IRF)
void launch_relayout_rowmajor(const double* d_src, int rows, int cols, double* d_panel, int ngp, int row0, hipStream_t stream);
// mt = row tiles per workgroup (1, 2 or 4 -- the look-ahead launch also 6; ngroups*mt == ntiles)
void launch_conv_step(const StepArgs& a, int mt, hipStream_t stream);
void launch_conv_block(const BlockArgs& a, int mt, hipStream_t stream);
// P[j][row] = sum over the radiation chunks c of partials[c][j][row], E[j][row] = the same over the excitation chunks
// (fixed order; nchunks_ex may be 0)
void launch_reduce_block(const ReduceArgs& r, hipStream_t stream);
int reduce_block_grid(const ReduceArgs& r);
void launch_finalize(const FinalizeArgs& a, hipStream_t stream);
void launch_near_split(const NearArgs& a, hipStream_t stream);
void launch_wide_step(const WideStepArgs& a, hipStream_t stream);
void launch_scatter(const ScatterArgs& a, hipStream_t stream);
void launch_taper(const TaperArgs& a, hipStream_t stream);
// eta[j] = sum_i amp[i] * cos(-omega[i]*t[j] + phase[i]), then the ramp rule of src/wave_types.cpp:759-769
void launch_eta_synthesis(const double* d_t, int nt, const double* d_amp, const double* d_omega, const double* d_phase, int nf,
                          double ramp_duration, double* d_eta, hipStream_t stream);
// R[i] += c * sum_j M[i][j] * w[j]   (i < rows)
void launch_added_mass_mv(const double* d_M, int rows, int cols, const double* d_w, double c, double* d_R, hipStream_t stream);
// tagged[row] = {R_in[row] + c * sum_j M[row][j] * w[j], seq} as 16-byte granules (host boundary)
void launch_added_mass_mv_tagged(const double* d_M, int rows, int cols, const double* d_w, const double* d_R_in, double c,
                                 unsigned long long* d_tagged, unsigned long long seq, const double* d_canary_in, unsigned long long* d_canary_out,
                                 hipStream_t stream);
// out[(row*D + col)*S + s] = K[row][s*D + col]  (reference indexing; diagnostics)
void launch_unrelayout(const Panel& K, int Dloc, int D, int S, double* d_out, hipStream_t stream);
// out[s] = K[row][s*D + col], s < S  (one series; diagnostics)
void launch_extract_series(const Panel& K, int row, int col, int D, int S, double* d_out, hipStream_t stream);
// ring_vT[col][slot] = ring_v[slot][col] for all slots, ring_vT[col][Hcap] = ring_v[0][col] (after the ring has been re-allocated or injected)
void launch_ring_transpose(const double* d_ring_v, int Hcap, int HcapT, int D, double* d_ring_vT, hipStream_t stream);
// synthetic many-body coefficient generator: fills the whole panel matrix (padding = 0)
void launch_synth_rirf(double* d_K, int ntiles, int ngp, int Dloc, int D, int S, int row0, double dt, unsigned long long seed, double rho,
                       hipStream_t stream);

}  // namespace hc
