// hc_kernels.hpp -- launch interface of the gfx950 kernels of the hydro-force path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace hc {

// ---- per-step argument blocks (passed by value as kernel arguments; no per-step H2D descriptor copies) ----

// Velocity-history ring in HBM: ring_v[Hcap][D], ring_t[Hcap]; sample k (0 = newest) lives in slot
// (head - k + Hcap) % Hcap.  The sample of the CURRENT step (k = 0) is read from `state`, never from the
// ring, so the workgroup that stores it into slot `head` (finalize_kernel) races with nobody.
struct HistoryView {
    const double* state;  // this step: pos[3N] | rpy[3N] | linvel[3N] | angvel[3N]
    int N, D;
    double t;
    const double* ring_t;
    const double* ring_v;
    int head, H, Hcap;    // H counts the current sample
    double dt_hint;       // t - previous sample time (bracket-search hint only, > 0)
};

// One launch per step covers the radiation matrix K[D_local x S*D] and, for irregular waves, the excitation
// matrix Kex[D_local x Lpad]; a workgroup owns R consecutive rows x one column chunk of one of them and leaves one
// partial sum per row in `partials[chunk][D_local]`.  The right-hand sides are never materialised: the
// interpolated, width-scaled velocity history u[s][col] and the free-surface samples e[j] are formed in registers
// from the ring / the eta table (both L2-resident) while K streams from HBM.
struct ConvArgs {
    const double* K;
    size_t ldk;       // row stride of K in doubles (even)
    int F;            // S*D
    int chunk_cols;   // multiple of 512
    int nchunks_rad;
    int max_steps_per_chunk;  // size of the per-workgroup bracket table: chunk_cols / D + 2
    HistoryView hist;
    int S;
    const double* tau;    // [S] radiation IRF sample times
    const double* width;  // [S] trapezoid widths
    const double* Kex;
    size_t ldkex;     // Lpad
    int L, Lpad;
    int chunk_cols_ex;
    int nchunks_ex;
    const double* ex_tau;    // [L]
    const double* ex_width;  // [L]
    const double* eta_t;     // [nt]
    const double* eta;       // [nt]
    int nt;
    double eta_dt;           // nominal spacing of eta_t (search hint only)
    double* partials;  // [(nchunks_rad + nchunks_ex)][Dloc]
    int Dloc;
    int nrowtiles;
    int* error_flag;   // set to 1 / 2 if a query time is not bracketed (reference: runtime_error)
};

struct FinalizeArgs {
    const double* partials;
    int nchunks_rad, nchunks_ex;
    int Dloc, N, b0;
    const double* state;
    // hydrostatics
    const double* lin;       // [nloc][36]
    const double* cg;        // [nloc][3]
    const double* cb_m_cg;   // [nloc][3]
    const double* disp_vol;  // [nloc]
    double rho;
    double gx, gy, gz;
    // waves
    int wave_mode;            // 0 none, 1 regular, 2 irregular
    const double* reg_mag;    // [Dloc]
    double reg_phase[6];      // body-0 phases (reference indexes the phase by DoF only, src/wave_types.cpp:323)
    double reg_amplitude, reg_omega, t;
    int do_hs, do_rad, do_waves;
    // outputs
    double* hs;
    double* rad;
    double* waves;
    double* total;
    double* user_out;  // may be null
    // history push of this step's sample into ring slot `head` (src/hydro_forces.cpp:559-574)
    int do_push, head, D;
    double* ring_t;
    double* ring_v;
};

struct TaperArgs {
    const double* Kraw;
    double* Kproc;
    size_t ldk;
    int Dloc, D, S;
    int effective_steps;
    int smoothing;  // 0 sg5, 1 moving average
    int window;     // moving average window (already max(3, window_length))
    int tc_index, tc_end;
    double final_amplitude;
};

// ---- launchers (all asynchronous on `stream`) ----
void launch_relayout_rirf(const double* d_Kb_6xDxS, double* d_K, int D, int S, size_t ldk, int row0, double rho, hipStream_t stream);
// rows_per_tile is 6 or 12 (Dloc % rows_per_tile == 0)
void launch_conv(const ConvArgs& a, int rows_per_tile, hipStream_t stream);
void launch_finalize(const FinalizeArgs& a, hipStream_t stream);
void launch_taper(const TaperArgs& a, hipStream_t stream);
// eta[j] = sum_i amp[i] * cos(-omega[i]*t[j] + phase[i]), then the ramp rule of src/wave_types.cpp:759-769
void launch_eta_synthesis(const double* d_t, int nt, const double* d_amp, const double* d_omega, const double* d_phase, int nf,
                          double ramp_duration, double* d_eta, hipStream_t stream);
// R[i] += c * sum_j M[i][j] * w[j]   (i < rows)
void launch_added_mass_mv(const double* d_M, int rows, int cols, const double* d_w, double c, double* d_R, hipStream_t stream);
// out[(row*D + col)*S + s] = K[row][s*D + col]  (reference indexing; diagnostics)
void launch_unrelayout(const double* d_K, size_t ldk, int Dloc, int D, int S, double* d_out, hipStream_t stream);
// synthetic many-body coefficient generator
void launch_synth_rirf(double* d_K, size_t ldk, int Dloc, int D, int S, int row0, double dt, unsigned long long seed, double rho,
                       hipStream_t stream);

const char* conv_kernel_name();

}  // namespace hc
