// hc_limits.hpp -- compile-time capacities shared by the kernels (hc_kernels.hpp) and the host-only look-ahead planner
// (hc_plan.hpp); no HIP dependency.
#pragma once

namespace hc {

// most future steps one blocked pass covers (blocks of 16 = N dimension of v_mfma_f64_16x16x4_f64) and the IRF samples s < kScatterSamples
// that can be targets of a scatter.  The shipped depths are 16 and 32; the tuning build (-DHC_TUNING) also holds the depth-64 pass that
// was measured in round 5 and not taken (EXPERIMENTS.md), which needs room for 64 steps and a scatter reach of 66 samples.  The
// capacities size the plan (reset at every block boundary on the host), the argument blocks (stored through the BAR at every
// dispatch) and the term / partial buffers, so the release build keeps the small ones.
#ifdef HC_TUNING
constexpr int kLookahead      = 64;
constexpr int kScatterSamples = 80;
#else
constexpr int kLookahead      = 32;
constexpr int kScatterSamples = 64;
#endif
constexpr int kDepthDefault = 32;  // the look-ahead depth of a fresh context (hc_set_lookahead)
constexpr int kNearMax        = 8;   // IRF samples a step contracts itself (own sample + a deferred one)
constexpr int kTermMax        = 192; // term slots a step adds (scatter results; x column slices for wide systems)
constexpr int kTargets        = 3;   // later block steps one (sample, IRF sample) result can contribute to

// Stage clock of the step kernel (tuning build only: FinalizeArgs::stamps, hc_tuning_step_stamps): one row of kStampStages 100 MHz
// s_memrealtime values per workgroup of a launch, for the last kStampSteps steps.
constexpr int kStampStages    = 12;
constexpr int kStampWGs       = 64;
constexpr int kStampSteps     = 64;

constexpr int kStepHalvesMinColumns = 96;  // step_hot_kernel: from this many columns on, two workgroups share a row tile (StepHotArgs::halves)

constexpr int kSubBlock       = 8;   // steps per sub-block of the two-level form (wide systems)
constexpr int kMiniChunks     = 256; // radiation chunks a NARROW short pass may have (one step offset per chunk travels in the argument block)

// The step kernel's body state behind its argument block (direct dispatch, hc_direct.hpp: kSlotBytes / kExtraBytes): at byte
// kSlotArgBytes from the kernarg segment pointer, in the order [velocities by DoF column 6N | positions 3N | angles 3N | canary word].
constexpr int kSlotArgBytes      = 4096;
constexpr int kSlotStateDoubles  = 2048;  // kExtraBytes / 8
constexpr int kSlotStateMaxBodies = 170;  // 12 N + 1 doubles fit, and 6 N <= 4 x 256: four early loads per work-item cover every column
                                          // (every system whose step is ONE launch: wide systems begin at 6 N = 1024)

#if defined(__HIPCC__)
#define HC_HOST_DEVICE __host__ __device__
#else
#define HC_HOST_DEVICE
#endif

// Bracket of query time q for the short pass of the two-level form.  time[0] is the (not yet known, zero) sample of the step that
// follows the sub-block, time[1..kw] are the sub-block's samples newest first, time[kw + 1] the sample before them -- all on the
// plan's predicted grid, so the comparisons are the planner's, bit for bit.  Weights of InterpolateVelocity6D
// (src/hydro_forces.cpp:343-371), already masked to the sub-block's samples: *wn belongs to sample *lo, *wo to sample *lo + 1.
// false: q is not bracketed (cannot happen for q <= time[0]).
HC_HOST_DEVICE inline bool mini_bracket(const double* time, int kw, double q, double* wo, double* wn, int* lo_out) {
    int lo = 0;
    while (lo <= kw && !(time[lo + 1] <= q)) ++lo;  // smallest lo with time[lo + 1] <= q (AdvanceToBracket, :374-381)
    *wo     = 0.0;
    *wn     = 0.0;
    *lo_out = lo;
    if (lo > kw) return true;  // both samples are older than the sub-block: the pass of the block (or an earlier short pass) has them
    const double newer = time[lo], older = time[lo + 1];
    double o, n;
    if (q == older) { o = 1.0; n = 0.0; }
    else if (q == newer) { o = 0.0; n = 1.0; }
    else if (q > older && q < newer) {
        const double td = newer - older;
        o = (td != 0.0) ? ((newer - q) / td) : 0.0;
        n = 1.0 - o;
    } else {
        return false;
    }
    *wn = (lo >= 1) ? n : 0.0;
    *wo = (lo + 1 <= kw) ? o : 0.0;
    return true;
}

}  // namespace hc
