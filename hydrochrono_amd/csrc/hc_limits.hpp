// hc_limits.hpp -- compile-time capacities shared by the kernels (hc_kernels.hpp) and the host-only look-ahead planner
// (hc_plan.hpp); no HIP dependency.
#pragma once

namespace hc {

constexpr int kLookahead = 32;  // most future steps one blocked pass covers (1 or 2 blocks of 16 = N dimension of v_mfma_f64_16x16x4_f64)
constexpr int kNearMax        = 8;   // IRF samples a step contracts itself (own sample + a deferred one)
constexpr int kTermMax        = 96;  // scatter results a step adds
constexpr int kScatterSamples = 64;  // IRF samples s < kScatterSamples can be targets of a scatter
constexpr int kTargets        = 3;   // later block steps one (sample, IRF sample) result can contribute to

}  // namespace hc
