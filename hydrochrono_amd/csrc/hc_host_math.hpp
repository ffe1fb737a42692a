// hc_host_math.hpp -- init-time host arithmetic of the hydro-force path (product code).
//
// Everything here runs once per configuration on the host: grids, trapezoid widths, wave spectrum, random
// phases, dispersion relation, cubic-B-spline resampling of the excitation IRF.  The per-step work and the
// eta(t) synthesis live in hc_kernels.hip.  Reference lines are cited per function.
#pragma once
#include <cstdint>
#include <vector>

namespace hc {

// Eigen::VectorXd::LinSpaced(n, lo, hi) semantics (used by src/wave_types.cpp:584,594-595,654,737).
std::vector<double> linspaced(int n, double lo, double hi);

// Trapezoid half-widths of a sample grid (src/hydro_forces.cpp:181-190, src/wave_types.cpp:608-620).
std::vector<double> trapezoid_widths(const std::vector<double>& grid);

// Pierson-Moskowitz / JONSWAP spectral density in Hz (src/wave_types.cpp:679-715). `f` must be ascending.
std::vector<double> jonswap_spectrum_hz(const std::vector<double>& f, double Hs, double Tp, double gamma, bool normalized);

// Random phases in [0, 2pi): std::mt19937(seed) + the two-draw 53-bit canonical of
// std::uniform_real_distribution<double> (src/wave_types.cpp:664-669).
std::vector<double> random_phases(int n, int seed);

// Linear dispersion relation, Newton iteration exactly as src/wave_types.cpp:178-255. Throws std::runtime_error.
double wave_number(double omega, double water_depth, double g);

// Global cubic B-spline interpolation through `n_old` samples at parameters linspaced(n_old,0,1) with
// knot-averaged clamped knots, evaluated at linspaced(n_new,0,1): Eigen SplineFitting<Spline<double,6>>::
// Interpolate(vals,3,u) + spline(u_new) as used by IrregularWaves::ResampleIRF (src/wave_types.cpp:593-602).
// vals_in is [6][n_old] row-major; returns [6][n_new].
std::vector<double> resample_cubic_bspline6(const std::vector<double>& vals_in, int n_old, int n_new);

}  // namespace hc
