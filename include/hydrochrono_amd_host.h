/* hydrochrono_amd_host.h -- host-only init-time utilities of the hydro-force path (no GPU needed).
 *
 * The same routines hc_set_wave_irregular() uses internally, exported so that callers that only need the
 * derived inputs (e.g. a result exporter writing frequencies / spectral densities, cf.
 * src/simulation_exporter.cpp:365-393) or CPU-side tests can reach them.  All arrays are caller-allocated.
 */
#ifndef HYDROCHRONO_AMD_HOST_H
#define HYDROCHRONO_AMD_HOST_H

#ifdef __cplusplus
extern "C" {
#endif

/* Eigen::VectorXd::LinSpaced(n, lo, hi) as used at src/wave_types.cpp:584,594-595,654,737 */
void hc_host_linspaced(int n, double lo, double hi, double* out_n);
/* trapezoid half-widths of a grid: src/hydro_forces.cpp:181-190, GetWidthArray src/wave_types.cpp:608-620 */
void hc_host_trapezoid_widths(const double* grid, int n, double* out_n);
/* JONSWAPSpectrumHz / PiersonMoskowitzSpectrumHz, src/wave_types.cpp:679-715 (f ascending) */
void hc_host_jonswap_spectrum_hz(const double* f, int n, double Hs, double Tp, double gamma, int is_normalized, double* out_n);
/* std::mt19937(seed) + uniform_real_distribution<double>(0, 2*pi), src/wave_types.cpp:664-669 */
void hc_host_random_phases(int n, int seed, double* out_n);
/* ComputeWaveNumber, src/wave_types.cpp:178-255; returns NaN where the reference throws */
double hc_host_wave_number(double omega, double water_depth, double g);
/* IrregularWaves::ResampleIRF value resampling, src/wave_types.cpp:593-602: vals [6][n_old] -> out [6][n_new];
 * returns 0 on success */
int hc_host_resample_irf(const double* vals_6xn_old, int n_old, int n_new, double* out_6xn_new);

#ifdef __cplusplus
}
#endif
#endif
