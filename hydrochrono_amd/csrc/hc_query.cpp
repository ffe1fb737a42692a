// hc_query.cpp -- introspection entry points of the C ABI (profile, sizes, processed kernels, wave tables) and the host-math exports
// of include/hydrochrono_amd_host.h.
#include "hc_internal.hpp"

using namespace hc::detail;

extern "C" {

// ---- introspection ----------------------------------------------------------------------------
int hc_enable_profiling(hc_ctx* c, int on) {
    HC_API_BEGIN(c)
    if (!on) profile_drain(c);
    c->profiling       = on != 0;
    c->profile_stride  = on > 1 ? on : 1;
    c->profile_counter = 0;
    HC_API_END(c)
}

int hc_get_profile(hc_ctx* c, hc_profile_stats* out) {
    HC_API_BEGIN(c)
    require(out, HC_ERR_INVALID, "null pointer");
    profile_drain(c);
    *out = c->prof;
    HC_API_END(c)
}

int hc_reset_profile(hc_ctx* c) {
    HC_API_BEGIN(c)
    profile_drain(c);
    const double bytes = c->prof.conv_kernel_bytes, bbytes = c->prof.block_kernel_bytes, obytes = c->prof.block_kernel_bytes_once;
    c->prof = hc_profile_stats{};
    c->prof.conv_kernel_bytes       = bytes;
    c->prof.block_kernel_bytes      = bbytes;
    c->prof.block_kernel_bytes_once = obytes;
    HC_API_END(c)
}

int hc_get_sizes(hc_ctx* c, int* N, int* n_local, int* S, int* L, int* nf, int* nt, int* H, int* Hcap) {
    HC_API_BEGIN(c)
    if (N) *N = c->N;
    if (n_local) *n_local = c->nloc;
    if (S) *S = c->S;
    if (L) *L = c->L;
    if (nf) *nf = c->nf;
    if (nt) *nt = c->nt;
    if (H) *H = static_cast<int>(c->times.size());
    if (Hcap) *Hcap = c->Hcap;
    HC_API_END(c)
}

int hc_get_rirf_width(hc_ctx* c, double* w) {
    HC_API_BEGIN(c)
    require(c->finalized && w, HC_ERR_INVALID, "not finalized or null pointer");
    std::copy(c->width.begin(), c->width.end(), w);
    HC_API_END(c)
}

int hc_get_rirf_effective(hc_ctx* c, double* out) {
    HC_API_BEGIN(c)
    require(c->finalized && out, HC_ERR_INVALID, "not finalized or null pointer");
    ensure_processed(c);
    const size_t n = static_cast<size_t>(c->Dloc) * c->D * c->S;
    hc::DeviceBuffer<double> tmp;
    tmp.alloc(n);
    hc::launch_unrelayout(rad_panel(c), c->Dloc, c->D, c->S, tmp.p, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipMemcpyAsync(out, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

int hc_get_rirf_value(hc_ctx* c, int row_local, int col, int st, double* out) {
    HC_API_BEGIN(c)
    require(c->finalized && out, HC_ERR_INVALID, "not finalized or null pointer");
    // index guards of TestHydro::GetRIRFval (src/hydro_forces.cpp:694-697): std::out_of_range there
    require(row_local >= 0 && row_local < c->Dloc && col >= 0 && col < c->D && st >= 0 && st < c->S, HC_ERR_OUT_OF_RANGE,
            "rirf index out of range in GetRIRFval");
    ensure_processed(c);
    hc::DeviceBuffer<double> series;
    series.alloc(static_cast<size_t>(c->S));
    hc::launch_extract_series(rad_panel(c), row_local, col, c->D, c->S, series.p, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipMemcpyAsync(out, series.p + st, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

int hc_get_excitation_irf_resampled(hc_ctx* c, int body, double* t, double* width, double* vals) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular, HC_ERR_INVALID, "no irregular wave model attached");
    check_body(c, body);
    require(is_local(c, body), HC_ERR_INVALID, "body is not owned by this context");
    const hc::ExGroup& g = c->ex_groups[c->ex_group_of[body]];
    if (t) std::copy(c->ex_tau.begin() + g.off, c->ex_tau.begin() + g.off + g.L, t);
    if (width) std::copy(c->ex_width.begin() + g.off, c->ex_width.begin() + g.off + g.L, width);
    if (vals)
        for (int d = 0; d < 6; ++d) {
            const size_t off = static_cast<size_t>(6 * (body - c->b0) + d) * c->L + g.off;
            std::copy(c->ex_vals.begin() + off, c->ex_vals.begin() + off + g.L, vals + static_cast<size_t>(d) * g.L);
        }
    HC_API_END(c)
}

int hc_get_excitation_irf_size(hc_ctx* c, int body, int* L) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular && L, HC_ERR_INVALID, "no irregular wave model attached, or null pointer");
    check_body(c, body);
    require(c->ex_group_of[body] >= 0, HC_ERR_INVALID, "no excitation IRF was ingested for this body");
    *L = c->ex_groups[c->ex_group_of[body]].L;
    HC_API_END(c)
}

int hc_get_shard(hc_ctx* c, int* body_begin, int* body_end) {
    if (!c) return HC_ERR_INVALID;
    if (body_begin) *body_begin = c->b0;
    if (body_end) *body_end = c->b1;
    return HC_OK;
}

int hc_get_spectrum(hc_ctx* c, double* f, double* S, double* df, double* phase, double* k) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    if (f) std::copy(c->spec_f.begin(), c->spec_f.end(), f);
    if (S) std::copy(c->spec_S.begin(), c->spec_S.end(), S);
    if (df) std::copy(c->spec_df.begin(), c->spec_df.end(), df);
    if (phase) std::copy(c->spec_phase.begin(), c->spec_phase.end(), phase);
    if (k) std::copy(c->spec_k.begin(), c->spec_k.end(), k);
    HC_API_END(c)
}

int hc_get_eta_table(hc_ctx* c, double* t, double* eta) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    if (t) std::copy(c->eta_t.begin(), c->eta_t.end(), t);
    if (eta) std::copy(c->eta.begin(), c->eta.end(), eta);
    HC_API_END(c)
}

int hc_get_regular_coeffs(hc_ctx* c, double* mag, double* phase, double* wavenumber) {
    HC_API_BEGIN(c)
    require(c->wave_kind == hc::kWaveRegular, HC_ERR_INVALID, "no regular wave model attached");
    if (mag) std::copy(c->reg_mag.begin(), c->reg_mag.end(), mag);
    if (phase) std::copy(c->reg_phase.begin(), c->reg_phase.end(), phase);
    if (wavenumber) *wavenumber = c->reg_wavenumber;
    HC_API_END(c)
}

// Diagnostics (not part of the public header): copies an internal device buffer to the host.  which: 0 = P [16][Dpad],
// 1 = E [16][Dpad], 2 = Y [16][kScatterSamples][Dpad].
int hc_debug_read(hc_ctx* c, int which, double* out, long long n) {
    HC_API_BEGIN(c)
    HC_HIP(hipDeviceSynchronize());
    const hc::DeviceBuffer<double>& b = which == 0 ? c->d_P : (which == 1 ? c->d_E : c->d_Y);
    require(n >= 0 && static_cast<size_t>(n) <= b.n, HC_ERR_INVALID, "bad size");
    HC_HIP(hipMemcpy(out, b.p, static_cast<size_t>(n) * sizeof(double), hipMemcpyDeviceToHost));
    HC_API_END(c)
}

}  // extern "C"

// =================================================================================================
// include/hydrochrono_amd_host.h
// =================================================================================================
#include "../../include/hydrochrono_amd_host.h"

extern "C" {

void hc_host_linspaced(int n, double lo, double hi, double* out) {
    const auto v = hc::linspaced(n, lo, hi);
    std::copy(v.begin(), v.end(), out);
}
void hc_host_trapezoid_widths(const double* grid, int n, double* out) {
    const auto v = hc::trapezoid_widths(std::vector<double>(grid, grid + n));
    std::copy(v.begin(), v.end(), out);
}
void hc_host_jonswap_spectrum_hz(const double* f, int n, double Hs, double Tp, double gamma, int is_normalized, double* out) {
    const auto v = hc::jonswap_spectrum_hz(std::vector<double>(f, f + n), Hs, Tp, gamma, is_normalized != 0);
    std::copy(v.begin(), v.end(), out);
}
void hc_host_random_phases(int n, int seed, double* out) {
    const auto v = hc::random_phases(n, seed);
    std::copy(v.begin(), v.end(), out);
}
double hc_host_wave_number(double omega, double water_depth, double g) {
    try {
        return hc::wave_number(omega, water_depth, g);
    } catch (...) {
        return std::numeric_limits<double>::quiet_NaN();
    }
}
int hc_host_resample_irf(const double* vals, int n_old, int n_new, double* out) {
    try {
        const auto v = hc::resample_cubic_bspline6(std::vector<double>(vals, vals + static_cast<size_t>(6) * n_old), n_old, n_new);
        std::copy(v.begin(), v.end(), out);
    } catch (...) {
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

}  // extern "C"

