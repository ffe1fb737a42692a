// hc_host_math.cpp -- see hc_host_math.hpp.
#include "hc_host_math.hpp"

#include <algorithm>
#include <cmath>
#include <random>
#include <stdexcept>

namespace hc {

static const double kTwoPi = 6.283185307179586476925286766559;

std::vector<double> linspaced(int n, double lo, double hi) {
    std::vector<double> out;
    if (n <= 0) return out;
    out.resize(n);
    if (n == 1) {
        out[0] = hi;
        return out;
    }
    const double step = (hi - lo) / static_cast<double>(n - 1);
    if (std::fabs(hi) < std::fabs(lo)) {
        // generated from the high end so that the value of smaller magnitude is the exact one
        for (int i = 1; i < n; ++i) out[i] = hi - static_cast<double>(n - 1 - i) * step;
        out[0] = lo;
    } else {
        for (int i = 0; i < n - 1; ++i) out[i] = lo + static_cast<double>(i) * step;
        out[n - 1] = hi;
    }
    return out;
}

std::vector<double> trapezoid_widths(const std::vector<double>& grid) {
    const size_t n = grid.size();
    std::vector<double> w(n, 0.0);
    for (size_t i = 0; i + 1 < n; ++i) {
        const double half = 0.5 * std::fabs(grid[i + 1] - grid[i]);
        w[i] += half;  // right half-cell of sample i   (reference adds the right neighbour first)
    }
    for (size_t i = 1; i < n; ++i) w[i] += 0.5 * std::fabs(grid[i] - grid[i - 1]);
    return w;
}

std::vector<double> jonswap_spectrum_hz(const std::vector<double>& f, double Hs, double Tp, double gamma, bool normalized) {
    std::vector<double> S(f.size());
    const double inv_tp4 = std::pow(1 / Tp, 4);
    const double hs2     = std::pow(Hs / 2, 2);
    const double norm    = 1 - 0.287 * std::log(gamma);
    for (size_t i = 0; i < f.size(); ++i) {
        const double fi = f[i];
        double s        = 1.25 * inv_tp4 * hs2 * std::pow(fi, -5) * std::exp(-1.25 * inv_tp4 * std::pow(fi, -4));
        const double sigma = (fi <= 1.0 / Tp) ? 0.07 : 0.09;
        s *= std::pow(gamma, std::exp(-(1.0 / (2.0 * std::pow(sigma, 2))) * std::pow(fi * Tp - 1.0, 2)));
        if (normalized) s *= norm;
        S[i] = s;
    }
    return S;
}

std::vector<double> random_phases(int n, int seed) {
    // std::mt19937's output sequence is fixed by the standard; the real distribution is not, so the
    // 53-bit canonical (two 32-bit draws, low word first, divided by 2^64) is spelled out.
    std::mt19937 engine(static_cast<std::mt19937::result_type>(seed));
    std::vector<double> ph(std::max(n, 0));
    for (int i = 0; i < n; ++i) {
        const double lo = static_cast<double>(engine());
        const double hi = static_cast<double>(engine());
        double u        = (lo + hi * 4294967296.0) / 18446744073709551616.0;
        if (u >= 1.0) u = std::nextafter(1.0, 0.0);
        ph[i] = kTwoPi * u + 0.0;
    }
    return ph;
}

double wave_number(double omega, double water_depth, double g) {
    const double tolerance   = 1e-6;
    const int max_iterations = 100;
    if (omega <= 0.0) throw std::runtime_error("Angular frequency must be positive.");
    if (water_depth < 0.0) throw std::runtime_error("Water depth cannot be negative.");
    if (g <= 0.0) throw std::runtime_error("Gravity must be positive.");
    const double k_deep = omega * omega / g;
    if (water_depth == 0.0 || water_depth > 1000.0 || std::isinf(water_depth)) return k_deep;
    double k = k_deep;
    for (int it = 0; it < max_iterations; ++it) {
        const double th = std::tanh(k * water_depth);
        const double f  = omega * omega - g * k * th;
        // the reference's derivative carries a factor 2 on the first term (src/wave_types.cpp:227); kept,
        // because the stop rule |dk| <= 1e-6 makes the iterate it stops at depend on it
        const double df = -2.0 * g * th - g * k * water_depth * (1.0 - th * th);
        if (std::fabs(df) < tolerance) throw std::runtime_error("Numerical instability: derivative too close to zero.");
        const double dk = f / df;
        k -= dk;
        if (!(std::fabs(dk) > tolerance)) {
            if (it + 1 >= max_iterations) break;  // reference reports non-convergence when the cap is hit
            return k;
        }
    }
    throw std::runtime_error("Failed to converge within maximum iterations.");
}

namespace {

// Clamped cubic B-spline machinery on parameters u_i = i/(n-1).
struct Cubic {
    std::vector<double> knots;
    int n;

    explicit Cubic(const std::vector<double>& u) : n(static_cast<int>(u.size())) {
        knots.assign(n + 4, 0.0);
        for (int j = 1; j < n - 3; ++j) knots[j + 3] = (u[j] + u[j + 1] + u[j + 2]) / 3.0;
        for (int j = 0; j < 4; ++j) knots[n + j] = 1.0;
    }
    // index of the knot span containing x (spans are [knots[i], knots[i+1])), clamped to [3, n-1]
    int span(double x) const {
        if (x <= knots[0]) return 3;
        auto first = knots.begin() + 2;
        auto last  = knots.end() - 4;
        return static_cast<int>(std::upper_bound(first, last, x) - knots.begin()) - 1;
    }
    // the four cubic basis functions that are non-zero on `sp`, by the Cox-de Boor triangle
    void basis(double x, int sp, double b[4]) const {
        double dl[4], dr[4];
        b[0] = 1.0;
        for (int d = 1; d <= 3; ++d) {
            dl[d]      = x - knots[sp + 1 - d];
            dr[d]      = knots[sp + d] - x;
            double acc = 0.0;
            for (int r = 0; r < d; ++r) {
                const double q = b[r] / (dr[r + 1] + dl[d - r]);
                b[r]           = acc + dr[r + 1] * q;
                acc            = dl[d - r] * q;
            }
            b[d] = acc;
        }
    }
};

}  // namespace

std::vector<double> resample_cubic_bspline6(const std::vector<double>& vals_in, int n_old, int n_new) {
    if (n_old < 4) throw std::runtime_error("excitation IRF needs at least 4 samples for cubic resampling");
    if (static_cast<int>(vals_in.size()) != 6 * n_old) throw std::runtime_error("excitation IRF has wrong shape");
    const std::vector<double> u_old = linspaced(n_old, 0.0, 1.0);
    const std::vector<double> u_new = linspaced(n_new, 0.0, 1.0);
    Cubic sp(u_old);

    // Collocation system  sum_k N_k(u_i) c_k = y_i.  Row i touches columns first[i]..first[i]+3 only.
    // Stored as 4 entries per row and eliminated forward without pivoting: the collocation matrix of a
    // B-spline basis at Schoenberg-Whitney-admissible nodes is totally positive, for which elimination
    // without row exchanges is backward stable (de Boor & Pinkus 1977).
    const int n = n_old;
    std::vector<int> first(n);
    std::vector<double> band(static_cast<size_t>(n) * 4, 0.0);
    first[0]     = 0;
    band[0]      = 1.0;
    first[n - 1] = n - 4;
    band[static_cast<size_t>(n - 1) * 4 + 3] = 1.0;
    for (int i = 1; i < n - 1; ++i) {
        const int s = sp.span(u_old[i]);
        first[i]    = s - 3;
        sp.basis(u_old[i], s, &band[static_cast<size_t>(i) * 4]);
    }
    // dense-in-band working copy: column offset relative to the diagonal, width 7 ([-3, +3])
    const int W = 7;
    std::vector<double> M(static_cast<size_t>(n) * W, 0.0);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) {
            const int col = first[i] + k;
            const int off = col - i + 3;
            if (off >= 0 && off < W) M[static_cast<size_t>(i) * W + off] = band[static_cast<size_t>(i) * 4 + k];
        }
    std::vector<double> rhs(vals_in);  // [6][n]
    for (int c = 0; c < n; ++c) {
        const double piv = M[static_cast<size_t>(c) * W + 3];
        if (piv == 0.0) throw std::runtime_error("singular spline collocation matrix");
        const int rmax = std::min(n - 1, c + 3);
        for (int r = c + 1; r <= rmax; ++r) {
            const int off_rc = c - r + 3;
            const double m   = M[static_cast<size_t>(r) * W + off_rc] / piv;
            if (m == 0.0) continue;
            const int jmax = std::min(n - 1, c + 3);
            for (int j = c; j <= jmax; ++j) M[static_cast<size_t>(r) * W + (j - r + 3)] -= m * M[static_cast<size_t>(c) * W + (j - c + 3)];
            for (int d = 0; d < 6; ++d) rhs[static_cast<size_t>(d) * n + r] -= m * rhs[static_cast<size_t>(d) * n + c];
        }
    }
    std::vector<double> ctrl(static_cast<size_t>(6) * n);
    for (int d = 0; d < 6; ++d) {
        for (int i = n - 1; i >= 0; --i) {
            double s       = rhs[static_cast<size_t>(d) * n + i];
            const int jmax = std::min(n - 1, i + 3);
            for (int j = i + 1; j <= jmax; ++j) s -= M[static_cast<size_t>(i) * W + (j - i + 3)] * ctrl[static_cast<size_t>(d) * n + j];
            ctrl[static_cast<size_t>(d) * n + i] = s / M[static_cast<size_t>(i) * W + 3];
        }
    }
    std::vector<double> out(static_cast<size_t>(6) * n_new);
    for (int i = 0; i < n_new; ++i) {
        const int s = sp.span(u_new[i]);
        double b[4];
        sp.basis(u_new[i], s, b);
        for (int d = 0; d < 6; ++d) {
            const double* c = &ctrl[static_cast<size_t>(d) * n + (s - 3)];
            out[static_cast<size_t>(d) * n_new + i] = c[0] * b[0] + c[1] * b[1] + c[2] * b[2] + c[3] * b[3];
        }
    }
    return out;
}

}  // namespace hc
