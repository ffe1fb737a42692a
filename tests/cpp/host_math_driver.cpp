#include <cstdio>
#include <vector>
#include "hc_host_math.hpp"
int main() {
    auto a = hc::linspaced(7, -3.0, 2.0); hc::linspaced(1, 0, 1); hc::linspaced(0, 0, 1); hc::linspaced(5, 4.0, -1.0);
    hc::trapezoid_widths(a); hc::trapezoid_widths({1.0}); hc::trapezoid_widths({});
    auto f = hc::linspaced(64, 0.02, 0.5);
    hc::jonswap_spectrum_hz(f, 2.0, 8.0, 3.3, false); hc::jonswap_spectrum_hz(f, 2.0, 8.0, 1.0, true);
    hc::random_phases(64, 1); hc::random_phases(0, 7);
    for (double h : {0.0, 50.0, 2000.0}) for (double om : {0.1, 0.5, 1.0, 2.0, 6.0}) { volatile double k = hc::wave_number(om, h, 9.81); (void)k; }
    std::vector<double> vals(6 * 101); for (size_t i = 0; i < vals.size(); ++i) vals[i] = 0.01 * i * ((i % 7) - 3);
    hc::resample_cubic_bspline6(vals, 101, 37); hc::resample_cubic_bspline6(vals, 101, 1000);
    std::vector<double> v5(30, 1.5); hc::resample_cubic_bspline6(v5, 5, 300); 
    try { hc::resample_cubic_bspline6(std::vector<double>(18, 1.0), 3, 10); } catch (...) { std::puts("n_old=3 rejected"); }
    std::puts("host math ok");
}
