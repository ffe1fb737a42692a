// hc_h5data.hpp -- what a BEMIO file holds for bodies 1..N, as read (nothing scaled); filled by libhc_bemio.so (hc_bemio_read),
// held by the hc_h5_* entry points of the C ABI (host side only: no device context is involved).
#pragma once
#include <string>
#include <vector>

struct hc_h5data {
    double rho = 0.0, g = 0.0, water_depth = 0.0;
    std::vector<double> w;  // simulation_parameters/w
    struct Body {
        double disp_vol = 0.0;
        double cg[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
        double lin[36] = {};
        std::vector<double> ainf;          // {6, 6N}
        std::vector<double> rirf_t, K;     // {S}, {6, 6N, S}
        std::vector<double> mag, phase;    // {6, 1, nw}
        std::vector<double> exc_t, exc_f;  // {L}, {6, 1, L}
    };
    std::vector<Body> bodies;
};
