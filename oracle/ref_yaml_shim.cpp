// ref_yaml_shim.cpp -- C entry point around the REFERENCE's own hydro.yaml parser (test infrastructure only).
//
// src/hydro_yaml_parser.cpp is the one file of the reference that compiles standalone (std-only).  The Makefile
// compiles it from where it lies under /root/reference into oracle/_ref/libref_yaml.so together with this shim; no
// reference source is copied into the repository.  ref_yaml_dump() returns the parsed YAMLHydroData as a flat
// "key=value" text so the tests can compare the product's parser (hc_yaml_*) field by field.
#include <cstring>
#include <sstream>
#include <string>

#include "hydro_yaml_parser.h"  // from /root/reference/src (include path set by the Makefile)

static std::string g_out;

extern "C" const char* ref_yaml_dump(const char* path, int* status) {
    std::ostringstream o;
    o.precision(17);
    try {
        YAMLHydroData d = ReadHydroYAML(path);
        o << "nbodies=" << d.bodies.size() << "\n";
        for (size_t i = 0; i < d.bodies.size(); ++i) {
            const auto& b = d.bodies[i];
            o << "body" << i << ".name=" << b.name << "\n";
            o << "body" << i << ".h5_file=" << b.h5_file << "\n";
            o << "body" << i << ".include_excitation=" << (b.include_excitation ? 1 : 0) << "\n";
            o << "body" << i << ".include_radiation=" << (b.include_radiation ? 1 : 0) << "\n";
            o << "body" << i << ".radiation_calculation=" << b.radiation_calculation << "\n";
            o << "body" << i << ".radiation_convolution_mode=" << b.radiation_convolution_mode << "\n";
            o << "body" << i << ".td_smoothing=" << b.td_smoothing << "\n";
            o << "body" << i << ".td_window_length=" << b.td_window_length << "\n";
            o << "body" << i << ".td_rms_threshold_factor=" << b.td_rms_threshold_factor << "\n";
            o << "body" << i << ".td_taper_fraction_remaining=" << b.td_taper_fraction_remaining << "\n";
            o << "body" << i << ".td_export_plot_csv=" << (b.td_export_plot_csv ? 1 : 0) << "\n";
        }
        o << "waves.type=" << d.waves.type << "\n";
        o << "waves.height=" << d.waves.height << "\n";
        o << "waves.period=" << d.waves.period << "\n";
        o << "waves.direction=" << d.waves.direction << "\n";
        o << "waves.phase=" << d.waves.phase << "\n";
        o << "waves.spectrum=" << d.waves.spectrum << "\n";
        o << "waves.seed=" << d.waves.seed << "\n";
        o << "waves.period_values=";
        for (size_t i = 0; i < d.waves.period_values.size(); ++i) o << (i ? "," : "") << d.waves.period_values[i];
        o << "\n";
        o << "radiation_convolution_mode=" << d.radiation_convolution_mode << "\n";
        o << "td_smoothing=" << d.td_smoothing << "\n";
        o << "td_window_length=" << d.td_window_length << "\n";
        o << "td_rirf_end_time=" << d.td_rirf_end_time << "\n";
        o << "td_taper_start_percent=" << d.td_taper_start_percent << "\n";
        o << "td_taper_end_percent=" << d.td_taper_end_percent << "\n";
        o << "td_taper_final_amplitude=" << d.td_taper_final_amplitude << "\n";
        o << "td_export_plot_csv=" << (d.td_export_plot_csv ? 1 : 0) << "\n";
        *status = 0;
    } catch (const std::exception& e) {
        o.str("");
        o << e.what();
        *status = 1;
    }
    g_out = o.str();
    return g_out.c_str();
}
