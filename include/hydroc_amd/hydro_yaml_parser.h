// hydroc_amd/hydro_yaml_parser.h -- ReadHydroYAML(path) of the reference (src/hydro_yaml_parser.h:20, .cpp:154-610): the file is
// parsed by the library (hc_yaml_read, include/hydrochrono_amd_yaml.h -- checked field by field against the reference's own parser
// compiled from its source, tests/test_hydro_yaml.py) and copied into the reference's result type.  Throws std::runtime_error
// where the reference does (unreadable file, malformed content).
#pragma once

#include <stdexcept>
#include <string>

#include "../hydrochrono_amd_yaml.h"
#include "hydro_types.h"

namespace hydroc_amd {

inline YAMLHydroData ReadHydroYAML(const std::string& hydro_file_path) {
    char err[1024] = {0};
    hc_yaml* cfg   = nullptr;
    if (hc_yaml_read(hydro_file_path.c_str(), &cfg, err, sizeof err) != HC_OK) throw std::runtime_error(err);
    struct Release {
        hc_yaml* p;
        ~Release() { hc_yaml_free(p); }
    } release{cfg};
    auto text = [](const char* s) { return std::string(s ? s : ""); };
    YAMLHydroData data;
    const int nb = hc_yaml_num_bodies(cfg);
    data.bodies.resize(nb);
    for (int b = 0; b < nb; ++b) {
        HydroBody& body                 = data.bodies[b];
        body.name                       = text(hc_yaml_body_string(cfg, b, "name"));
        body.h5_file                    = text(hc_yaml_body_string(cfg, b, "h5_file"));
        body.radiation_calculation      = text(hc_yaml_body_string(cfg, b, "radiation_calculation"));
        body.radiation_convolution_mode = text(hc_yaml_body_string(cfg, b, "radiation_convolution_mode"));
        body.td_smoothing               = text(hc_yaml_body_string(cfg, b, "td_smoothing"));
        body.include_excitation         = hc_yaml_body_number(cfg, b, "include_excitation") != 0.0;
        body.include_radiation          = hc_yaml_body_number(cfg, b, "include_radiation") != 0.0;
        body.td_export_plot_csv         = hc_yaml_body_number(cfg, b, "td_export_plot_csv") != 0.0;
        body.td_window_length           = static_cast<int>(hc_yaml_body_number(cfg, b, "td_window_length"));
        body.td_rms_threshold_factor    = hc_yaml_body_number(cfg, b, "td_rms_threshold_factor");
        body.td_taper_fraction_remaining = hc_yaml_body_number(cfg, b, "td_taper_fraction_remaining");
    }
    data.waves.type      = text(hc_yaml_string(cfg, "waves.type"));
    data.waves.spectrum  = text(hc_yaml_string(cfg, "waves.spectrum"));
    data.waves.height    = hc_yaml_number(cfg, "waves.height");
    data.waves.period    = hc_yaml_number(cfg, "waves.period");
    data.waves.direction = hc_yaml_number(cfg, "waves.direction");
    data.waves.phase     = hc_yaml_number(cfg, "waves.phase");
    data.waves.seed      = static_cast<int>(hc_yaml_number(cfg, "waves.seed"));
    data.waves.period_values.resize(hc_yaml_period_values(cfg, nullptr, 0));
    if (!data.waves.period_values.empty())
        hc_yaml_period_values(cfg, data.waves.period_values.data(), static_cast<int>(data.waves.period_values.size()));
    data.radiation_convolution_mode = text(hc_yaml_string(cfg, "radiation_convolution_mode"));
    data.td_smoothing               = text(hc_yaml_string(cfg, "td_smoothing"));
    data.td_window_length           = static_cast<int>(hc_yaml_number(cfg, "td_window_length"));
    data.td_rirf_end_time           = hc_yaml_number(cfg, "td_rirf_end_time");
    data.td_taper_start_percent     = hc_yaml_number(cfg, "td_taper_start_percent");
    data.td_taper_end_percent       = hc_yaml_number(cfg, "td_taper_end_percent");
    data.td_taper_final_amplitude   = hc_yaml_number(cfg, "td_taper_final_amplitude");
    data.td_export_plot_csv         = hc_yaml_number(cfg, "td_export_plot_csv") != 0.0;
    return data;
}

// "Optional helper to parse convolution mode string" of the reference's header (src/hydro_yaml_parser.h:23-26)
enum class RadiationConvolutionModeParsed { Baseline, TaperedDirect };

}  // namespace hydroc_amd
