// hydroc_amd/hydro_types.h -- the parsed content of a hydro.yaml file, with the reference's type and member names
// (src/hydro_types.h:19-71) so that code written against YAMLHydroData keeps compiling.  Filled by ReadHydroYAML
// (hydroc_amd/hydro_yaml_parser.h), consumed by SetupHydroFromYAML (hydroc_amd/setup_hydro_from_yaml.h).
#pragma once

#include <string>
#include <vector>

namespace hydroc_amd {

struct HydroBody {  // one entry of `bodies:` (src/hydro_types.h:19-36)
    std::string name;
    std::string h5_file;
    bool include_excitation                = true;
    bool include_radiation                 = true;
    std::string radiation_calculation      = "convolution";
    std::string radiation_convolution_mode = "Baseline";
    std::string td_smoothing               = "sg";
    int td_window_length                   = 5;
    double td_rms_threshold_factor         = 0.02;
    double td_taper_fraction_remaining     = 0.25;
    bool td_export_plot_csv                = false;
};

struct WaveSettings {  // `waves:` (src/hydro_types.h:41-53)
    std::string type     = "regular";  // "regular", "irregular", "no_wave" ("still", "still_ci")
    double height        = 0.0;
    double period        = 0.0;
    double direction     = 0.0;
    double phase         = 0.0;
    std::string spectrum = "pierson_moskowitz";
    int seed             = -1;
    std::vector<double> period_values;  // expanded sweep of `period`
};

struct YAMLHydroData {  // src/hydro_types.h:58-71
    std::vector<HydroBody> bodies;
    WaveSettings waves;
    std::string radiation_convolution_mode = "Baseline";  // Baseline | TaperedDirect
    std::string td_smoothing               = "sg";
    int td_window_length                   = 5;
    double td_rirf_end_time                = -1.0;
    double td_taper_start_percent          = 0.8;
    double td_taper_end_percent            = 1.0;
    double td_taper_final_amplitude        = 0.0;
    bool td_export_plot_csv                = false;
};

}  // namespace hydroc_amd
