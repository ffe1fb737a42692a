// hydroc_amd/hydro_types.h -- the parsed content of a hydro.yaml file, with the reference's type and member names
// (src/hydro_types.h:19-71) so that code written against YAMLHydroData keeps compiling.  Filled by ReadHydroYAML
// (hydroc_amd/hydro_yaml_parser.h), consumed by SetupHydroFromYAML (hydroc_amd/setup_hydro_from_yaml.h).
//
// Which members reach the GPU path (the reference's SetupHydroFromYAML reads exactly these, src/setup_hydro_from_yaml.cpp):
//   bodies[k].name       -> matched against the names of the multibody system's bodies, YAML order        (:84-122)
//   bodies[0].h5_file    -> the BEMIO file read for ALL matched bodies                                     (:93-97)
//   waves.type / height / period / phase / seed -> the wave model                                          (:28-79)
//   radiation_convolution_mode and the td_* members of YAMLHydroData -> convolution mode / TaperedDirect   (:151-190)
// Everything else is parsed and carried, as in the reference, but nothing on the force path looks at it.
#pragma once

#include <string>
#include <vector>

namespace hydroc_amd {

// One entry of `hydrodynamics: bodies:` (src/hydro_types.h:19-36).
struct HydroBody {
    std::string name;                                         // must equal ChBody::GetName() of the body it describes
    std::string h5_file;                                      // BEMIO-HDF5 file (resolved relative to the yaml file by the parser)
    bool include_excitation                = true;            // carried only
    bool include_radiation                 = true;            // carried only
    std::string radiation_calculation      = "convolution";   // carried only ("state_space" is not implemented by the reference either)
    std::string radiation_convolution_mode = "Baseline";      // per-body copies of the system-wide settings below: carried only
    std::string td_smoothing               = "sg";
    int td_window_length                   = 5;
    double td_rms_threshold_factor         = 0.02;
    double td_taper_fraction_remaining     = 0.25;
    bool td_export_plot_csv                = false;
};

// `hydrodynamics: waves:` (src/hydro_types.h:41-53).
struct WaveSettings {
    std::string type     = "regular";            // "regular" | "irregular" | "no_wave" ("still", "still_ci"); case-insensitive
    double height        = 0.0;                  // regular: amplitude = height / 2; irregular: significant height Hs
    double period        = 0.0;                  // regular: omega = 2 pi / period; irregular: peak period Tp
    double direction     = 0.0;                  // degrees; carried only (the force path evaluates eta at x = 0)
    double phase         = 0.0;                  // -> RegularWave::regular_wave_phase_, which the force does not use
    std::string spectrum = "pierson_moskowitz";  // carried only: YAML irregular waves are always Pierson-Moskowitz (gamma = 1)
    int seed             = -1;                   // irregular: mt19937 seed, <= 0 -> 1
    std::vector<double> period_values;           // expanded sweep of `period` (values / linspace / range); carried only
};

// Everything ReadHydroYAML returns (src/hydro_types.h:58-71).
struct YAMLHydroData {
    std::vector<HydroBody> bodies;
    WaveSettings waves;
    std::string radiation_convolution_mode = "Baseline";  // "Baseline" | "TaperedDirect" (case-insensitive)
    std::string td_smoothing               = "sg";        // "sg" (Savitzky-Golay, 5 points) | "moving_average"
    int td_window_length                   = 5;           // moving-average window; made odd and >= 3 by SetupHydroFromYAML
    double td_rirf_end_time                = -1.0;        // truncate the IRF at this time (s); < 0: keep all of it
    double td_taper_start_percent          = 0.8;         // half-cosine taper from this fraction of the series ...
    double td_taper_end_percent            = 1.0;         // ... to this one, zero behind it
    double td_taper_final_amplitude        = 0.0;         // amplitude the taper ends at (fraction of the original)
    bool td_export_plot_csv                = false;       // rirf_body<b>_summary.csv into the diagnostics directory
};

}  // namespace hydroc_amd
