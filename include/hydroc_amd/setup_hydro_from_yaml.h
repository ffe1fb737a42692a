// hydroc_amd/setup_hydro_from_yaml.h -- SetupHydroFromYAML of the reference (src/setup_hydro_from_yaml.h:33-39, .cpp:28-193):
// parsed hydro.yaml + the bodies of the multibody system -> an initialised TestHydro.  With ChBody arguments the result is wired
// into the bodies' ChSystem (forces, added-mass load), as the reference's is; the runner's lines
//
//     YAMLHydroData hydro_data = ReadHydroYAML(hydro_file);
//     auto test_hydro = SetupHydroFromYAML(hydro_data, bodies, loop_dt, sim_duration_hint, 0.0);
//
// (src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp:454-457) compile unchanged.  Addition: an optional trailing device list
// (one body-row shard per listed GPU) or a single device id.
#pragma once

#include <algorithm>
#include <cctype>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "hydro_forces.h"
#include "hydro_types.h"
#include "hydro_yaml_parser.h"

namespace hydroc_amd {
namespace yaml_setup {

inline std::string lowercase(std::string s) {
    for (char& ch : s) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    return s;
}

// CreateWaveFromSettings (src/setup_hydro_from_yaml.cpp:28-79): regular -> A = height / 2, omega = 2 pi / period; irregular ->
// IrregularWaveParams defaults (Pierson-Moskowitz, f in [0.001, 1] Hz) with the YAML height / period and seed (<= 0 -> 1);
// no_wave / still_ci / still -> NoWave(N); anything else is an error
inline std::shared_ptr<WaveBase> wave_from_settings(const WaveSettings& ws, unsigned int num_bodies, double timestep, double sim_duration,
                                                    double ramp_duration) {
    const std::string type = lowercase(ws.type);
    if (type == "regular") {
        auto w                     = std::make_shared<RegularWave>(num_bodies);
        w->regular_wave_amplitude_ = ws.height / 2.0;
        w->regular_wave_omega_     = 2.0 * 3.14159265358979323846 / ws.period;
        w->regular_wave_phase_     = ws.phase;
        return w;
    }
    if (type == "irregular") {
        IrregularWaveParams p;
        p.num_bodies_          = num_bodies;
        p.simulation_dt_       = timestep;
        p.simulation_duration_ = sim_duration;
        p.ramp_duration_       = ramp_duration;
        p.wave_height_         = ws.height;
        p.wave_period_         = ws.period;
        p.seed_                = ws.seed > 0 ? ws.seed : 1;
        return std::make_shared<IrregularWaves>(p);
    }
    if (type == "no_wave" || type == "still_ci" || type == "still") return std::make_shared<NoWave>(num_bodies);
    throw std::runtime_error("Unsupported wave type: " + ws.type);
}

// MatchBodiesByName (:84-122): YAML order; the name of a body is whatever GetName() of the element type returns
template <class BodyPtr>
std::vector<BodyPtr> match_bodies(const std::vector<HydroBody>& hydro_bodies, const std::vector<BodyPtr>& system_bodies) {
    std::vector<BodyPtr> matched;
    for (const HydroBody& hb : hydro_bodies) {
        auto it = std::find_if(system_bodies.begin(), system_bodies.end(), [&](const BodyPtr& b) { return b->GetName() == hb.name; });
        if (it != system_bodies.end()) matched.push_back(*it);  // (the reference logs a warning for a YAML body the system lacks)
    }
    return matched;
}

// system-wide convolution settings (:151-190)
inline void apply_convolution_settings(TestHydro& hydro, const YAMLHydroData& d) {
    if (lowercase(d.radiation_convolution_mode) != "tapereddirect") {
        hydro.SetRadiationConvolutionMode(TestHydro::RadiationConvolutionMode::Baseline);
        return;
    }
    hydro.SetRadiationConvolutionMode(TestHydro::RadiationConvolutionMode::TaperedDirect);
    TestHydro::TaperedDirectOptions o;
    if (!d.td_smoothing.empty()) o.smoothing = d.td_smoothing;
    o.window_length = std::max(3, d.td_window_length != 0 ? d.td_window_length : o.window_length);
    o.window_length += (o.window_length % 2 == 0) ? 1 : 0;  // odd
    o.rirf_end_time         = d.td_rirf_end_time;
    o.taper_start_percent   = d.td_taper_start_percent;
    o.taper_end_percent     = d.td_taper_end_percent;
    o.taper_final_amplitude = d.td_taper_final_amplitude;
    o.export_plot_csv       = d.td_export_plot_csv;
    hydro.SetTaperedDirectOptions(o);
}

template <class BodyPtr>
std::unique_ptr<TestHydro> setup(const YAMLHydroData& hydro_data, const std::vector<BodyPtr>& bodies, double timestep, double sim_duration,
                                 double ramp_duration, const std::vector<int>& device_ids) {
    auto matched = match_bodies(hydro_data.bodies, bodies);
    if (matched.empty()) throw std::runtime_error("No hydrodynamic bodies found in Chrono system");
    const std::string h5_file_path = hydro_data.bodies.front().h5_file;  // the first body's file serves all bodies (:93-97)
    auto wave = wave_from_settings(hydro_data.waves, static_cast<unsigned int>(matched.size()), timestep, sim_duration, ramp_duration);
    auto test_hydro = std::make_unique<TestHydro>(std::move(matched), h5_file_path, std::move(wave), device_ids);
    apply_convolution_settings(*test_hydro, hydro_data);
    return test_hydro;
}

}  // namespace yaml_setup

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
// The reference's signature (+ the optional device list / device id).
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const YAMLHydroData& hydro_data, const std::vector<std::shared_ptr<chrono::ChBody>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration,
                                                     const std::vector<int>& device_ids = {0}) {
    return yaml_setup::setup(hydro_data, bodies, timestep, sim_duration, ramp_duration, device_ids);
}
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const YAMLHydroData& hydro_data, const std::vector<std::shared_ptr<chrono::ChBody>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration, int device_id) {
    return yaml_setup::setup(hydro_data, bodies, timestep, sim_duration, ramp_duration, std::vector<int>{device_id});
}
#endif

// Chrono-free drivers: the same through the BodyView interface
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const YAMLHydroData& hydro_data, const std::vector<std::shared_ptr<BodyView>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration,
                                                     const std::vector<int>& device_ids = {0}) {
    return yaml_setup::setup(hydro_data, bodies, timestep, sim_duration, ramp_duration, device_ids);
}
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const YAMLHydroData& hydro_data, const std::vector<std::shared_ptr<BodyView>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration, int device_id) {
    return yaml_setup::setup(hydro_data, bodies, timestep, sim_duration, ramp_duration, std::vector<int>{device_id});
}
// Round-3 forms with the file path in place of the parsed data (ReadHydroYAML folded in)
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const std::string& hydro_yaml_path, const std::vector<std::shared_ptr<BodyView>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration,
                                                     const std::vector<int>& device_ids = {0}) {
    return yaml_setup::setup(ReadHydroYAML(hydro_yaml_path), bodies, timestep, sim_duration, ramp_duration, device_ids);
}
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const std::string& hydro_yaml_path, const std::vector<std::shared_ptr<BodyView>>& bodies,
                                                     double timestep, double sim_duration, double ramp_duration, int device_id) {
    return yaml_setup::setup(ReadHydroYAML(hydro_yaml_path), bodies, timestep, sim_duration, ramp_duration, std::vector<int>{device_id});
}

}  // namespace hydroc_amd
