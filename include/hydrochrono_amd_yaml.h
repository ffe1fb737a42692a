/* hydrochrono_amd_yaml.h -- hydro.yaml ingest and YAML -> context wiring (host only until hc_create_from_hydro_yaml).
 *
 * Replaces ReadHydroYAML (src/hydro_yaml_parser.cpp:154-610, result type YAMLHydroData of src/hydro_types.h:19-71) and
 * SetupHydroFromYAML (src/setup_hydro_from_yaml.cpp:126-193) so that "identical YAML/BEMIO inputs" holds end to end.
 * Field names are the reference's member names.
 */
#ifndef HYDROCHRONO_AMD_YAML_H
#define HYDROCHRONO_AMD_YAML_H

#include <stddef.h>

#include "hydrochrono_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hc_yaml hc_yaml;

/* ReadHydroYAML(path).  On failure returns HC_ERR_RUNTIME (the reference throws std::runtime_error) and copies the
 * message into err. */
int hc_yaml_read(const char* hydro_file_path, hc_yaml** out, char* err, size_t errlen);
void hc_yaml_free(hc_yaml* cfg);

/* YAMLHydroData::bodies */
int hc_yaml_num_bodies(const hc_yaml* cfg);
/* field in {"name","h5_file","radiation_calculation","radiation_convolution_mode","td_smoothing"}; NULL if unknown */
const char* hc_yaml_body_string(const hc_yaml* cfg, int body, const char* field);
/* field in {"include_excitation","include_radiation","td_export_plot_csv","td_window_length",
 *           "td_rms_threshold_factor","td_taper_fraction_remaining"}; NaN if unknown */
double hc_yaml_body_number(const hc_yaml* cfg, int body, const char* field);

/* field in {"waves.type","waves.spectrum","radiation_convolution_mode","td_smoothing"}; NULL if unknown */
const char* hc_yaml_string(const hc_yaml* cfg, const char* field);
/* field in {"waves.height","waves.period","waves.direction","waves.phase","waves.seed","td_window_length",
 *           "td_rirf_end_time","td_taper_start_percent","td_taper_end_percent","td_taper_final_amplitude",
 *           "td_export_plot_csv"}; NaN if unknown */
double hc_yaml_number(const hc_yaml* cfg, const char* field);
/* WaveSettings::period_values; returns the count, copies min(count, cap) values */
int hc_yaml_period_values(const hc_yaml* cfg, double* out, int cap);

/* SetupHydroFromYAML(hydro_data, bodies, timestep, sim_duration, ramp_duration): match the YAML bodies against the names
 * of the bodies present in the multibody system (order of the YAML), read the first body's h5 file for the matched
 * bodies, attach the wave model (regular: A = height/2, omega = 2*pi/period; irregular: Pierson-Moskowitz defaults with the
 * YAML seed (<= 0 -> 1); no_wave / still_ci / still: NoWave) and the convolution mode / TaperedDirect options.
 * matched_index[k] receives the index into system_body_names of the k-th hydro body (capacity n_names). */
int hc_create_from_hydro_yaml(const hc_yaml* cfg, const char* const* system_body_names, int n_names, double timestep,
                              double sim_duration, double ramp_duration, int device_id, hc_ctx** out, int* matched_index,
                              int* n_matched, char* err, size_t errlen);

/* The same setup for a system row-sharded over n_shards contexts of this process (device_ids[g]: the device of shard g; contiguous
 * balanced split of the matched bodies, as TestHydro(bodies, h5, waves, device_ids) makes it): out_ctxs[0..n_shards) are then
 * evaluated with hc_step_multi.  On failure no context is left behind. */
int hc_create_from_hydro_yaml_sharded(const hc_yaml* cfg, const char* const* system_body_names, int n_names, double timestep,
                                      double sim_duration, double ramp_duration, const int* device_ids, int n_shards, hc_ctx** out_ctxs,
                                      int* matched_index, int* n_matched, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
