// hc_yaml.cpp -- hydro.yaml ingest (include/hydrochrono_amd_yaml.h).
//
// Accepts the YAML subset the reference's hand-rolled reader accepts (src/hydro_yaml_parser.cpp:154-610): a
// `hydrodynamics:` mapping at column 0 with `bodies:` (list of `- name:` items at column 4, item keys at column 6),
// `waves:` (keys at column 4, optional nested `period:` block) and `convolution:` / `radiation_convolution:` (keys at
// column 4, nested smoothing / taper / diagnostics blocks at column 6), plus the flat system-wide keys at column 2.
// The file is first scanned into (indent, key, value) entries; each section is then read by its own small routine.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hydrochrono_amd_yaml.h"

namespace {

struct Body {  // HydroBody, src/hydro_types.h:19-36
    std::string name, h5_file;
    bool include_excitation = true, include_radiation = true;
    std::string radiation_calculation = "convolution", radiation_convolution_mode = "Baseline", td_smoothing = "sg";
    int td_window_length = 5;
    double td_rms_threshold_factor = 0.02, td_taper_fraction_remaining = 0.25;
    bool td_export_plot_csv = false;
};

}  // namespace

struct hc_yaml {  // YAMLHydroData, src/hydro_types.h:41-71
    std::vector<Body> bodies;
    std::string wave_type = "regular", wave_spectrum = "pierson_moskowitz";
    double wave_height = 0.0, wave_period = 0.0, wave_direction = 0.0, wave_phase = 0.0;
    int wave_seed = -1;
    std::vector<double> period_values;
    std::string radiation_convolution_mode = "Baseline", td_smoothing = "sg";
    int td_window_length = 5;
    double td_rirf_end_time = -1.0, td_taper_start_percent = 0.8, td_taper_end_percent = 1.0, td_taper_final_amplitude = 0.0;
    bool td_export_plot_csv = false;
};

namespace {

std::string trim(std::string s) {
    const char* ws = " \t";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    const size_t b = s.find_last_not_of(ws);
    return s.substr(a, b - a + 1);
}
std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
    return s;
}

struct Entry {
    int indent = 0;
    std::string text;   // trimmed line
    bool keyed = false; // has "key: value" form (after an optional "- ")
    std::string key, value;
};

// "key: value" split with the reference's conventions: first ':' separates, '#' starts a comment inside the value,
// one pair of double quotes around the value is dropped.
bool split_key_value(const std::string& trimmed_line, std::string& key, std::string& value) {
    std::string t = trimmed_line;
    if (!t.empty() && t.back() == '\r') t.pop_back();
    if (t.empty() || t[0] == '#') return false;
    const size_t colon = t.find(':');
    if (colon == std::string::npos) return false;
    key   = trim(t.substr(0, colon));
    value = t.substr(colon + 1);
    const size_t hash = value.find('#');
    if (hash != std::string::npos) value.erase(hash);
    value = trim(value);
    if (!value.empty() && value.back() == '\r') value.pop_back();
    if (value.size() >= 2 && value.front() == '"' && value.back() == '"') value = value.substr(1, value.size() - 2);
    return true;
}

double to_double(const std::string& s, double fallback) {
    try {
        return std::stod(s);
    } catch (const std::exception&) {
        return fallback;
    }
}
bool to_bool(const std::string& s, bool fallback) {
    const std::string l = lower(s);
    if (l == "true" || l == "yes" || l == "1") return true;
    if (l == "false" || l == "no" || l == "0") return false;
    return fallback;
}
void to_int(const std::string& s, int& target) {
    try {
        target = std::stoi(s);
    } catch (...) {
    }
}

std::string resolve_path(const std::string& p, const std::string& yaml_file) {
    namespace fs = std::filesystem;
    const fs::path fp(p);
    if (fp.is_absolute()) return p;
    const fs::path joined = fs::path(yaml_file).parent_path() / fp;
    try {
        return fs::weakly_canonical(joined).string();
    } catch (const std::exception&) {
        return joined.string();
    }
}

std::vector<double> parse_number_list(const std::string& v) {  // "[a, b, c]" -> numbers
    std::vector<double> out;
    const size_t lb = v.find('['), rb = v.find(']');
    if (lb == std::string::npos || rb == std::string::npos || rb <= lb) return out;
    std::string inner = v.substr(lb + 1, rb - lb - 1);
    std::replace(inner.begin(), inner.end(), ',', ' ');
    std::istringstream iss(inner);
    double x;
    while (iss >> x) out.push_back(x);
    return out;
}

struct Reader {
    const std::vector<Entry>& e;
    size_t i = 0;
    hc_yaml& d;
    const std::string& file;
    // wave bookkeeping that is validated after the whole file has been seen
    bool amplitude_set = false;
    double amplitude   = 0.0;
    bool period_seen = false, form_values = false;

    Reader(const std::vector<Entry>& entries, hc_yaml& data, const std::string& f) : e(entries), d(data), file(f) {}

    bool section_header(const Entry& x) const {
        return x.indent == 2 && (x.text == "bodies:" || x.text == "waves:" || x.text == "convolution:" || x.text == "radiation_convolution:");
    }

    void bodies() {
        bool open = false;
        Body cur;
        auto flush = [&] {
            if (open && !cur.name.empty()) d.bodies.push_back(cur);
            open = false;
        };
        for (; i < e.size(); ++i) {
            const Entry& x = e[i];
            if ((x.indent == 0 && x.text == "hydrodynamics:") || section_header(x)) break;
            if (x.indent == 4 && x.text.compare(0, 6, "- name") == 0) {
                flush();
                cur  = Body();
                open = true;
                std::string k, v;
                if (split_key_value(x.text.substr(2), k, v) && k == "name") cur.name = v;
                continue;
            }
            if (!(open && x.indent == 6 && x.keyed)) continue;
            const std::string& k = x.key;
            const std::string& v = x.value;
            if (k == "name") cur.name = v;
            else if (k == "h5_file") cur.h5_file = resolve_path(v, file);
            else if (k == "include_excitation") cur.include_excitation = to_bool(v, true);
            else if (k == "include_radiation") cur.include_radiation = to_bool(v, true);
            else if (k == "radiation_calculation") cur.radiation_calculation = v;
            else if (k == "radiation_convolution_mode") cur.radiation_convolution_mode = v;
            else if (k == "td_smoothing") cur.td_smoothing = v;
            else if (k == "td_window_length") to_int(v, cur.td_window_length);
            else if (k == "td_rms_threshold_factor") cur.td_rms_threshold_factor = to_double(v, cur.td_rms_threshold_factor);
            else if (k == "td_taper_fraction_remaining") cur.td_taper_fraction_remaining = to_double(v, cur.td_taper_fraction_remaining);
            else if (k == "td_export_plot_csv") cur.td_export_plot_csv = to_bool(v, false);
        }
        flush();
    }

    // Note on `period:` forms.  The reference documents nested forms (period:\n  values / linspace / range) but its reader
    // closes the nested block on the very line that opens it (src/hydro_yaml_parser.cpp:533-536 runs for the `period:` line
    // itself), so those files end in "waves.period: invalid or empty specification".  Only the scalar form and the inline
    // `period: { values: [a, b, c] }` form work there, and those are the forms read here; a `period:` key with an empty
    // value leaves the period unset and fails validation with the same message.
    void waves() {
        for (; i < e.size(); ++i) {
            const Entry& x = e[i];
            if ((x.indent == 0 && x.text == "hydrodynamics:") || section_header(x)) break;
            if (!(x.keyed && x.indent == 4)) continue;
            const std::string kl = lower(x.key);
            const std::string& v = x.value;
            if (kl == "type") d.wave_type = v;
            else if (kl == "height" || kl == "h") d.wave_height = to_double(v, 0.0);
            else if (kl == "amplitude" || kl == "a") { amplitude = to_double(v, 0.0); amplitude_set = true; }
            else if (kl == "period" || kl == "t" || kl == "tp" || kl == "p") {
                period_seen = true;
                form_values = false;
                d.period_values.clear();
                const bool structured = v.find('{') != std::string::npos || v.find('[') != std::string::npos || v.empty();
                if (!structured) {
                    d.wave_period = to_double(v, 0.0);
                    d.period_values.push_back(d.wave_period);
                } else if (v.find("values") != std::string::npos && v.find('[') != std::string::npos) {
                    const auto vals = parse_number_list(v);
                    if (!vals.empty()) {
                        d.period_values = vals;
                        d.wave_period   = vals.front();
                        form_values     = true;
                    }
                }
            } else if (kl == "direction") d.wave_direction = to_double(v, 0.0);
            else if (kl == "phase") d.wave_phase = to_double(v, 0.0);
            else if (kl == "spectrum") d.wave_spectrum = v;
            else if (kl == "seed") { d.wave_seed = -1; to_int(v, d.wave_seed); }
        }
    }

    void convolution() {
        enum { kNone, kSmoothing, kTaper, kDiag } sub = kNone;
        for (; i < e.size(); ++i) {
            const Entry& x = e[i];
            if ((x.indent == 0 && x.text == "hydrodynamics:") || section_header(x)) break;
            if (!x.keyed) continue;
            const std::string& k = x.key;
            const std::string& v = x.value;
            if (x.indent == 4) {
                if (k == "mode") d.radiation_convolution_mode = v;
                else if (k == "smoothing") {
                    if (!v.empty()) d.td_smoothing = v;
                    else sub = kSmoothing;
                } else if (k == "taper") sub = kTaper;
                else if (k == "diagnostics") sub = kDiag;
            } else if (x.indent == 6 && sub == kSmoothing) {
                if (k == "type") d.td_smoothing = v;
                else if (k == "window_length") to_int(v, d.td_window_length);
            } else if (x.indent == 6 && sub == kTaper) {
                if (k == "start_percent") d.td_taper_start_percent = to_double(v, d.td_taper_start_percent);
                else if (k == "end_percent") d.td_taper_end_percent = to_double(v, d.td_taper_end_percent);
                else if (k == "final_amplitude") d.td_taper_final_amplitude = to_double(v, d.td_taper_final_amplitude);
                else if (k == "end_time") d.td_rirf_end_time = to_double(v, d.td_rirf_end_time);
            } else if (x.indent == 6 && sub == kDiag) {
                if (k == "export_csv") d.td_export_plot_csv = to_bool(v, false);
            }
        }
    }

    void flat_key(const Entry& x) {  // system-wide keys directly under hydrodynamics:
        if (x.key == "radiation_convolution_mode") d.radiation_convolution_mode = x.value;
        else if (x.key == "td_smoothing") d.td_smoothing = x.value;
        else if (x.key == "td_window_length") to_int(x.value, d.td_window_length);
        else if (x.key == "td_export_plot_csv") d.td_export_plot_csv = to_bool(x.value, false);
    }
};

void read_file(const std::string& path, hc_yaml& d) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("Could not open hydro file: " + path);
    std::vector<Entry> entries;
    std::string line;
    while (std::getline(in, line)) {
        Entry x;
        while (x.indent < static_cast<int>(line.size()) && (line[x.indent] == ' ' || line[x.indent] == '\t')) ++x.indent;
        x.text = trim(line);
        if (x.text.empty() || x.text[0] == '#') continue;
        x.keyed = split_key_value(x.text, x.key, x.value);
        entries.push_back(std::move(x));
    }
    Reader r(entries, d, path);
    bool have_root = false;
    while (r.i < entries.size()) {
        const Entry& x = entries[r.i];
        if (x.indent == 0 && x.text == "hydrodynamics:") {
            have_root = true;
            ++r.i;
        } else if (!have_root) {
            ++r.i;
        } else if (x.indent == 2 && x.text == "bodies:") {
            ++r.i;
            r.bodies();
        } else if (x.indent == 2 && x.text == "waves:") {
            ++r.i;
            r.waves();
        } else if (x.indent == 2 && (x.text == "convolution:" || x.text == "radiation_convolution:")) {
            ++r.i;
            r.convolution();
        } else {
            if (x.indent == 2 && x.keyed) r.flat_key(x);
            ++r.i;
        }
    }
    // ---- validation, in the reference's order (src/hydro_yaml_parser.cpp:552-607) ----
    if (r.period_seen) {
        if (d.period_values.empty()) {
            if (d.wave_period > 0.0) d.period_values.push_back(d.wave_period);
            else throw std::runtime_error("waves.period: invalid or empty specification");
        }
    } else if (d.period_values.empty() && d.wave_period > 0.0) {
        d.period_values.push_back(d.wave_period);
    }
    if (r.amplitude_set) {
        const double derived = 2.0 * r.amplitude;
        if (d.wave_height > 0.0) {
            if (std::abs(d.wave_height - derived) > 1e-9)
                throw std::runtime_error("waves: both height and amplitude provided but inconsistent (expected height = 2*amplitude)");
        } else {
            d.wave_height = derived;
        }
    }
    if (lower(d.wave_type) == "regular") {
        if (d.wave_height <= 0.0) throw std::runtime_error("waves: regular requires wave height (use 'height' or 'h', or 'amplitude'/'a')");
        if (!(d.wave_period > 0.0 || !d.period_values.empty()))
            throw std::runtime_error("waves: regular requires wave period (use 'period' or shorthand 't', 'tp', or 'p')");
    }
    if (!have_root) throw std::runtime_error("No 'hydrodynamics:' section found in hydro file: " + path);
}

void copy_err(const std::string& m, char* err, size_t n) {
    if (err && n) std::snprintf(err, n, "%s", m.c_str());
}

}  // namespace

extern "C" {

int hc_yaml_read(const char* path, hc_yaml** out, char* err, size_t errlen) {
    if (!path || !out) return HC_ERR_INVALID;
    *out = nullptr;
    auto* d = new hc_yaml;
    try {
        read_file(path, *d);
    } catch (const std::exception& e) {
        copy_err(e.what(), err, errlen);
        delete d;
        return HC_ERR_RUNTIME;
    }
    *out = d;
    return HC_OK;
}

void hc_yaml_free(hc_yaml* cfg) { delete cfg; }

int hc_yaml_num_bodies(const hc_yaml* c) { return c ? static_cast<int>(c->bodies.size()) : 0; }

const char* hc_yaml_body_string(const hc_yaml* c, int body, const char* field) {
    if (!c || !field || body < 0 || body >= static_cast<int>(c->bodies.size())) return nullptr;
    const Body& b = c->bodies[body];
    const std::string f(field);
    if (f == "name") return b.name.c_str();
    if (f == "h5_file") return b.h5_file.c_str();
    if (f == "radiation_calculation") return b.radiation_calculation.c_str();
    if (f == "radiation_convolution_mode") return b.radiation_convolution_mode.c_str();
    if (f == "td_smoothing") return b.td_smoothing.c_str();
    return nullptr;
}

double hc_yaml_body_number(const hc_yaml* c, int body, const char* field) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (!c || !field || body < 0 || body >= static_cast<int>(c->bodies.size())) return nan;
    const Body& b = c->bodies[body];
    const std::string f(field);
    if (f == "include_excitation") return b.include_excitation;
    if (f == "include_radiation") return b.include_radiation;
    if (f == "td_export_plot_csv") return b.td_export_plot_csv;
    if (f == "td_window_length") return b.td_window_length;
    if (f == "td_rms_threshold_factor") return b.td_rms_threshold_factor;
    if (f == "td_taper_fraction_remaining") return b.td_taper_fraction_remaining;
    return nan;
}

const char* hc_yaml_string(const hc_yaml* c, const char* field) {
    if (!c || !field) return nullptr;
    const std::string f(field);
    if (f == "waves.type") return c->wave_type.c_str();
    if (f == "waves.spectrum") return c->wave_spectrum.c_str();
    if (f == "radiation_convolution_mode") return c->radiation_convolution_mode.c_str();
    if (f == "td_smoothing") return c->td_smoothing.c_str();
    return nullptr;
}

double hc_yaml_number(const hc_yaml* c, const char* field) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (!c || !field) return nan;
    const std::string f(field);
    if (f == "waves.height") return c->wave_height;
    if (f == "waves.period") return c->wave_period;
    if (f == "waves.direction") return c->wave_direction;
    if (f == "waves.phase") return c->wave_phase;
    if (f == "waves.seed") return c->wave_seed;
    if (f == "td_window_length") return c->td_window_length;
    if (f == "td_rirf_end_time") return c->td_rirf_end_time;
    if (f == "td_taper_start_percent") return c->td_taper_start_percent;
    if (f == "td_taper_end_percent") return c->td_taper_end_percent;
    if (f == "td_taper_final_amplitude") return c->td_taper_final_amplitude;
    if (f == "td_export_plot_csv") return c->td_export_plot_csv;
    return nan;
}

int hc_yaml_period_values(const hc_yaml* c, double* out, int cap) {
    if (!c) return 0;
    const int n = static_cast<int>(c->period_values.size());
    if (out)
        for (int i = 0; i < std::min(n, cap); ++i) out[i] = c->period_values[i];
    return n;
}

// one context of the system: the rows of bodies [b0, b1) of the N matched bodies on `device_id`
static int create_one_from_yaml(const hc_yaml* cfg, int N, int b0, int b1, double timestep, double sim_duration, double ramp_duration,
                                int device_id, hc_ctx** out, char* err, size_t errlen);

int hc_create_from_hydro_yaml_sharded(const hc_yaml* cfg, const char* const* names, int n_names, double timestep, double sim_duration,
                                      double ramp_duration, const int* device_ids, int n_shards, hc_ctx** out_ctxs, int* matched_index,
                                      int* n_matched, char* err, size_t errlen) {
    if (!cfg || !out_ctxs || (n_names > 0 && !names) || n_shards <= 0 || !device_ids) return HC_ERR_INVALID;
    for (int g = 0; g < n_shards; ++g) out_ctxs[g] = nullptr;
    // MatchBodiesByName (src/setup_hydro_from_yaml.cpp:84-122): YAML order, first h5 file for everybody
    std::vector<int> match;
    for (const Body& hb : cfg->bodies)
        for (int k = 0; k < n_names; ++k)
            if (names[k] && hb.name == names[k]) {
                match.push_back(k);
                break;
            }
    if (n_matched) *n_matched = static_cast<int>(match.size());
    if (matched_index)
        for (size_t k = 0; k < match.size(); ++k) matched_index[k] = match[k];
    if (match.empty()) {
        copy_err("No hydrodynamic bodies found in Chrono system", err, errlen);
        return HC_ERR_RUNTIME;
    }
    const int N = static_cast<int>(match.size());
    if (n_shards > N) {
        copy_err("more shards than hydrodynamic bodies", err, errlen);
        return HC_ERR_INVALID;
    }
    const int base = N / n_shards, extra = N % n_shards;  // contiguous balanced split, like TestHydro(bodies, h5, waves, device_ids)
    for (int g = 0; g < n_shards; ++g) {
        const int b0 = g * base + std::min(g, extra), b1 = b0 + base + (g < extra ? 1 : 0);
        const int rc = create_one_from_yaml(cfg, N, b0, b1, timestep, sim_duration, ramp_duration, device_ids[g], &out_ctxs[g], err, errlen);
        if (rc != HC_OK) {
            for (int k = 0; k < g; ++k) {
                hc_destroy(out_ctxs[k]);
                out_ctxs[k] = nullptr;
            }
            return rc;
        }
    }
    return HC_OK;
}

int hc_create_from_hydro_yaml(const hc_yaml* cfg, const char* const* names, int n_names, double timestep, double sim_duration,
                              double ramp_duration, int device_id, hc_ctx** out, int* matched_index, int* n_matched, char* err,
                              size_t errlen) {
    if (!out) return HC_ERR_INVALID;
    return hc_create_from_hydro_yaml_sharded(cfg, names, n_names, timestep, sim_duration, ramp_duration, &device_id, 1, out, matched_index,
                                             n_matched, err, errlen);
}

static int create_one_from_yaml(const hc_yaml* cfg, int N, int b0, int b1, double timestep, double sim_duration, double ramp_duration,
                                int device_id, hc_ctx** out, char* err, size_t errlen) {
    *out = nullptr;
    const std::string h5 = cfg->bodies.front().h5_file;
    hc_ctx* ctx = nullptr;
    int rc = hc_create_sharded(N, b0, b1, device_id, &ctx);
    if (rc != HC_OK) {
        copy_err(hc_last_error(nullptr), err, errlen);
        return rc;
    }
    auto fail = [&](int code) {
        copy_err(hc_last_error(ctx), err, errlen);
        hc_destroy(ctx);
        return code;
    };
    if ((rc = hc_load_bemio_h5(ctx, h5.c_str())) != HC_OK) return fail(rc);
    if ((rc = hc_finalize(ctx)) != HC_OK) return fail(rc);
    // CreateWaveFromSettings (:28-79)
    const std::string type = lower(cfg->wave_type);
    if (type == "regular") {
        rc = hc_set_wave_regular(ctx, N, cfg->wave_height / 2.0, 2.0 * M_PI / cfg->wave_period);
    } else if (type == "irregular") {
        hc_irregular_wave_params p;
        hc_irregular_wave_params_default(&p);
        p.num_bodies          = N;
        p.simulation_dt       = timestep;
        p.simulation_duration = sim_duration;
        p.ramp_duration       = ramp_duration;
        p.wave_height         = cfg->wave_height;
        p.wave_period         = cfg->wave_period;
        p.seed                = cfg->wave_seed > 0 ? cfg->wave_seed : 1;
        rc = hc_set_wave_irregular(ctx, &p);
    } else if (type == "no_wave" || type == "still_ci" || type == "still") {
        rc = hc_set_wave_none(ctx, N);
    } else {
        copy_err("Unsupported wave type: " + cfg->wave_type, err, errlen);
        hc_destroy(ctx);
        return HC_ERR_RUNTIME;
    }
    if (rc != HC_OK) return fail(rc);
    // convolution mode (:151-190)
    if (lower(cfg->radiation_convolution_mode) == "tapereddirect") {
        hc_tapered_direct_options o;
        hc_tapered_direct_options_default(&o);
        const std::string sm = cfg->td_smoothing.empty() ? std::string("sg") : cfg->td_smoothing;
        o.smoothing     = (sm == "moving_average") ? 1 : 0;
        o.window_length = std::max(3, cfg->td_window_length != 0 ? cfg->td_window_length : 5);
        if (o.window_length % 2 == 0) o.window_length += 1;
        o.rirf_end_time         = cfg->td_rirf_end_time;
        o.taper_start_percent   = cfg->td_taper_start_percent;
        o.taper_end_percent     = cfg->td_taper_end_percent;
        o.taper_final_amplitude = cfg->td_taper_final_amplitude;
        o.export_plot_csv       = cfg->td_export_plot_csv ? 1 : 0;
        if ((rc = hc_set_convolution_mode(ctx, 1)) != HC_OK) return fail(rc);
        if ((rc = hc_set_tapered_direct_options(ctx, &o)) != HC_OK) return fail(rc);
    } else {
        if ((rc = hc_set_convolution_mode(ctx, 0)) != HC_OK) return fail(rc);
    }
    *out = ctx;
    return HC_OK;
}

}  // extern "C"
