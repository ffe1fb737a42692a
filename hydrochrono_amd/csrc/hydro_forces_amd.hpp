// hydro_forces_amd.hpp -- C++ host-side mirror of the reference's plugin surface over the C ABI
// (include/hydrochrono_amd.h).  Header-only; link with libhydrochrono_amd.so.
//
// Same names, argument meaning and error behaviour as the reference for the hot path:
//   WaveBase / NoWave / RegularWave / IrregularWaves(IrregularWaveParams)   include/hydroc/wave_types.h:52-467
//   TestHydro(bodies, h5_file, waves), AddWaves, ComputeForce*, CoordinateFuncForBody, SetRadiationConvolutionMode,
//   SetTaperedDirectOptions, GetProfileStats                               include/hydroc/hydro_forces.h:164-285
//   ChLoadAddedMass::{ComputeJacobian, LoadIntLoadResidual_Mv}             include/hydroc/chloadaddedmass.h:22-90
// C status codes are rethrown as the exception types the reference throws (std::runtime_error / std::out_of_range).
//
// Bodies are seen through the small `HydroBody` interface (name, time, pose, velocities) -- exactly the ChBody getters
// the reference calls (src/hydro_forces.cpp:106-107,279-280,550,567-568).  `MockBody` implements it with plain fields
// for drivers without Chrono; with Project Chrono on the include path (HYDROCHRONO_AMD_WITH_CHRONO or auto-detected)
// `ChronoBody` wraps a chrono::ChBody and the ChFunction / ChForce / ChLoadCustomMultiple adapters at the bottom of
// this file wire everything into a ChSystem the way the reference's ForceFunc6d / ChLoadAddedMass do.
#pragma once

#include <algorithm>
#include <array>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hydrochrono_amd.h"
#include "../../include/hydrochrono_amd_yaml.h"

#if !defined(HYDROCHRONO_AMD_WITH_CHRONO) && defined(__has_include)
#if __has_include(<chrono/physics/ChBody.h>)
#define HYDROCHRONO_AMD_WITH_CHRONO 1
#endif
#endif

namespace hydroc_amd {

inline void check(hc_ctx* ctx, int rc) {
    if (rc == HC_OK) return;
    const std::string msg = hc_last_error(ctx);
    if (rc == HC_ERR_OUT_OF_RANGE) throw std::out_of_range(msg);
    throw std::runtime_error(msg);
}

// ---------------------------------------------------------------------------------------------------------------
// Body view
// ---------------------------------------------------------------------------------------------------------------
struct HydroBody {
    virtual ~HydroBody()                                   = default;
    virtual std::string GetName() const                    = 0;  // "body<k>", 1-based (src/hydro_forces.cpp:106-107)
    virtual double GetChTime() const                       = 0;
    virtual std::array<double, 3> GetPos() const           = 0;
    virtual std::array<double, 3> GetCardanAnglesXYZ() const = 0;  // GetRot().GetCardanAnglesXYZ()
    virtual std::array<double, 3> GetPosDt() const         = 0;
    virtual std::array<double, 3> GetAngVelParent() const  = 0;
};

struct MockBody : HydroBody {
    std::string name;
    double time = 0.0;
    std::array<double, 3> pos{0, 0, 0}, rpy{0, 0, 0}, linvel{0, 0, 0}, angvel{0, 0, 0};
    explicit MockBody(std::string n) : name(std::move(n)) {}
    std::string GetName() const override { return name; }
    double GetChTime() const override { return time; }
    std::array<double, 3> GetPos() const override { return pos; }
    std::array<double, 3> GetCardanAnglesXYZ() const override { return rpy; }
    std::array<double, 3> GetPosDt() const override { return linvel; }
    std::array<double, 3> GetAngVelParent() const override { return angvel; }
};

// ---------------------------------------------------------------------------------------------------------------
// Wave models (configuration holders; the arithmetic lives behind hc_set_wave_*)
// ---------------------------------------------------------------------------------------------------------------
enum class WaveMode { noWaveCIC = 0, regular = 1, irregular = 2 };

class WaveBase {
  public:
    virtual ~WaveBase()             = default;
    virtual WaveMode GetWaveMode()  = 0;
    virtual void Attach(hc_ctx* ctx) = 0;  // AddH5Data + Initialize of the reference, executed by the library
};

class NoWave : public WaveBase {
  public:
    NoWave() : num_bodies_(1) {}
    explicit NoWave(unsigned num_b) : num_bodies_(num_b) {}
    WaveMode GetWaveMode() override { return WaveMode::noWaveCIC; }
    void Attach(hc_ctx* ctx) override { check(ctx, hc_set_wave_none(ctx, static_cast<int>(num_bodies_))); }

  private:
    unsigned num_bodies_;
};

class RegularWave : public WaveBase {
  public:
    RegularWave() : num_bodies_(1) {}
    explicit RegularWave(unsigned num_b) : num_bodies_(num_b) {}
    WaveMode GetWaveMode() override { return WaveMode::regular; }
    void Attach(hc_ctx* ctx) override {
        check(ctx, hc_set_wave_regular(ctx, static_cast<int>(num_bodies_), regular_wave_amplitude_, regular_wave_omega_));
    }
    double regular_wave_amplitude_ = 0.0;
    double regular_wave_omega_     = 0.0;
    double regular_wave_phase_     = 0.0;  // unused by the force, as in the reference

  private:
    unsigned num_bodies_;
};

struct IrregularWaveParams {  // include/hydroc/wave_types.h:277-292
    unsigned int num_bodies_        = 1;
    double simulation_dt_           = 0.0;
    double simulation_duration_     = 0.0;
    double ramp_duration_           = 0.0;
    std::string eta_file_path_;     // not supported (undefined behaviour in the reference)
    double wave_height_             = 0.0;
    double wave_period_             = 0.0;
    double frequency_min_           = 0.001;
    double frequency_max_           = 1.0;
    double nfrequencies_            = 0;
    double peak_enhancement_factor_ = 1.0;
    bool is_normalized_             = false;
    int seed_                       = 1;
    bool wave_stretching_           = true;
};

class IrregularWaves : public WaveBase {
  public:
    explicit IrregularWaves(const IrregularWaveParams& params) : params_(params) {}
    WaveMode GetWaveMode() override { return WaveMode::irregular; }
    void Attach(hc_ctx* ctx) override {
        if (!params_.eta_file_path_.empty()) throw std::runtime_error("eta_file_path_ is not supported by the GPU path");
        hc_irregular_wave_params p;
        hc_irregular_wave_params_default(&p);
        p.num_bodies              = static_cast<int>(params_.num_bodies_);
        p.simulation_dt           = params_.simulation_dt_;
        p.simulation_duration     = params_.simulation_duration_;
        p.ramp_duration           = params_.ramp_duration_;
        p.wave_height             = params_.wave_height_;
        p.wave_period             = params_.wave_period_;
        p.frequency_min           = params_.frequency_min_;
        p.frequency_max           = params_.frequency_max_;
        p.nfrequencies            = params_.nfrequencies_;
        p.peak_enhancement_factor = params_.peak_enhancement_factor_;
        p.is_normalized           = params_.is_normalized_ ? 1 : 0;
        p.seed                    = params_.seed_;
        check(ctx, hc_set_wave_irregular(ctx, &p));
        ctx_ = ctx;
    }
    // exporter inputs (src/wave_types.cpp:461-478)
    std::vector<double> GetFreeSurfaceTime() const { return table(true); }
    std::vector<double> GetFreeSurfaceElevation() const { return table(false); }
    std::vector<double> GetFrequenciesHz() const {
        int nf = 0;
        check(ctx_, hc_get_sizes(ctx_, nullptr, nullptr, nullptr, nullptr, &nf, nullptr, nullptr, nullptr));
        std::vector<double> f(nf);
        check(ctx_, hc_get_spectrum(ctx_, f.data(), nullptr, nullptr, nullptr, nullptr));
        return f;
    }

  private:
    std::vector<double> table(bool time) const {
        int nt = 0;
        check(ctx_, hc_get_sizes(ctx_, nullptr, nullptr, nullptr, nullptr, nullptr, &nt, nullptr, nullptr));
        std::vector<double> v(nt);
        check(ctx_, time ? hc_get_eta_table(ctx_, v.data(), nullptr) : hc_get_eta_table(ctx_, nullptr, v.data()));
        return v;
    }
    IrregularWaveParams params_;
    hc_ctx* ctx_ = nullptr;
};

struct HydroProfileStats {  // include/hydroc/hydro_forces.h:153-160
    double hydrostatics_seconds = 0.0, radiation_seconds = 0.0, waves_seconds = 0.0;
    int hydrostatics_calls = 0, radiation_calls = 0, waves_calls = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// TestHydro
// ---------------------------------------------------------------------------------------------------------------
class TestHydro {
  public:
    TestHydro()                                = delete;
    TestHydro(const TestHydro&)                = delete;
    TestHydro& operator=(const TestHydro&)     = delete;

    TestHydro(std::vector<std::shared_ptr<HydroBody>> user_bodies, const std::string& h5_file_name,
              std::shared_ptr<WaveBase> waves = std::make_shared<NoWave>(), int device_id = 0)
        : TestHydro(std::move(user_bodies), h5_file_name, std::move(waves), std::vector<int>{device_id}) {}
    // Multi-GPU inside the one Chrono process (drop-in variant): one body-row shard per entry of `device_ids`
    // (a device may be named more than once), contiguous balanced split of the bodies; every call below fans out to the
    // shard contexts, the per-step evaluation goes through hc_step_multi (all GPUs started before any is waited for,
    // host-side gather).  One entry = the single-GPU object.
    TestHydro(std::vector<std::shared_ptr<HydroBody>> user_bodies, const std::string& h5_file_name, std::shared_ptr<WaveBase> waves,
              const std::vector<int>& device_ids)
        : bodies_(std::move(user_bodies)), num_bodies_(static_cast<int>(bodies_.size())) {
        if (bodies_.empty()) throw std::runtime_error("TestHydro needs at least one body");
        if (device_ids.empty() || static_cast<int>(device_ids.size()) > num_bodies_)
            throw std::runtime_error("TestHydro: between one shard and one shard per body");
        // body numbers come from the names "body<k>", 1-based (ForceFunc6d ctor, src/hydro_forces.cpp:104-108)
        for (auto& b : bodies_) {
            std::string temp = b->GetName();
            body_numbers_.push_back(std::stoi(temp.erase(0, 4)));
        }
        const int G = static_cast<int>(device_ids.size()), base = num_bodies_ / G, extra = num_bodies_ % G;
        try {
            for (int g = 0; g < G; ++g) {
                const int b0 = g * base + (g < extra ? g : extra), b1 = b0 + base + (g < extra ? 1 : 0);
                hc_ctx* c = nullptr;
                if (hc_create_sharded(num_bodies_, b0, b1, device_ids[g], &c) != HC_OK) throw std::runtime_error(hc_last_error(nullptr));
                ctxs_.push_back(c);
                check(c, hc_load_bemio_h5(c, h5_file_name.c_str()));
                check(c, hc_finalize(c));
            }
            ctx_ = ctxs_[0];
            chrono_loop_defaults();
            AddWaves(std::move(waves));
        } catch (...) {
            for (hc_ctx* c : ctxs_) hc_destroy(c);
            throw;
        }
        total_force_.assign(6 * num_bodies_, 0.0);
    }
    // Adopts a context that is already configured (used by SetupHydroFromYAML below).
    TestHydro(std::vector<std::shared_ptr<HydroBody>> user_bodies, hc_ctx* configured_ctx)
        : TestHydro(std::move(user_bodies), std::vector<hc_ctx*>{configured_ctx}) {}
    // ... or the configured shard contexts of one system (together they own bodies [0, N)).
    TestHydro(std::vector<std::shared_ptr<HydroBody>> user_bodies, std::vector<hc_ctx*> configured_ctxs)
        : bodies_(std::move(user_bodies)), num_bodies_(static_cast<int>(bodies_.size())), ctxs_(std::move(configured_ctxs)) {
        if (ctxs_.empty()) throw std::runtime_error("TestHydro: no context");
        ctx_ = ctxs_[0];
        chrono_loop_defaults();
        for (auto& b : bodies_) {
            std::string temp = b->GetName();
            body_numbers_.push_back(std::stoi(temp.erase(0, 4)));
        }
        total_force_.assign(6 * num_bodies_, 0.0);
    }
    ~TestHydro() {
        for (hc_ctx* c : ctxs_) hc_destroy(c);
    }

    void AddWaves(std::shared_ptr<WaveBase> waves) {  // src/hydro_forces.cpp:244-261
        user_waves_ = std::move(waves);
        for (auto it = ctxs_.rbegin(); it != ctxs_.rend(); ++it) user_waves_->Attach(*it);  // (the wave object keeps the first context for its getters)
    }
    std::shared_ptr<WaveBase> GetWave() const { return user_waves_; }
    void SetGravitationalAcceleration(double gx, double gy, double gz) {  // ChSystem setting read at :268
        const double g[3] = {gx, gy, gz};
        for (hc_ctx* c : ctxs_) check(c, hc_set_gravity(c, g));
    }

    enum class RadiationConvolutionMode { Baseline, TaperedDirect };
    void SetRadiationConvolutionMode(RadiationConvolutionMode mode) {
        for (hc_ctx* c : ctxs_) check(c, hc_set_convolution_mode(c, mode == RadiationConvolutionMode::TaperedDirect ? 1 : 0));
    }
    struct TaperedDirectOptions {
        std::string smoothing        = "sg";
        int window_length            = 5;
        double rirf_end_time         = -1.0;
        double taper_start_percent   = 0.8;
        double taper_end_percent     = 1.0;
        double taper_final_amplitude = 0.0;
        bool export_plot_csv         = false;  // rirf_processing_body<b>_dof<d>.csv in the diagnostics directory (src/hydro_forces.cpp:509-531)
    };
    void SetTaperedDirectOptions(const TaperedDirectOptions& o) {
        hc_tapered_direct_options c;
        hc_tapered_direct_options_default(&c);
        c.smoothing             = (o.smoothing == "moving_average") ? 1 : 0;
        c.window_length         = o.window_length;
        c.rirf_end_time         = o.rirf_end_time;
        c.taper_start_percent   = o.taper_start_percent;
        c.taper_end_percent     = o.taper_end_percent;
        c.taper_final_amplitude = o.taper_final_amplitude;
        c.export_plot_csv       = o.export_plot_csv ? 1 : 0;
        for (hc_ctx* x : ctxs_) check(x, hc_set_tapered_direct_options(x, &c));
    }
    // include/hydroc/hydro_forces.h:269
    void SetDiagnosticsOutputDirectory(const std::string& dir) {
        for (hc_ctx* x : ctxs_) check(x, hc_set_diagnostics_output_directory(x, dir.c_str()));
    }

    // Not in the reference (it has no such notion): when the look-ahead pass of a block runs, see hc_set_pass_schedule.  This class
    // is driven by a Chrono loop, which does its own work between two force evaluations, so it selects "one block ahead" for systems
    // with 256 MB of K and more when it is constructed (the C ABI's own default does so for wide systems only): with 30 / 100 us of
    // host work between calls a 64-body step takes 12.8 / 12.7 us instead of 17.4 / 15.6, and no step waits for a whole pass.
    void SetPassSchedule(bool one_block_ahead, int slices = 0) {
        for (hc_ctx* x : ctxs_) check(x, hc_set_pass_schedule(x, one_block_ahead ? 1 : 0, slices));
    }
    // (HC_PASS_AHEAD in the environment keeps its say.  Systems whose whole K is below 256 MB -- about 30 bodies; the reference's own
    // one- to three-body demos are 0.3 to 2.6 MB -- keep the library's default: their pass takes at most a few tens of microseconds,
    // which any host work between the calls hides already, and the extra launches of the schedule would cost a back-to-back caller
    // about a microsecond per step.)
    void chrono_loop_defaults() {
        if (std::getenv("HC_PASS_AHEAD") || ctxs_.empty()) return;
        int N = 0, S = 0;
        check(ctxs_[0], hc_get_sizes(ctxs_[0], &N, nullptr, &S, nullptr, nullptr, nullptr, nullptr, nullptr));
        if (8.0 * (6.0 * N) * (6.0 * N) * S < 256e6) return;
        for (hc_ctx* x : ctxs_) check(x, hc_set_pass_schedule(x, 1, 0));
    }

    std::vector<double> ComputeForceHydrostatics() {
        gather_state();
        std::vector<double> out(6 * num_bodies_);
        for (hc_ctx* c : ctxs_) check(c, hc_compute_hydrostatics(c, pos_.data(), rpy_.data(), out.data() + row0(c)));
        return out;
    }
    std::vector<double> ComputeForceRadiationDampingConv() {
        gather_state();
        std::vector<double> out(6 * num_bodies_);
        for (hc_ctx* c : ctxs_) check(c, hc_compute_radiation(c, bodies_[0]->GetChTime(), lin_.data(), ang_.data(), out.data() + row0(c)));
        return out;
    }
    std::vector<double> ComputeForceWaves() {
        std::vector<double> out(6 * num_bodies_);
        for (hc_ctx* c : ctxs_) check(c, hc_compute_waves(c, bodies_[0]->GetChTime(), out.data() + row0(c)));
        return out;
    }

    // src/hydro_forces.cpp:727-767.  b is 1-based.  All 6N callbacks of one Chrono update share one evaluation.
    double CoordinateFuncForBody(int b, int dof_index) {
        if (dof_index < 0 || dof_index >= 6 || b < 1 || b > num_bodies_) throw std::out_of_range("Invalid index in CoordinateFuncForBody");
        const double t = bodies_[0]->GetChTime();
        if (!(have_time_ && t == prev_time_)) {
            prev_time_ = t;
            have_time_ = true;
            gather_state();
            if (ctxs_.size() == 1) check(ctx_, hc_step(ctx_, t, pos_.data(), rpy_.data(), lin_.data(), ang_.data(), total_force_.data()));
            else check(ctx_, hc_step_multi(ctxs_.data(), static_cast<int>(ctxs_.size()), t, pos_.data(), rpy_.data(), lin_.data(), ang_.data(), total_force_.data()));
        }
        return total_force_[6 * (b - 1) + dof_index];
    }

    // GPU seconds per term; the shards of a multi-GPU object run side by side, so the largest shard figure is reported
    HydroProfileStats GetProfileStats() const {
        HydroProfileStats s;
        for (hc_ctx* c : ctxs_) {
            hc_profile_stats p;
            check(c, hc_get_profile(c, &p));
            s.hydrostatics_seconds = std::max(s.hydrostatics_seconds, p.hydrostatics_seconds);
            s.radiation_seconds    = std::max(s.radiation_seconds, p.radiation_seconds);
            s.waves_seconds        = std::max(s.waves_seconds, p.waves_seconds);
            s.hydrostatics_calls   = p.hydrostatics_calls;
            s.radiation_calls      = p.radiation_calls;
            s.waves_calls          = p.waves_calls;
        }
        return s;
    }

    // ChLoadAddedMass data (src/chloadaddedmass.cpp)
    std::vector<double> GetAddedMassMatrix() const {
        const size_t D = static_cast<size_t>(6) * num_bodies_;
        std::vector<double> M(D * D);
        for (hc_ctx* c : ctxs_) check(c, hc_added_mass_matrix(c, M.data() + static_cast<size_t>(row0(c)) * D));  // each shard: its rows
        return M;
    }
    void AddedMassMv(double* R, const double* w, double c, int n_sys) const {
        if (ctxs_.size() == 1) check(ctx_, hc_added_mass_mv(ctx_, w, c, R, n_sys));
        else check(ctx_, hc_added_mass_mv_multi(ctxs_.data(), static_cast<int>(ctxs_.size()), w, c, R, n_sys));
    }

    hc_ctx* context() const { return ctx_; }
    const std::vector<hc_ctx*>& contexts() const { return ctxs_; }
    int num_shards() const { return static_cast<int>(ctxs_.size()); }
    int body_number(int i) const { return body_numbers_[i]; }
    int num_bodies() const { return num_bodies_; }

  private:
    void gather_state() {
        const size_t n = static_cast<size_t>(3) * num_bodies_;
        pos_.resize(n); rpy_.resize(n); lin_.resize(n); ang_.resize(n);
        for (int b = 0; b < num_bodies_; ++b) {
            const auto p = bodies_[b]->GetPos(), r = bodies_[b]->GetCardanAnglesXYZ(), v = bodies_[b]->GetPosDt(),
                       w = bodies_[b]->GetAngVelParent();
            for (int k = 0; k < 3; ++k) {
                pos_[3 * b + k] = p[k]; rpy_[3 * b + k] = r[k]; lin_[3 * b + k] = v[k]; ang_[3 * b + k] = w[k];
            }
        }
    }
    static int row0(hc_ctx* c) {  // first output row of a shard context
        int b0 = 0;
        check(c, hc_get_shard(c, &b0, nullptr));
        return 6 * b0;
    }
    std::vector<std::shared_ptr<HydroBody>> bodies_;
    int num_bodies_;
    std::vector<int> body_numbers_;
    std::vector<hc_ctx*> ctxs_;  // one per body-row shard (one = the single-GPU object)
    hc_ctx* ctx_ = nullptr;      // ctxs_[0]
    std::shared_ptr<WaveBase> user_waves_;
    std::vector<double> total_force_, pos_, rpy_, lin_, ang_;
    bool have_time_   = false;
    double prev_time_ = -1.0;
};

// SetupHydroFromYAML(hydro_data, bodies, timestep, sim_duration, ramp_duration) of the reference
// (src/setup_hydro_from_yaml.cpp:126-193) with ReadHydroYAML folded in: bodies are matched to the YAML entries by name
// (YAML order), the first body's h5 file is read, waves and convolution options come from the YAML.
inline std::unique_ptr<TestHydro> SetupHydroFromYAML(const std::string& hydro_yaml_path,
                                                     const std::vector<std::shared_ptr<HydroBody>>& bodies, double timestep,
                                                     double sim_duration, double ramp_duration, const std::vector<int>& device_ids = {0}) {
    char err[1024] = {0};
    hc_yaml* cfg   = nullptr;
    if (hc_yaml_read(hydro_yaml_path.c_str(), &cfg, err, sizeof err) != HC_OK) throw std::runtime_error(err);
    std::vector<std::string> names;
    for (auto& b : bodies) names.push_back(b->GetName());
    std::vector<const char*> cnames;
    for (auto& n : names) cnames.push_back(n.c_str());
    std::vector<int> matched(bodies.size() + 1);
    int n_matched = 0;
    std::vector<hc_ctx*> ctxs(device_ids.size(), nullptr);  // one row shard per listed device (multi-GPU inside this process)
    const int rc  = hc_create_from_hydro_yaml_sharded(cfg, cnames.data(), static_cast<int>(cnames.size()), timestep, sim_duration, ramp_duration,
                                                      device_ids.data(), static_cast<int>(device_ids.size()), ctxs.data(), matched.data(),
                                                      &n_matched, err, sizeof err);
    hc_yaml_free(cfg);
    if (rc != HC_OK) throw std::runtime_error(err);
    std::vector<std::shared_ptr<HydroBody>> hydro_bodies;
    for (int k = 0; k < n_matched; ++k) hydro_bodies.push_back(bodies[matched[k]]);
    return std::make_unique<TestHydro>(std::move(hydro_bodies), std::move(ctxs));
}

}  // namespace hydroc_amd

// =================================================================================================================
// Project Chrono adapters (compiled only when Chrono headers are available; in the build container they are compiled and driven
// against stand-in Chrono headers, tests/cpp/chrono_stub + tests/test_chrono_adapter.py -- test infrastructure, not a Chrono build)
// =================================================================================================================
#ifdef HYDROCHRONO_AMD_WITH_CHRONO
#include <chrono/functions/ChFunction.h>
#include <chrono/physics/ChBody.h>
#include <chrono/physics/ChForce.h>
#include <chrono/physics/ChLoad.h>
#include <chrono/physics/ChLoadContainer.h>
#include <chrono/physics/ChSystem.h>

namespace hydroc_amd {

struct ChronoBody : HydroBody {
    std::shared_ptr<chrono::ChBody> body;
    explicit ChronoBody(std::shared_ptr<chrono::ChBody> b) : body(std::move(b)) {}
    std::string GetName() const override { return body->GetName(); }
    double GetChTime() const override { return body->GetChTime(); }
    std::array<double, 3> GetPos() const override { auto v = body->GetPos(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetCardanAnglesXYZ() const override { auto v = body->GetRot().GetCardanAnglesXYZ(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetPosDt() const override { auto v = body->GetPosDt(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetAngVelParent() const override { auto v = body->GetAngVelParent(); return {v.x(), v.y(), v.z()}; }
};

// ComponentFunc (include/hydroc/hydro_forces.h:45-86): GetVal's argument is ignored, time comes from body 0.
class ComponentFunc : public chrono::ChFunction {
  public:
    ComponentFunc(TestHydro* hydro, int body_1based, int dof) : hydro_(hydro), b_(body_1based), i_(dof) {}
    ComponentFunc* Clone() const override { return new ComponentFunc(*this); }
    double GetVal(double) const override { return hydro_->CoordinateFuncForBody(b_, i_); }

  private:
    TestHydro* hydro_;
    int b_, i_;
};

// ChLoadAddedMass (include/hydroc/chloadaddedmass.h:22-90)
class ChLoadAddedMass : public chrono::ChLoadCustomMultiple {
  public:
    ChLoadAddedMass(TestHydro* hydro, std::vector<std::shared_ptr<chrono::ChLoadable>>& bodies, chrono::ChSystem* system)
        : chrono::ChLoadCustomMultiple(bodies), hydro_(hydro), system_(system) {
        const int D = 6 * hydro_->num_bodies();
        const auto M = hydro_->GetAddedMassMatrix();
        infinite_added_mass_.setZero(D, D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) infinite_added_mass_(i, j) = M[static_cast<size_t>(i) * D + j];
        infinite_added_mass_system_ = infinite_added_mass_;
    }
    ChLoadAddedMass* Clone() const override { return new ChLoadAddedMass(*this); }
    void ComputeQ(chrono::ChState*, chrono::ChStateDelta*) override {}
    void ComputeJacobian(chrono::ChState*, chrono::ChStateDelta*) override {  // src/chloadaddedmass.cpp:27-53
        auto mmrows = system_->GetNumCoordsVelLevel();
        if (mmrows != infinite_added_mass_system_.rows() && mmrows > 0) {
            infinite_added_mass_system_.setZero(mmrows, mmrows);
            auto amrows = infinite_added_mass_.rows();
            infinite_added_mass_system_.block(0, 0, amrows, amrows) = infinite_added_mass_;
        }
        m_jacobians->M = infinite_added_mass_system_;
        m_jacobians->R.setZero();
        m_jacobians->K.setZero();
    }
    void LoadIntLoadResidual_Mv(chrono::ChVectorDynamic<>& R, const chrono::ChVectorDynamic<>& w, const double c) override {
        if (!this->m_jacobians) return;
        hydro_->AddedMassMv(R.data(), w.data(), c, static_cast<int>(R.size()));  // R += c*M*w on the GPU (:55-70)
    }

  private:
    bool IsStiff() override { return true; }
    TestHydro* hydro_;
    chrono::ChSystem* system_;
    chrono::ChMatrixDynamic<double> infinite_added_mass_, infinite_added_mass_system_;
};

// Wires a TestHydro into a ChSystem exactly as the reference's constructor does (src/hydro_forces.cpp:146-168,218-234):
// per body two WORLD_DIR ChForce objects ("hydroforce", "hydrotorque") fed by six ComponentFunc, plus the added-mass load.
class ChronoHydroSystem {
  public:
    // device_ids: one body-row shard per entry (multi-GPU inside this one process, see TestHydro); default = GPU 0 alone
    ChronoHydroSystem(std::vector<std::shared_ptr<chrono::ChBody>> bodies, const std::string& h5, std::shared_ptr<WaveBase> waves,
                      const std::vector<int>& device_ids = {0})
        : chbodies_(std::move(bodies)) {
        std::vector<std::shared_ptr<HydroBody>> views;
        for (auto& b : chbodies_) views.push_back(std::make_shared<ChronoBody>(b));
        hydro_ = std::make_unique<TestHydro>(views, h5, std::move(waves), device_ids);
        auto g = chbodies_[0]->GetSystem()->GetGravitationalAcceleration();
        hydro_->SetGravitationalAcceleration(g.x(), g.y(), g.z());
        for (size_t k = 0; k < chbodies_.size(); ++k) {
            const int bnum = hydro_->body_number(static_cast<int>(k));
            auto force = chrono_types::make_shared<chrono::ChForce>();
            auto torque = chrono_types::make_shared<chrono::ChForce>();
            force->SetAlign(chrono::ChForce::AlignmentFrame::WORLD_DIR);
            torque->SetAlign(chrono::ChForce::AlignmentFrame::WORLD_DIR);
            force->SetName("hydroforce");
            torque->SetName("hydrotorque");
            force->SetF_x(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 0));
            force->SetF_y(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 1));
            force->SetF_z(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 2));
            torque->SetF_x(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 3));
            torque->SetF_y(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 4));
            torque->SetF_z(chrono_types::make_shared<ComponentFunc>(hydro_.get(), bnum, 5));
            torque->SetMode(chrono::ChForce::ForceType::TORQUE);
            chbodies_[k]->AddForce(force);
            chbodies_[k]->AddForce(torque);
        }
        std::vector<std::shared_ptr<chrono::ChLoadable>> loadables(chbodies_.begin(), chbodies_.end());
        container_ = chrono_types::make_shared<chrono::ChLoadContainer>();
        load_      = chrono_types::make_shared<ChLoadAddedMass>(hydro_.get(), loadables, chbodies_[0]->GetSystem());
        chbodies_[0]->GetSystem()->Add(container_);
        container_->Add(load_);
    }
    TestHydro& hydro() { return *hydro_; }

  private:
    std::vector<std::shared_ptr<chrono::ChBody>> chbodies_;
    std::unique_ptr<TestHydro> hydro_;
    std::shared_ptr<chrono::ChLoadContainer> container_;
    std::shared_ptr<ChLoadAddedMass> load_;
};

}  // namespace hydroc_amd
#endif  // HYDROCHRONO_AMD_WITH_CHRONO
