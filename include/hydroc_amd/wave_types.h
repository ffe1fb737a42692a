// hydroc_amd/wave_types.h -- the reference's wave classes (include/hydroc/wave_types.h:40-467) as configuration holders over the
// C ABI: same class names, constructors, public members and getters; the arithmetic (AddH5Data + Initialize, GetForceAtTime) runs
// behind hc_set_wave_* / hc_compute_waves on the GPU.  Header-only; link with libhydrochrono_amd.so.
//
// A reference program changes `#include <hydroc/wave_types.h>` to `#include <hydroc_amd/wave_types.h>` and adds
// `using namespace hydroc_amd;` -- the wave set-up lines stay as they are (demos/sphere/demo_sphere_reg_waves.cpp:126-128,
// tests/regression/sphere/irreg_waves/sphere_irreg_waves_test.cpp:113-122).
//
// Not mirrored (off the force path): GetElevation / GetVelocity / GetAcceleration (wave kinematics), eta_file_path_
// (undefined behaviour in the reference, src/wave_types.cpp:480-500 vs :784-785).  The mesh helper the irregular demos call
// (SetUpWaveMesh / GetMeshFile / GetWaveMeshVelocity) is there so that they compile.
#pragma once

#include <algorithm>
#include <array>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hydrochrono_amd.h"
#include "../hydrochrono_amd_host.h"

namespace hydroc_amd {

// C status codes become the exception types the reference throws (std::runtime_error / std::out_of_range).
inline void check(hc_ctx* ctx, int rc) {
    if (rc == HC_OK) return;
    const std::string msg = hc_last_error(ctx);
    if (rc == HC_ERR_OUT_OF_RANGE) throw std::out_of_range(msg);
    throw std::runtime_error(msg);
}

enum class WaveMode { noWaveCIC = 0, regular = 1, irregular = 2 };  // include/hydroc/wave_types.h:40-47

// The two spectrum helpers the reference's header exports (include/hydroc/wave_types.h:14-20, src/wave_types.cpp:679-715), on the host
// routine the library itself uses for IrregularWaves (Eigen::VectorXd becomes std::vector<double>): spectral density in m^2/Hz at the
// frequencies f (Hz), which are sorted in place first, as the reference does (:681).
inline std::vector<double> JONSWAPSpectrumHz(std::vector<double>& f, double Hs, double Tp, double gamma = 3.3, bool is_normalized = false) {
    std::sort(f.begin(), f.end());
    std::vector<double> S(f.size());
    hc_host_jonswap_spectrum_hz(f.data(), static_cast<int>(f.size()), Hs, Tp, gamma, is_normalized ? 1 : 0, S.data());
    return S;
}
inline std::vector<double> PiersonMoskowitzSpectrumHz(std::vector<double>& f, double Hs, double Tp) { return JONSWAPSpectrumHz(f, Hs, Tp, 1.0, false); }

class WaveBase {  // :52-79
  public:
    virtual ~WaveBase()            = default;
    virtual void Initialize() {}   // the library initialises the model when it is attached
    virtual WaveMode GetWaveMode() = 0;
    // AddH5Data + Initialize of the reference, executed by the library for one (shard) context; TestHydro::AddWaves calls it
    virtual void Attach(hc_ctx* ctx) = 0;
    // 6 * num_bodies forces of the model at time t (GetForceAtTime of the reference returns an Eigen::VectorXd)
    std::vector<double> GetForceAtTime(double t) {
        if (!ctx_) throw std::runtime_error("wave model is not attached to a TestHydro");
        int N = 0, n_local = 0;
        check(ctx_, hc_get_sizes(ctx_, &N, &n_local, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        if (n_local != N) throw std::runtime_error("GetForceAtTime: ask the TestHydro of a sharded system (ComputeForceWaves)");
        std::vector<double> f(static_cast<size_t>(6) * N);
        check(ctx_, hc_compute_waves(ctx_, t, f.data()));
        return f;
    }
    double mwl_ = 0.0, g_ = 9.81, water_depth_ = 0.0;  // public members of the reference's base class (:74-78); unused by the force path

  protected:
    hc_ctx* ctx_ = nullptr;  // the first context the model was attached to (getters)
};

class NoWave : public WaveBase {  // :84-104
  public:
    NoWave() : num_bodies_(1) {}
    NoWave(unsigned int num_b) : num_bodies_(num_b) {}
    WaveMode GetWaveMode() override { return WaveMode::noWaveCIC; }
    void Attach(hc_ctx* ctx) override {
        check(ctx, hc_set_wave_none(ctx, static_cast<int>(num_bodies_)));
        ctx_ = ctx;
    }

  private:
    unsigned int num_bodies_;
};

class RegularWave : public WaveBase {  // :109-158
  public:
    RegularWave() : num_bodies_(1) {}
    RegularWave(unsigned int num_b) : num_bodies_(num_b) {}
    WaveMode GetWaveMode() override { return WaveMode::regular; }
    void Attach(hc_ctx* ctx) override {
        check(ctx, hc_set_wave_regular(ctx, static_cast<int>(num_bodies_), regular_wave_amplitude_, regular_wave_omega_));
        ctx_ = ctx;
    }
    // user input variables
    double regular_wave_amplitude_ = 0.0;
    double regular_wave_omega_     = 0.0;
    double regular_wave_phase_     = 0.0;  // unused by the force, as in the reference (src/wave_types.cpp:315-327)

  private:
    unsigned int num_bodies_;
};

struct IrregularWaveParams {  // :277-292
    unsigned int num_bodies_        = 1;
    double simulation_dt_           = 0.0;
    double simulation_duration_     = 0.0;
    double ramp_duration_           = 0.0;
    std::string eta_file_path_;     // not supported
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

class IrregularWaves : public WaveBase {  // :294-380
  public:
    IrregularWaves(const IrregularWaveParams& params) : params_(params) {}
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
    // Exporter inputs (src/wave_types.cpp:461-478, read by the runner at run_hydrochrono_from_yaml.cpp:668-679).  GetSpectrum returns
    // the spectral densities S(f) its comment promises; the reference returns a member it never fills ("do not reproduce").
    std::vector<double> GetSpectrum() { return spectrum(1); }
    std::vector<double> GetFreeSurfaceElevation() { return table(false); }
    std::vector<double> GetFreeSurfaceTime() const { return table(true); }
    std::vector<double> GetFrequenciesHz() const { return spectrum(0); }
    // Visualisation helper the reference's irregular-wave demos call (demos/sphere/demo_sphere_irreg_waves.cpp:144-153,
    // src/wave_types.cpp:846-864): the free-surface elevation over the simulated time as a Wavefront OBJ ribbon (x = -t, y = -10 / +10,
    // z = eta(t)) that the demo drags past the body with GetWaveMeshVelocity().  Off the force path; written from the eta(t) table the
    // force path uses, against the table's own time stamps (t >= 0).
    void SetUpWaveMesh(std::string filename = "fse_mesh.obj") {
        mesh_file_name_ = std::move(filename);
        const std::vector<double> t = table(true), eta = table(false);
        std::FILE* out = std::fopen(mesh_file_name_.c_str(), "w");
        if (!out) throw std::runtime_error("SetUpWaveMesh: cannot write " + mesh_file_name_);
        std::fprintf(out, "# free-surface elevation ribbon (hydroc_amd)\n");
        size_t n = 0;
        for (size_t i = 0; i < t.size(); ++i) {
            if (t[i] < 0.0 || t[i] > params_.simulation_duration_) continue;
            std::fprintf(out, "v %.6f %.6f %.6f\nv %.6f %.6f %.6f\n", -t[i], -10.0, eta[i], -t[i], 10.0, eta[i]);
            ++n;
        }
        for (size_t i = 0; i + 1 < n; ++i)  // two triangles per step of the ribbon (OBJ indices are 1-based)
            std::fprintf(out, "f %zu %zu %zu\nf %zu %zu %zu\n", 2 * i + 1, 2 * i + 2, 2 * i + 4, 2 * i + 1, 2 * i + 4, 2 * i + 3);
        std::fclose(out);
    }
    std::string GetMeshFile() { return mesh_file_name_; }
    std::array<double, 3> GetWaveMeshVelocity() { return {1.0, 0.0, 0.0}; }  // (an Eigen::Vector3d in the reference; ChVector3d(v[0], v[1], v[2]))

  private:
    std::vector<double> spectrum(int which) const {
        need_ctx();
        int nf = 0;
        check(ctx_, hc_get_sizes(ctx_, nullptr, nullptr, nullptr, nullptr, &nf, nullptr, nullptr, nullptr));
        std::vector<double> v(nf);
        check(ctx_, which == 0 ? hc_get_spectrum(ctx_, v.data(), nullptr, nullptr, nullptr, nullptr)
                               : hc_get_spectrum(ctx_, nullptr, v.data(), nullptr, nullptr, nullptr));
        return v;
    }
    std::vector<double> table(bool time) const {
        need_ctx();
        int nt = 0;
        check(ctx_, hc_get_sizes(ctx_, nullptr, nullptr, nullptr, nullptr, nullptr, &nt, nullptr, nullptr));
        std::vector<double> v(nt);
        check(ctx_, time ? hc_get_eta_table(ctx_, v.data(), nullptr) : hc_get_eta_table(ctx_, nullptr, v.data()));
        return v;
    }
    void need_ctx() const {
        if (!ctx_) throw std::runtime_error("IrregularWaves is not attached to a TestHydro");
    }
    IrregularWaveParams params_;
    std::string mesh_file_name_;
};

}  // namespace hydroc_amd
