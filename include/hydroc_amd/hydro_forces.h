// hydroc_amd/hydro_forces.h -- the reference's plugin surface for the hydro-force path (include/hydroc/hydro_forces.h:45-285) over the
// C ABI (include/hydrochrono_amd.h).  Header-only; link with libhydrochrono_amd.so.
//
// Source compatibility is the point of this file.  A program written against the reference
//
//     #include <hydroc/hydro_forces.h>                               ->   #include <hydroc_amd/hydro_forces.h>
//                                                                         using namespace hydroc_amd;
//     std::vector<std::shared_ptr<ChBody>> bodies;
//     bodies.push_back(sphereBody);
//     TestHydro hydro_forces(bodies, h5fname);                            (unchanged; demos/sphere/demo_sphere_reg_waves.cpp:130-133)
//     hydro_forces.AddWaves(my_hydro_inputs);                             (unchanged)
//
// keeps its hydro lines: with Project Chrono on the include path the constructor does what the reference's does
// (src/hydro_forces.cpp:170-242) -- BEMIO-HDF5 ingest, one ForceFunc6d per body (two WORLD_DIR ChForce objects "hydroforce" /
// "hydrotorque" fed by six ComponentFunc, added to the body, :96-168), ChLoadAddedMass in a ChLoadContainer added to the bodies'
// ChSystem (:223-234), AddWaves -- and every Chrono update then reaches the GPU as ONE hc_step (or hc_step_multi) per distinct
// time (CoordinateFuncForBody, :727-767).  Same names, argument meaning and error behaviour:
//   TestHydro(bodies, h5_file, waves = NoWave), AddWaves, GetWave, ComputeForceHydrostatics / RadiationDampingConv / Waves,
//   GetRIRFval, SetRadiationConvolutionMode, SetTaperedDirectOptions, SetDiagnosticsOutputDirectory, CoordinateFuncForBody,
//   GetProfileStats; ComponentFunc(ForceFunc6d*, i), ForceFunc6d(body, TestHydro*)::CoordinateFunc(i); HydroProfileStats.
// Additions (not in the reference): an optional trailing `device_ids` argument -- one body-row shard per listed GPU inside this one
// process -- and SetPassSchedule.
//
// Without Chrono (drivers, tests, the examples/) bodies are seen through the small `BodyView` interface -- exactly the ChBody
// getters the reference calls (src/hydro_forces.cpp:106-107,279-280,550,567-568); `MockBody` implements it with plain fields.
// Deliberate deviations from the reference are listed in DESIGN.md 1 (default NoWave() with more than one body is an error
// instead of an out-of-bounds read; a step back in time is handled).
#pragma once

#include <algorithm>
#include <array>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hydrochrono_amd.h"
#include "wave_types.h"

#if !defined(HYDROCHRONO_AMD_WITH_CHRONO) && defined(__has_include)
#if __has_include(<chrono/physics/ChBody.h>)
#define HYDROCHRONO_AMD_WITH_CHRONO 1
#endif
#endif

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
#include <chrono/functions/ChFunction.h>
#include <chrono/physics/ChBody.h>
#include <chrono/physics/ChForce.h>
#include <chrono/physics/ChLoad.h>
#include <chrono/physics/ChLoadContainer.h>
#include <chrono/physics/ChSystem.h>
#endif

namespace hydroc_amd {

// ---------------------------------------------------------------------------------------------------------------
// Body view: what the force path reads from a body
// ---------------------------------------------------------------------------------------------------------------
struct BodyView {
    virtual ~BodyView()                                      = default;
    virtual std::string GetName() const                      = 0;  // "body<k>", 1-based (src/hydro_forces.cpp:106-107)
    virtual double GetChTime() const                         = 0;
    virtual std::array<double, 3> GetPos() const             = 0;
    virtual std::array<double, 3> GetCardanAnglesXYZ() const = 0;  // GetRot().GetCardanAnglesXYZ()
    virtual std::array<double, 3> GetPosDt() const           = 0;
    virtual std::array<double, 3> GetAngVelParent() const    = 0;
};

struct MockBody : BodyView {
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

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
struct ChronoBody : BodyView {
    std::shared_ptr<chrono::ChBody> body;
    explicit ChronoBody(std::shared_ptr<chrono::ChBody> b) : body(std::move(b)) {}
    std::string GetName() const override { return body->GetName(); }
    double GetChTime() const override { return body->GetChTime(); }
    std::array<double, 3> GetPos() const override { auto v = body->GetPos(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetCardanAnglesXYZ() const override { auto v = body->GetRot().GetCardanAnglesXYZ(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetPosDt() const override { auto v = body->GetPosDt(); return {v.x(), v.y(), v.z()}; }
    std::array<double, 3> GetAngVelParent() const override { auto v = body->GetAngVelParent(); return {v.x(), v.y(), v.z()}; }
};

class TestHydro;
class ForceFunc6d;
class ChLoadAddedMass;

// ComponentFunc (include/hydroc/hydro_forces.h:45-86): one degree of freedom of a body's hydro force as a ChFunction.  GetVal's
// argument is ignored -- the time comes from the first body (src/hydro_forces.cpp:79-85,739).
class ComponentFunc : public chrono::ChFunction {
  public:
    ComponentFunc() : base_(nullptr), index_(6) {}
    ComponentFunc(ForceFunc6d* b, int i) : base_(b), index_(i) {}
    ComponentFunc(const ComponentFunc& old) : chrono::ChFunction(old), base_(old.base_), index_(old.index_) {}
    ComponentFunc* Clone() const override { return new ComponentFunc(*this); }
    double GetVal(double x) const override;

  private:
    ForceFunc6d* base_;
    int index_;
};

// ForceFunc6d (include/hydroc/hydro_forces.h:91-148): the six components of one body, wired as two ChForce objects.  Owned by
// TestHydro behind a stable address (the reference re-points its ComponentFunc array after vector growth, :117-134; here the
// objects never move, so the class is not copyable).
class ForceFunc6d {
  public:
    ForceFunc6d(std::shared_ptr<chrono::ChBody> object, TestHydro* all_hydro_forces_user);
    ForceFunc6d(const ForceFunc6d&)            = delete;
    ForceFunc6d& operator=(const ForceFunc6d&) = delete;
    double CoordinateFunc(int i);
    int body_number() const { return b_num_; }

  private:
    std::shared_ptr<chrono::ChBody> body_;
    int b_num_;  // 1-based, from the body name "body<k>"
    std::shared_ptr<ComponentFunc> force_ptrs_[6];
    std::shared_ptr<chrono::ChForce> chrono_force_, chrono_torque_;
    TestHydro* all_hydro_forces_;
};
#endif  // HYDROCHRONO_AMD_WITH_CHRONO

struct HydroProfileStats {  // include/hydroc/hydro_forces.h:153-160
    double hydrostatics_seconds = 0.0, radiation_seconds = 0.0, waves_seconds = 0.0;
    int hydrostatics_calls = 0, radiation_calls = 0, waves_calls = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// TestHydro
// ---------------------------------------------------------------------------------------------------------------
class TestHydro {
  public:
    TestHydro()                            = delete;
    TestHydro(const TestHydro&)            = delete;
    TestHydro& operator=(const TestHydro&) = delete;

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
    // The reference's constructor (include/hydroc/hydro_forces.h:178-180, src/hydro_forces.cpp:170-242): reads the h5 file, wires the
    // forces and the added-mass load into the bodies' ChSystem, attaches the waves.
    TestHydro(std::vector<std::shared_ptr<chrono::ChBody>> user_bodies, std::string h5_file_name,
              std::shared_ptr<WaveBase> waves = std::make_shared<NoWave>())
        : TestHydro(std::move(user_bodies), std::move(h5_file_name), std::move(waves), std::vector<int>{0}) {}
    // ... on several GPUs: one body-row shard per entry of `device_ids` (a device may be listed more than once)
    TestHydro(std::vector<std::shared_ptr<chrono::ChBody>> user_bodies, std::string h5_file_name, std::shared_ptr<WaveBase> waves,
              const std::vector<int>& device_ids);
#endif

    TestHydro(std::vector<std::shared_ptr<BodyView>> user_bodies, const std::string& h5_file_name,
              std::shared_ptr<WaveBase> waves = std::make_shared<NoWave>(), int device_id = 0)
        : TestHydro(std::move(user_bodies), h5_file_name, std::move(waves), std::vector<int>{device_id}) {}
    // Multi-GPU inside the one Chrono process (drop-in variant): contiguous balanced split of the bodies over the
    // shards; every call below fans out to the shard contexts, the per-step evaluation goes through hc_step_multi (all GPUs
    // started before any is waited for, host-side gather).  One entry = the single-GPU object.
    TestHydro(std::vector<std::shared_ptr<BodyView>> user_bodies, const std::string& h5_file_name, std::shared_ptr<WaveBase> waves,
              const std::vector<int>& device_ids)
        : bodies_(std::move(user_bodies)), num_bodies_(static_cast<int>(bodies_.size())) {
        create_contexts(h5_file_name, std::move(waves), device_ids);
    }
    // Adopts contexts that are already configured (hc_create_from_hydro_yaml[_sharded], or a C caller's own set-up): together they
    // own bodies [0, N).  Their configuration -- pass schedule included -- is left as it is.  The contexts belong to the object
    // from the call on, also when it throws.
    TestHydro(std::vector<std::shared_ptr<BodyView>> user_bodies, hc_ctx* configured_ctx)
        : TestHydro(std::move(user_bodies), std::vector<hc_ctx*>{configured_ctx}) {}
    TestHydro(std::vector<std::shared_ptr<BodyView>> user_bodies, std::vector<hc_ctx*> configured_ctxs)
        : bodies_(std::move(user_bodies)), num_bodies_(static_cast<int>(bodies_.size())), ctxs_(std::move(configured_ctxs)) {
        try {
            if (ctxs_.empty() || bodies_.empty()) throw std::runtime_error("TestHydro: no context / no body");
            ctx_ = ctxs_[0];
            read_body_numbers();
            total_force_.assign(6 * static_cast<size_t>(num_bodies_), 0.0);
        } catch (...) {
            destroy_contexts();
            throw;
        }
    }
    ~TestHydro() { destroy_contexts(); }

    void AddWaves(std::shared_ptr<WaveBase> waves) {  // src/hydro_forces.cpp:244-261
        user_waves_ = std::move(waves);
        for (auto it = ctxs_.rbegin(); it != ctxs_.rend(); ++it) user_waves_->Attach(*it);  // (the wave object keeps the first context for its getters)
    }
    std::shared_ptr<WaveBase> GetWave() const { return user_waves_; }
    // ChSystem::GetGravitationalAcceleration(), which the reference reads in every hydrostatics evaluation (:267-269).  The
    // ChBody constructors follow the system's value by themselves; Chrono-free drivers set it here (default (0, 0, -9.81)).
    void SetGravitationalAcceleration(double gx, double gy, double gz) {
        const double g[3] = {gx, gy, gz};
        for (hc_ctx* c : ctxs_) check(c, hc_set_gravity(c, g));
        gravity_ = {gx, gy, gz};
    }

    enum class RadiationConvolutionMode { Baseline, TaperedDirect };
    void SetRadiationConvolutionMode(RadiationConvolutionMode mode) {
        for (hc_ctx* c : ctxs_) check(c, hc_set_convolution_mode(c, mode == RadiationConvolutionMode::TaperedDirect ? 1 : 0));
    }
    struct TaperedDirectOptions {  // include/hydroc/hydro_forces.h:246-259
        std::string smoothing        = "sg";  // "sg" (Savitzky-Golay) or "moving_average"
        int window_length            = 5;
        double rirf_end_time         = -1.0;
        double taper_start_percent   = 0.8;
        double taper_end_percent     = 1.0;
        double taper_final_amplitude = 0.0;
        bool export_plot_csv         = false;  // rirf_body<b>_summary.csv in the diagnostics directory (src/hydro_forces.cpp:509-531)
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
    void SetDiagnosticsOutputDirectory(const std::string& dir) {  // include/hydroc/hydro_forces.h:269
        for (hc_ctx* x : ctxs_) check(x, hc_set_diagnostics_output_directory(x, dir.c_str()));
    }

    // Not in the reference (it has no such notion): when the look-ahead pass of a block runs, see hc_set_pass_schedule.  The contexts
    // come with the library's ADAPTIVE schedule (one_block_ahead < 0): a Chrono loop, which does its own work between two force
    // evaluations, gets the pass of every block one block ahead, beside the steps (64 bodies, 30 / 100 us of host work between calls:
    // 12.8 / 12.7 us per step instead of 17.4 / 15.6, and no step waits for a whole pass); a driver that steps back to back gets it at
    // block start.  0 / 1 pin a schedule.
    void SetPassSchedule(int one_block_ahead, int slices = 0) {
        for (hc_ctx* x : ctxs_) check(x, hc_set_pass_schedule(x, one_block_ahead < 0 ? -1 : (one_block_ahead ? 1 : 0), slices));
    }

    std::vector<double> ComputeForceHydrostatics() {
        gather_state();
        std::vector<double> out(6 * static_cast<size_t>(num_bodies_));
        for (hc_ctx* c : ctxs_) check(c, hc_compute_hydrostatics(c, pos_.data(), rpy_.data(), out.data() + row0(c)));
        return out;
    }
    std::vector<double> ComputeForceRadiationDampingConv() {
        gather_state();
        std::vector<double> out(6 * static_cast<size_t>(num_bodies_));
        for (hc_ctx* c : ctxs_) check(c, hc_compute_radiation(c, bodies_[0]->GetChTime(), lin_.data(), ang_.data(), out.data() + row0(c)));
        return out;
    }
    std::vector<double> ComputeForceWaves() {  // (an Eigen::VectorXd in the reference)
        std::vector<double> out(6 * static_cast<size_t>(num_bodies_));
        for (hc_ctx* c : ctxs_) check(c, hc_compute_waves(c, bodies_[0]->GetChTime(), out.data() + row0(c)));
        return out;
    }
    // src/hydro_forces.cpp:693-711: the radiation IRF value the convolution uses (rho-scaled; the processed kernel in
    // TaperedDirect mode).  Reads one value back from the GPU -- a debugging accessor, as in the reference.
    double GetRIRFval(int row, int col, int st) {
        double v = 0.0;
        const int b = row / 6;
        if (row < 0 || b >= num_bodies_) throw std::out_of_range("GetRIRFval: row index out of range");
        for (hc_ctx* c : ctxs_) {
            int b0 = 0, b1 = 0;
            check(c, hc_get_shard(c, &b0, &b1));
            if (b >= b0 && b < b1) check(c, hc_get_rirf_value(c, row - 6 * b0, col, st, &v));
        }
        return v;
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
        return total_force_[6 * static_cast<size_t>(b - 1) + dof_index];
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
    // file -> shard contexts -> waves; on failure nothing is left behind
    void create_contexts(const std::string& h5_file_name, std::shared_ptr<WaveBase> waves, const std::vector<int>& device_ids) {
        if (bodies_.empty()) throw std::runtime_error("TestHydro needs at least one body");
        if (device_ids.empty() || static_cast<int>(device_ids.size()) > num_bodies_)
            throw std::runtime_error("TestHydro: between one shard and one shard per body");
        const int G = static_cast<int>(device_ids.size()), base = num_bodies_ / G, extra = num_bodies_ % G;
        try {
            read_body_numbers();
            for (int g = 0; g < G; ++g) {
                const int b0 = g * base + (g < extra ? g : extra), b1 = b0 + base + (g < extra ? 1 : 0);
                hc_ctx* c = nullptr;
                if (hc_create_sharded(num_bodies_, b0, b1, device_ids[g], &c) != HC_OK) throw std::runtime_error(hc_last_error(nullptr));
                ctxs_.push_back(c);
                check(c, hc_load_bemio_h5(c, h5_file_name.c_str()));
                check(c, hc_finalize(c));
            }
            ctx_ = ctxs_[0];
            if (!waves) waves = std::make_shared<NoWave>(static_cast<unsigned>(num_bodies_));
            AddWaves(std::move(waves));
        } catch (...) {
            destroy_contexts();
            throw;
        }
        total_force_.assign(6 * static_cast<size_t>(num_bodies_), 0.0);
    }
    void destroy_contexts() {
        for (hc_ctx* c : ctxs_)
            if (c) hc_destroy(c);
        ctxs_.clear();
        ctx_ = nullptr;
    }
    // body numbers come from the names "body<k>", 1-based (ForceFunc6d ctor, src/hydro_forces.cpp:104-108)
    void read_body_numbers() {
        for (auto& b : bodies_) {
            std::string temp = b->GetName();
            body_numbers_.push_back(std::stoi(temp.erase(0, 4)));
        }
    }
    void gather_state() {
#ifdef HYDROCHRONO_AMD_WITH_CHRONO
        follow_system_gravity();
#endif
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
    std::vector<std::shared_ptr<BodyView>> bodies_;
    int num_bodies_;
    std::vector<int> body_numbers_;
    std::vector<hc_ctx*> ctxs_;  // one per body-row shard (one = the single-GPU object)
    hc_ctx* ctx_ = nullptr;      // ctxs_[0]
    std::shared_ptr<WaveBase> user_waves_;
    std::vector<double> total_force_, pos_, rpy_, lin_, ang_;
    std::array<double, 3> gravity_{0.0, 0.0, -9.81};
    bool have_time_   = false;
    double prev_time_ = -1.0;

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
    static std::vector<std::shared_ptr<BodyView>> views_of(const std::vector<std::shared_ptr<chrono::ChBody>>& chbodies) {
        std::vector<std::shared_ptr<BodyView>> views;
        for (auto& b : chbodies) views.push_back(std::make_shared<ChronoBody>(b));
        return views;
    }
    // the reference asks the system for g at every evaluation (src/hydro_forces.cpp:267-269); a changed value reaches the contexts
    // before the next one (hc_set_gravity waits for the queue, so it is not called while nothing changes)
    void follow_system_gravity() {
        if (!system_) return;
        const auto g = system_->GetGravitationalAcceleration();
        if (g.x() != gravity_[0] || g.y() != gravity_[1] || g.z() != gravity_[2]) SetGravitationalAcceleration(g.x(), g.y(), g.z());
    }
    void wire_into_chrono();
    std::vector<std::shared_ptr<chrono::ChBody>> chbodies_;
    chrono::ChSystem* system_ = nullptr;
    std::vector<std::unique_ptr<ForceFunc6d>> force_per_body_;
    std::shared_ptr<chrono::ChLoadContainer> my_loadcontainer;
    std::shared_ptr<ChLoadAddedMass> my_loadbodyinertia;
#endif
};

}  // namespace hydroc_amd

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
namespace hydroc_amd {

inline double ComponentFunc::GetVal(double) const { return base_ ? base_->CoordinateFunc(index_) : 0.0; }

inline ForceFunc6d::ForceFunc6d(std::shared_ptr<chrono::ChBody> object, TestHydro* all_hydro_forces_user)
    : body_(std::move(object)), all_hydro_forces_(all_hydro_forces_user) {
    std::string temp = body_->GetName();  // "body<k>" -> k
    b_num_           = std::stoi(temp.erase(0, 4));
    for (int i = 0; i < 6; ++i) force_ptrs_[i] = chrono_types::make_shared<ComponentFunc>(this, i);
    chrono_force_  = chrono_types::make_shared<chrono::ChForce>();
    chrono_torque_ = chrono_types::make_shared<chrono::ChForce>();
    chrono_force_->SetAlign(chrono::ChForce::AlignmentFrame::WORLD_DIR);
    chrono_torque_->SetAlign(chrono::ChForce::AlignmentFrame::WORLD_DIR);
    chrono_force_->SetName("hydroforce");
    chrono_torque_->SetName("hydrotorque");
    chrono_force_->SetF_x(force_ptrs_[0]);
    chrono_force_->SetF_y(force_ptrs_[1]);
    chrono_force_->SetF_z(force_ptrs_[2]);
    chrono_torque_->SetF_x(force_ptrs_[3]);
    chrono_torque_->SetF_y(force_ptrs_[4]);
    chrono_torque_->SetF_z(force_ptrs_[5]);
    chrono_torque_->SetMode(chrono::ChForce::ForceType::TORQUE);
    body_->AddForce(chrono_force_);
    body_->AddForce(chrono_torque_);
}

inline double ForceFunc6d::CoordinateFunc(int i) {
    if (i < 0 || i >= 6) return 0.0;  // the reference prints a message and returns 0 (src/hydro_forces.cpp:136-144)
    return all_hydro_forces_->CoordinateFuncForBody(b_num_, i);
}

// Round-3 name of "a TestHydro wired into a ChSystem", kept for callers written against it.
class ChronoHydroSystem {
  public:
    ChronoHydroSystem(std::vector<std::shared_ptr<chrono::ChBody>> bodies, const std::string& h5, std::shared_ptr<WaveBase> waves,
                      const std::vector<int>& device_ids = {0})
        : hydro_(std::move(bodies), h5, std::move(waves), device_ids) {}
    TestHydro& hydro() { return hydro_; }

  private:
    TestHydro hydro_;
};

}  // namespace hydroc_amd

#include "chloadaddedmass.h"  // ChLoadAddedMass and the part of TestHydro that creates it
#endif  // HYDROCHRONO_AMD_WITH_CHRONO
