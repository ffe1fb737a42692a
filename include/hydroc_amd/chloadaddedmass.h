// hydroc_amd/chloadaddedmass.h -- ChLoadAddedMass of the reference (include/hydroc/chloadaddedmass.h:22-90,
// src/chloadaddedmass.cpp:12-70): the rho-scaled 6N x 6N infinite-frequency added mass as a stiff ChLoadCustomMultiple.  The
// Jacobian block comes from hc_added_mass_matrix.  `R += c*M*w` (Chrono's LoadIntLoadResidual_Mv, at least once per step) runs on the
// GPU (hc_added_mass_mv / _multi) for systems of more than kHostProductMaxDofs coordinates and on this object's own host copy of
// the matrix -- the one it keeps for the Jacobian -- below: a GPU product is a PCIe round trip of 5.3 - 9.6 us whatever its size
// (6 x 6 to 384 x 384), the reference's host product takes 0.02 us at 6 x 6, 0.07 at 12 x 12, 4 at 96 x 96 and 85 at 384 x 384
// (bench.py: added_mass_mv; MEASURED.md).  The C ABI entry itself always runs on the GPU.  Created by the
// TestHydro constructor; included by hydro_forces.h (needs Project Chrono, or the stand-in headers of tests/cpp/chrono_stub).
#pragma once

#include "h5fileinfo.h"
#include "hydro_forces.h"

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
namespace hydroc_amd {

class ChLoadAddedMass : public chrono::ChLoadCustomMultiple {
  public:
    // 6N up to which LoadIntLoadResidual_Mv multiplies on the host (16 bodies; bench.py reports the measured crossover beside it)
    static constexpr int kHostProductMaxDofs = 96;
    // (the reference's constructor takes the per-body file data, :12-25; here the matrix is asked from the object that read the file)
    ChLoadAddedMass(TestHydro* hydro, std::vector<std::shared_ptr<chrono::ChLoadable>>& bodies, chrono::ChSystem* system)
        : chrono::ChLoadCustomMultiple(bodies), hydro_(hydro), system_(system) {
        const int D  = 6 * hydro_->num_bodies();
        const auto M = hydro_->GetAddedMassMatrix();
        infinite_added_mass_.setZero(D, D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) infinite_added_mass_(i, j) = M[static_cast<size_t>(i) * D + j];
        infinite_added_mass_system_ = infinite_added_mass_;
    }
    // The reference's public constructor (include/hydroc/chloadaddedmass.h:33-35, src/chloadaddedmass.cpp:12-25; used on its own by
    // tests/chloadaddedmass_t01.cpp:44-58): the per-body chunks of a file read with H5FileInfo.  The load owns a small device context
    // that holds nothing but the stacked 6N x 6N matrix (inf_added_mass is rho-scaled already, so the context's rho is 1; its radiation
    // kernel is two zero samples and is never evaluated), and `R += c*M*w` runs there.
    ChLoadAddedMass(const std::vector<HydroData::BodyInfo>& user_h5_body_data, std::vector<std::shared_ptr<chrono::ChLoadable>>& bodies,
                    chrono::ChSystem* system, int device_id = 0)
        : chrono::ChLoadCustomMultiple(bodies), hydro_(nullptr), system_(system) {
        const int N = static_cast<int>(user_h5_body_data.size());
        if (N <= 0) throw std::runtime_error("ChLoadAddedMass: no body data");
        const int D  = 6 * N;
        hc_ctx* raw  = nullptr;
        if (hc_create(N, device_id, &raw) != HC_OK) throw std::runtime_error(hc_last_error(nullptr));
        own_ctx_ = std::shared_ptr<hc_ctx>(raw, [](hc_ctx* c) { hc_destroy(c); });
        check(raw, hc_set_simulation_parameters(raw, 1.0, 9.81, 0.0));
        const double t2[2] = {0.0, 1.0};
        const std::vector<double> k0(static_cast<size_t>(6) * D * 2, 0.0);
        for (int b = 0; b < N; ++b) {
            const HydroData::BodyInfo& q = user_h5_body_data[static_cast<size_t>(b)];
            if (q.inf_added_mass.rows() != 6 || q.inf_added_mass.cols() != D || q.cg.size() < 3 || q.cb.size() < 3 || q.lin_matrix.rows() != 6 ||
                q.lin_matrix.cols() != 6)
                throw std::runtime_error("ChLoadAddedMass: body data of body " + std::to_string(b) + " does not have the shapes of a " +
                                         std::to_string(N) + "-body file");
            check(raw, hc_set_body_properties(raw, b, q.disp_vol, q.cg.data(), q.cb.data()));
            check(raw, hc_set_hydrostatic_stiffness(raw, b, q.lin_matrix.data()));
            check(raw, hc_set_added_mass_inf(raw, b, q.inf_added_mass.data()));
            check(raw, hc_set_rirf(raw, b, t2, 2, k0.data()));
        }
        check(raw, hc_finalize(raw));
        std::vector<double> M(static_cast<size_t>(D) * D);
        check(raw, hc_added_mass_matrix(raw, M.data()));
        infinite_added_mass_.setZero(D, D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) infinite_added_mass_(i, j) = M[static_cast<size_t>(i) * D + j];
        infinite_added_mass_system_ = infinite_added_mass_;
    }
    ChLoadAddedMass* Clone() const override { return new ChLoadAddedMass(*this); }
    void ComputeQ(chrono::ChState*, chrono::ChStateDelta*) override {}
    // src/chloadaddedmass.cpp:27-53: M = the added-mass block at (0, 0) of a system-sized matrix (the hydro bodies come first in
    // Chrono's state ordering), R = K = 0
    void ComputeJacobian(chrono::ChState*, chrono::ChStateDelta*) override {
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
    // :55-70: R += c * M * w
    void LoadIntLoadResidual_Mv(chrono::ChVectorDynamic<>& R, const chrono::ChVectorDynamic<>& w, const double c) override {
        if (!this->m_jacobians) return;
        const int D = static_cast<int>(infinite_added_mass_.rows());
        if (D <= host_product_limit_ && R.size() >= D && w.size() >= D) {
            // small system: the product on the host copy (src/chloadaddedmass.cpp:70 is the same Eigen expression on the system-sized
            // matrix, whose rows and columns beyond 6N are zero)
            for (int i = 0; i < D; ++i) {
                double s = 0.0;
                for (int j = 0; j < D; ++j) s += infinite_added_mass_(i, j) * w(j);
                R(i) += c * s;
            }
            return;
        }
        if (hydro_) hydro_->AddedMassMv(R.data(), w.data(), c, static_cast<int>(R.size()));
        else check(own_ctx_.get(), hc_added_mass_mv(own_ctx_.get(), w.data(), c, R.data(), static_cast<int>(R.size())));
    }

    // Not in the reference: up to how many coordinates the product stays on the host (default kHostProductMaxDofs; 0 = always the GPU)
    void SetHostProductLimit(int dofs) { host_product_limit_ = dofs; }

  private:
    bool IsStiff() override { return true; }
    int host_product_limit_ = kHostProductMaxDofs;
    TestHydro* hydro_;                 // the object that read the file (the load it creates itself), or null:
    std::shared_ptr<hc_ctx> own_ctx_;  // ... a context of the load's own (the constructor over HydroData::BodyInfo; clones share it)
    chrono::ChSystem* system_;
    chrono::ChMatrixDynamic<double> infinite_added_mass_, infinite_added_mass_system_;
};

// The ChBody constructor of TestHydro (include/hydroc/hydro_forces.h:178-180), defined here because it creates the load.
inline TestHydro::TestHydro(std::vector<std::shared_ptr<chrono::ChBody>> user_bodies, std::string h5_file_name,
                            std::shared_ptr<WaveBase> waves, const std::vector<int>& device_ids)
    : bodies_(views_of(user_bodies)), num_bodies_(static_cast<int>(user_bodies.size())), chbodies_(std::move(user_bodies)) {
    create_contexts(h5_file_name, std::move(waves), device_ids);
    try {
        wire_into_chrono();
    } catch (...) {
        destroy_contexts();
        throw;
    }
}

// src/hydro_forces.cpp:218-234: a ForceFunc6d per body, the added-mass load in a container of the bodies' system
inline void TestHydro::wire_into_chrono() {
    system_ = chbodies_[0]->GetSystem();
    if (!system_) throw std::runtime_error("TestHydro: add the bodies to a ChSystem before constructing the hydro forces");
    follow_system_gravity();
    for (auto& b : chbodies_) force_per_body_.push_back(std::make_unique<ForceFunc6d>(b, this));
    my_loadcontainer = chrono_types::make_shared<chrono::ChLoadContainer>();
    std::vector<std::shared_ptr<chrono::ChLoadable>> loadables(chbodies_.begin(), chbodies_.end());
    my_loadbodyinertia = chrono_types::make_shared<ChLoadAddedMass>(this, loadables, system_);
    system_->Add(my_loadcontainer);
    my_loadcontainer->Add(my_loadbodyinertia);
}

}  // namespace hydroc_amd
#endif
