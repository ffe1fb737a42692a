// hydroc_amd/chloadaddedmass.h -- ChLoadAddedMass of the reference (include/hydroc/chloadaddedmass.h:22-90,
// src/chloadaddedmass.cpp:12-70): the rho-scaled 6N x 6N infinite-frequency added mass as a stiff ChLoadCustomMultiple.  The
// Jacobian block comes from hc_added_mass_matrix, `R += c*M*w` runs on the GPU (hc_added_mass_mv / _multi).  Created by the
// TestHydro constructor; included by hydro_forces.h (needs Project Chrono, or the stand-in headers of tests/cpp/chrono_stub).
#pragma once

#include "hydro_forces.h"

#ifdef HYDROCHRONO_AMD_WITH_CHRONO
namespace hydroc_amd {

class ChLoadAddedMass : public chrono::ChLoadCustomMultiple {
  public:
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
        hydro_->AddedMassMv(R.data(), w.data(), c, static_cast<int>(R.size()));
    }

  private:
    bool IsStiff() override { return true; }
    TestHydro* hydro_;
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
