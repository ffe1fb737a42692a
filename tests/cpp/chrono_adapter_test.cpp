// Drives the guarded Project Chrono adapter block of include/hydroc_amd/hydro_forces.h (ComponentFunc, the two
// WORLD_DIR ChForce objects per body, ChLoadAddedMass in a ChLoadContainer -- src/hydro_forces.cpp:63-168,223-234,
// src/chloadaddedmass.cpp:27-70) against the stand-in Chrono headers under tests/cpp/chrono_stub/ (test infrastructure).
//   usage: chrono_adapter_test <sphere.h5> <nsteps>
// Runs the reference's sphere decay test (tests/regression/sphere/demo_sphere_decay.cpp:52-120: z0 = -1, dt = 0.015) with
// the force read through ChForce -> ComponentFunc::GetVal (six callbacks per step, one evaluation) and the added mass
// through ChLoadAddedMass::ComputeJacobian; prints "t z" per step, then the LoadIntLoadResidual_Mv check.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define HYDROCHRONO_AMD_WITH_CHRONO 1
#include "../../include/hydroc_amd/setup_hydro_from_yaml.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int nsteps = std::atoi(argv[2]);
    using namespace chrono;
    ChSystem system;
    system.SetGravitationalAcceleration(ChVector3d(0, 0, -9.81));
    auto sphere = chrono_types::make_shared<ChBody>();
    sphere->SetName("body1");
    sphere->pos = ChVector3d(0, 0, -1.0);
    system.AddBody(sphere);
    // one extra non-hydro body: the system has more coordinates than the added-mass block (src/chloadaddedmass.cpp:35-44)
    auto other = chrono_types::make_shared<ChBody>();
    other->SetName("ground");
    system.AddBody(other);

    try {
        hydroc_amd::ChronoHydroSystem hydro({sphere}, argv[1], std::make_shared<hydroc_amd::NoWave>(1));
        if (sphere->forces.size() != 2 || sphere->forces[0]->name != "hydroforce" || sphere->forces[1]->name != "hydrotorque") return 3;
        if (sphere->forces[1]->mode != ChForce::ForceType::TORQUE || sphere->forces[0]->align != ChForce::AlignmentFrame::WORLD_DIR) return 3;
        if (system.containers.size() != 1 || system.containers[0]->loads.size() != 1) return 3;
        auto load = system.containers[0]->loads[0];
        load->StubUpdate(system.GetNumCoordsVelLevel());
        const auto& M = load->m_jacobians->M;
        if (M.rows() != 12 || !load->IsStiff()) return 4;
        const double a33 = M(2, 2);
        const double mass = 261.8e3, g = 9.81, dt = 0.015;
        double z = -1.0, v = 0.0;
        for (int n = 0; n < nsteps; ++n) {
            system.time     = n * dt;
            sphere->pos     = ChVector3d(0, 0, z);
            sphere->pos_dt  = ChVector3d(0, 0, v);
            const ChVector3d F = sphere->forces[0]->Evaluate(system.time);  // three GetVal callbacks
            const ChVector3d T = sphere->forces[1]->Evaluate(system.time);  // three more, same cached evaluation
            (void)T;
            const double Fz = F.z() - mass * g;
            v += dt * Fz / (mass + a33);
            z += dt * v;
            std::printf("%.9f %.12f\n", system.time, z);
        }
        // R += c * M * w through the load against the Jacobian block
        ChVectorDynamic<> w(12), R(12);
        for (int i = 0; i < 12; ++i) {
            w(i) = 0.1 * (i + 1);
            R(i) = 1.0;
        }
        load->LoadIntLoadResidual_Mv(R, w, 0.5);
        double worst = 0.0;
        for (int i = 0; i < 12; ++i) {
            double ref = 1.0;
            for (int j = 0; j < 12; ++j) ref += 0.5 * M(i, j) * w(j);
            worst = std::fmax(worst, std::fabs(R(i) - ref) / std::fmax(1.0, std::fabs(ref)));
        }
        std::printf("MV_CHECK %.3e\n", worst);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
