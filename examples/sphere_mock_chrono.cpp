// sphere_mock_chrono.cpp -- C++ driver of the GPU hydro-force path WITHOUT Project Chrono.
//
// Mirrors the reference's sphere regression drivers (tests/regression/sphere/demo_sphere_decay.cpp,
// .../reg_waves/sphere_reg_waves_test.cpp, .../irreg_waves/sphere_irreg_waves_test.cpp) through the C++ mirror
// (include/hydroc_amd/*.h): same TestHydro / wave-class calls, with a 1-DOF symplectic-Euler heave integrator in place
// of ChSystem::DoStepDynamics (force at (z_n, v_n, t_n), mass m + rho*Ainf_33).  Prints "t z" with 6 decimals like the
// reference's result files.
//   usage: sphere_mock_chrono <sphere.h5> decay|regular|irregular <nsteps>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../include/hydroc_amd/setup_hydro_from_yaml.h"

using namespace hydroc_amd;

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <sphere.h5> decay|regular|irregular <nsteps>\n", argv[0]);
        return 2;
    }
    const std::string h5 = argv[1], mode = argv[2];
    const int nsteps     = std::atoi(argv[3]);
    const double timestep = 0.015, mass = 261.8e3, g = 9.81;
    double pto_damping = 0.0;

    auto sphere = std::make_shared<MockBody>("body1");  // must match the .h5 body name
    std::vector<std::shared_ptr<BodyView>> bodies{sphere};
    try {
        std::shared_ptr<WaveBase> waves;
        if (mode == "decay") {
            sphere->pos = {0, 0, -1};
            waves       = std::make_shared<NoWave>(1);
        } else if (mode == "regular") {
            sphere->pos                   = {0, 0, -2};
            auto w                        = std::make_shared<RegularWave>(1);
            w->regular_wave_amplitude_    = 0.177;        // task10 case 1
            w->regular_wave_omega_        = 2.094395102;
            pto_damping                   = 398736.034;
            waves                         = w;
        } else {
            sphere->pos = {0, 0, -2};
            IrregularWaveParams p;
            p.num_bodies_          = 1;
            p.simulation_dt_       = timestep;
            p.simulation_duration_ = 600.0;
            p.ramp_duration_       = 60.0;
            p.wave_height_         = 2.0;
            p.wave_period_         = 12.0;
            p.frequency_min_       = 0.001;
            p.frequency_max_       = 1.0;
            p.nfrequencies_        = 1000;
            waves                  = std::make_shared<IrregularWaves>(p);
        }
        TestHydro hydro_forces(bodies, h5);
        hydro_forces.AddWaves(waves);
        hydro_forces.SetGravitationalAcceleration(0.0, 0.0, -g);
        const double a33 = hydro_forces.GetAddedMassMatrix()[2 * 6 + 2];
        double z = sphere->pos[2], v = 0.0;
        for (int n = 0; n < nsteps; ++n) {
            sphere->time      = n * timestep;
            sphere->pos[2]    = z;
            sphere->linvel[2] = v;
            double fz = 0.0;
            for (int i = 0; i < 6; ++i) {  // Chrono evaluates all six ComponentFunc; only the first one computes
                const double f = hydro_forces.CoordinateFuncForBody(1, i);
                if (i == 2) fz = f;
            }
            const double F = fz - mass * g - pto_damping * v;
            v += timestep * F / (mass + a33);
            z += timestep * v;
            std::printf("%.6f %.6f\n", (n + 1) * timestep, z);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
