// Multi-GPU inside one host process through the reference's plugin surface (drop-in variant): a ChronoHydroSystem whose
// TestHydro owns G body-row shard contexts (hc_create_sharded) and evaluates them with ONE hc_step_multi per Chrono time, read
// through ChForce -> ComponentFunc::GetVal like the reference's ForceFunc6d (src/hydro_forces.cpp:63-168), plus the added-mass
// load through ChLoadAddedMass::LoadIntLoadResidual_Mv (hc_added_mass_mv_multi).  Stand-in Chrono headers (tests/cpp/chrono_stub).
//   usage: shards_test <bemio.h5> <N> <states.bin> <nsteps> <dt> <n_shards> [regular|irregular|none]
//          shards_test <case.hydro.yaml> <N> <states.bin> <nsteps> <dt> <n_shards> yaml
// (yaml: the object comes from SetupHydroFromYAML(ReadHydroYAML(file), every body of the system, dt, 8.0, 0.5, devices) -- the
// runner's lines, src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp:440-457, with a device list)
// states.bin: [nsteps][12N] doubles = pos | rpy | linvel | angvel per step (written by the Python test, so that the oracle sees
// exactly the same inputs).  Prints per step the 6N totals with 17 significant digits, then "MV" and the 6N + 6 entries of R,
// then one "PROF" line per shard.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define HYDROCHRONO_AMD_WITH_CHRONO 1
#include "../../include/hydroc_amd/setup_hydro_from_yaml.h"

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const int N = std::atoi(argv[2]), nsteps = std::atoi(argv[4]), G = std::atoi(argv[6]);
    const double dt = std::atof(argv[5]);
    const std::string mode = argc > 7 ? argv[7] : "irregular";
    std::vector<double> states(static_cast<size_t>(nsteps) * 12 * N);
    {
        std::ifstream f(argv[3], std::ios::binary);
        if (!f.read(reinterpret_cast<char*>(states.data()), states.size() * sizeof(double))) return 2;
    }
    using namespace chrono;
    ChSystem system;
    system.SetGravitationalAcceleration(ChVector3d(0.3, -0.2, -9.7));  // tilted: every buoyancy-moment term is exercised
    std::vector<std::shared_ptr<ChBody>> bodies;
    for (int b = 0; b < N; ++b) {
        auto body = chrono_types::make_shared<ChBody>();
        body->SetName("body" + std::to_string(b + 1));
        system.AddBody(body);
        bodies.push_back(body);
    }
    auto extra = chrono_types::make_shared<ChBody>();  // a non-hydro body: the system has more coordinates than the added-mass block
    extra->SetName("ground");
    system.AddBody(extra);
    try {
        std::shared_ptr<hydroc_amd::WaveBase> waves;
        std::unique_ptr<hydroc_amd::TestHydro> from_yaml;
        std::unique_ptr<hydroc_amd::ChronoHydroSystem> from_h5;
        const std::vector<int> devices(static_cast<size_t>(G), 0);
        if (mode == "yaml") {
            std::vector<std::shared_ptr<ChBody>> all_bodies = system.bodies;  // the ground too: it is not in the YAML
            from_yaml = hydroc_amd::SetupHydroFromYAML(hydroc_amd::ReadHydroYAML(argv[1]), all_bodies, dt, 8.0, 0.5, devices);
        } else if (mode == "regular") {
            auto w = std::make_shared<hydroc_amd::RegularWave>(N);
            w->regular_wave_amplitude_ = 0.8;
            w->regular_wave_omega_     = 0.55;
            waves = w;
        } else if (mode == "none") {
            waves = std::make_shared<hydroc_amd::NoWave>(N);
        } else {
            hydroc_amd::IrregularWaveParams p;
            p.num_bodies_ = N;
            p.simulation_dt_ = dt;
            p.simulation_duration_ = 8.0;
            p.ramp_duration_ = 0.5;
            p.wave_height_ = 2.0;
            p.wave_period_ = 6.0;
            p.frequency_min_ = 0.05;
            p.frequency_max_ = 0.6;
            p.nfrequencies_ = 48;
            p.peak_enhancement_factor_ = 3.3;
            waves = std::make_shared<hydroc_amd::IrregularWaves>(p);
        }
        if (!from_yaml) from_h5 = std::make_unique<hydroc_amd::ChronoHydroSystem>(bodies, argv[1], waves, devices);
        hydroc_amd::TestHydro& hydro = from_yaml ? *from_yaml : from_h5->hydro();
        if (hydro.num_shards() != G) return 3;
        for (hc_ctx* c : hydro.contexts()) hydroc_amd::check(c, hc_enable_profiling(c, 1));
        auto load = system.containers[0]->loads[0];
        load->StubUpdate(system.GetNumCoordsVelLevel());
        for (int n = 0; n < nsteps; ++n) {
            const double* st = states.data() + static_cast<size_t>(n) * 12 * N;
            system.time = n * dt;
            for (int b = 0; b < N; ++b) {
                bodies[b]->pos        = ChVector3d(st[3 * b], st[3 * b + 1], st[3 * b + 2]);
                bodies[b]->rot.cardan = ChVector3d(st[3 * N + 3 * b], st[3 * N + 3 * b + 1], st[3 * N + 3 * b + 2]);
                bodies[b]->pos_dt     = ChVector3d(st[6 * N + 3 * b], st[6 * N + 3 * b + 1], st[6 * N + 3 * b + 2]);
                bodies[b]->angvel     = ChVector3d(st[9 * N + 3 * b], st[9 * N + 3 * b + 1], st[9 * N + 3 * b + 2]);
            }
            for (int b = 0; b < N; ++b) {  // six GetVal callbacks per body, one evaluation per time
                const ChVector3d F = bodies[b]->forces[0]->Evaluate(system.time), T = bodies[b]->forces[1]->Evaluate(system.time);
                std::printf("%.17g %.17g %.17g %.17g %.17g %.17g%c", F.x(), F.y(), F.z(), T.x(), T.y(), T.z(), b + 1 < N ? ' ' : '\n');
            }
        }
        const long n_sys = system.GetNumCoordsVelLevel();
        ChVectorDynamic<> w(n_sys), R(n_sys);
        for (long i = 0; i < n_sys; ++i) {
            w(i) = 0.1 * (i + 1) - 0.7;
            R(i) = 1.0 + 0.01 * i;
        }
        ChVectorDynamic<> R_host = R;
        auto added_mass = std::dynamic_pointer_cast<hydroc_amd::ChLoadAddedMass>(load);
        if (!added_mass) throw std::runtime_error("the registered load is not a ChLoadAddedMass");
        added_mass->SetHostProductLimit(0);  // the product of every shard on its GPU (hc_added_mass_mv_multi), whatever the size
        load->LoadIntLoadResidual_Mv(R, w, 0.5);
        std::printf("MV");
        for (long i = 0; i < n_sys; ++i) std::printf(" %.17g", R(i));
        std::printf("\n");
        added_mass->SetHostProductLimit(hydroc_amd::ChLoadAddedMass::kHostProductMaxDofs);  // the default: small systems multiply on the host copy
        load->LoadIntLoadResidual_Mv(R_host, w, 0.5);
        std::printf("MVHOST");
        for (long i = 0; i < n_sys; ++i) std::printf(" %.17g", R_host(i));
        std::printf("\n");
        // per shard: look-ahead passes, scatter launches, AQL dispatches, HIP launches (how the kernels reached the GPU)
        for (hc_ctx* c : hydro.contexts()) {
            hc_profile_stats p;
            hydroc_amd::check(c, hc_get_profile(c, &p));
            std::printf("PROF %lld %lld %lld %lld %d\n", p.block_kernel_launches, p.scatter_kernel_launches, p.direct_dispatches, p.hip_launches,
                        hc_direct_dispatch_active(c));
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
