// Source compatibility of the plugin surface (include/hydroc_amd/*.h) with programs written against the reference: the hydro
// lines below are the reference's own, character for character --
//   decay / regular : tests/regression/sphere/demo_sphere_decay.cpp:52-120, demos/sphere/demo_sphere_reg_waves.cpp:126-151
//                     ("bodies.push_back(sphereBody); TestHydro hydro_forces(bodies, h5fname); hydro_forces.AddWaves(...);
//                      ... system.DoStepDynamics(timestep);")
//   yaml            : src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp:440-457
//                     ("hydro_data = ReadHydroYAML(...); ... test_hydro = SetupHydroFromYAML(hydro_data, bodies, loop_dt,
//                      sim_duration_hint, 0.0);")
// -- with only the include and the namespace changed.  Chrono itself is the stand-in of tests/cpp/chrono_stub (its DoStepDynamics
// is the heave-only form of Chrono's default stepper), so what is pinned is the surface and, through the goldens the Python side
// compares the output with, the whole chain ChForce -> ComponentFunc -> ForceFunc6d -> TestHydro -> GPU and ChLoadAddedMass.
//   usage: chrono_dropin_test decay   <sphere.h5> <nsteps>
//          chrono_dropin_test regular <sphere.h5> <nsteps> <amplitude> <omega> <pto damping>
//          chrono_dropin_test yaml    <case.hydro.yaml> <nsteps> <z0> <pto damping> [<device>,<device>...]
//          chrono_dropin_test api     <sphere.h5> 0
//          chrono_dropin_test addedmass <file.h5> <num_bodies>       (tests/chloadaddedmass_t01.cpp:44-58: the load built from HydroData)
// Prints "t z" per step with 9 decimals, then "WIRED <forces on body1> <loads> <system matrix rows>".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#define HYDROCHRONO_AMD_WITH_CHRONO 1
#include <hydroc_amd/chloadaddedmass.h>       // reference: <hydroc/chloadaddedmass.h>
#include <hydroc_amd/h5fileinfo.h>            // reference: <hydroc/h5fileinfo.h>
#include <hydroc_amd/hydro_forces.h>          // reference: <hydroc/hydro_forces.h>
#include <hydroc_amd/hydro_yaml_parser.h>     // reference: "hydro_yaml_parser.h"
#include <hydroc_amd/setup_hydro_from_yaml.h> // reference: "setup_hydro_from_yaml.h"

using namespace chrono;
using namespace hydroc_amd;

static int report(ChSystem& system, const std::shared_ptr<ChBody>& sphereBody) {
    long rows = 0;
    size_t loads = 0;
    for (auto& c : system.containers) {
        loads += c->loads.size();
        for (auto& l : c->loads) rows = l->m_jacobians ? l->m_jacobians->M.rows() : 0;
    }
    std::printf("WIRED %zu %zu %ld\n", sphereBody->forces.size(), loads, rows);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    const int nsteps       = std::atoi(argv[3]);
    try {
        ChSystem system;
        system.SetGravitationalAcceleration(ChVector3d(0.0, 0.0, -9.81));
        double timestep = 0.015;
        std::shared_ptr<ChBody> sphereBody = chrono_types::make_shared<ChBody>();
        auto ground                        = chrono_types::make_shared<ChBody>();
        ground->SetName("ground");

        if (mode == "decay") {
            std::string h5fname = argv[2];
            sphereBody->SetName("body1");  // must set body name correctly! (must match .h5 file)
            sphereBody->SetPos(ChVector3d(0, 0, -1));
            sphereBody->SetMass(261.8e3);
            system.Add(sphereBody);
            system.Add(ground);

            auto default_dont_add_waves = std::make_shared<NoWave>(1);

            // attach hydrodynamic forces to body
            std::vector<std::shared_ptr<ChBody>> bodies;
            bodies.push_back(sphereBody);

            TestHydro hydro_forces(bodies, h5fname);
            hydro_forces.AddWaves(default_dont_add_waves);

            for (int n = 0; n < nsteps; ++n) {
                system.DoStepDynamics(timestep);
                std::printf("%.9f %.9f\n", system.GetChTime(), sphereBody->GetPos().z());
            }
            return report(system, sphereBody);
        }
        if (mode == "regular") {
            if (argc < 7) return 2;
            std::string h5fname = argv[2];
            system.Add(sphereBody);
            sphereBody->SetName("body1");  // must set body name correctly! (must match .h5 file)
            sphereBody->SetPos(ChVector3d(0, 0, -2));
            sphereBody->SetMass(261.8e3);
            system.Add(ground);
            sphereBody->heave_damping = std::atof(argv[6]);  // the demo's ChLinkTSDA damper

            auto my_hydro_inputs                     = std::make_shared<RegularWave>(1);
            my_hydro_inputs->regular_wave_amplitude_ = std::atof(argv[4]);
            my_hydro_inputs->regular_wave_omega_     = std::atof(argv[5]);

            std::vector<std::shared_ptr<ChBody>> bodies;
            bodies.push_back(sphereBody);
            TestHydro hydro_forces(bodies, h5fname);
            hydro_forces.AddWaves(my_hydro_inputs);

            for (int n = 0; n < nsteps; ++n) {
                system.DoStepDynamics(timestep);
                std::printf("%.9f %.9f\n", system.GetChTime(), sphereBody->GetPos().z());
            }
            return report(system, sphereBody);
        }
        if (mode == "yaml") {
            if (argc < 6) return 2;
            std::string hydro_file = argv[2];
            sphereBody->SetName("body1");
            sphereBody->SetPos(ChVector3d(0, 0, std::atof(argv[4])));
            sphereBody->SetMass(261.8e3);
            sphereBody->heave_damping = std::atof(argv[5]);
            system.Add(sphereBody);
            system.Add(ground);
            std::vector<int> device_ids;
            if (argc > 6) {
                std::stringstream list(argv[6]);
                for (std::string item; std::getline(list, item, ',');) device_ids.push_back(std::atoi(item.c_str()));
            }
            const double loop_dt = timestep, sim_duration_hint = 40.0;
            std::unique_ptr<TestHydro> test_hydro;
            YAMLHydroData hydro_data;

            hydro_data = ReadHydroYAML(hydro_file);
            // Get all bodies from the system
            std::vector<std::shared_ptr<chrono::ChBody>> bodies;
            for (auto& body : system.GetBodies()) {
                bodies.push_back(body);
            }
            if (device_ids.empty())
                test_hydro = SetupHydroFromYAML(hydro_data, bodies, loop_dt, sim_duration_hint, 0.0);
            else
                test_hydro = SetupHydroFromYAML(hydro_data, bodies, loop_dt, sim_duration_hint, 0.0, device_ids);

            for (int n = 0; n < nsteps; ++n) {
                system.DoStepDynamics(timestep);
                std::printf("%.9f %.9f\n", system.GetChTime(), sphereBody->GetPos().z());
            }
            // what the runner asks the object afterwards (run_hydrochrono_from_yaml.cpp:668-679)
            auto wave_ptr = test_hydro->GetWave();
            if (wave_ptr && wave_ptr->GetWaveMode() == WaveMode::irregular) {
                auto irreg = std::static_pointer_cast<IrregularWaves>(wave_ptr);
                std::printf("IRREG %zu %zu %zu %zu\n", irreg->GetFrequenciesHz().size(), irreg->GetSpectrum().size(),
                            irreg->GetFreeSurfaceTime().size(), irreg->GetFreeSurfaceElevation().size());
                // the visualisation lines of demos/sphere/demo_sphere_irreg_waves.cpp:144-153
                irreg->SetUpWaveMesh(hydro_file + ".fse_mesh.obj");
                std::printf("MESH %s %.1f\n", irreg->GetMeshFile().c_str(), irreg->GetWaveMeshVelocity()[0]);
            }
            std::printf("RIRF %.17g\n", test_hydro->GetRIRFval(2, 2, 1));
            return report(system, sphereBody);
        }
        if (mode == "addedmass") {
            std::string h5fname = argv[2];
            const size_t nBodies = static_cast<size_t>(nsteps);
            std::vector<std::shared_ptr<ChBody>> bodies;
            for (size_t b = 0; b < nBodies; ++b) {
                bodies.push_back(chrono_types::make_shared<ChBody>());
                bodies.back()->SetName("body" + std::to_string(b + 1));
            }

            HydroData infos = H5FileInfo(h5fname, static_cast<int>(nBodies)).ReadH5Data();

            std::shared_ptr<ChLoadAddedMass> my_loadbodyinertia;

            std::vector<std::shared_ptr<ChLoadable>> loadables;
            for (auto& b : bodies) loadables.push_back(b);

            ChSystem my_system;  // (ChSystemSMC in the reference's test)

            my_loadbodyinertia = chrono_types::make_shared<ChLoadAddedMass>(infos.GetBodyInfos(), loadables, &my_system);

            // ... and what Chrono then does with it, against the load a TestHydro over the same file creates for itself
            for (auto& b : bodies) my_system.Add(b);
            my_system.Add(ground);  // 6 more coordinates behind the hydro bodies
            const long n = my_system.GetNumCoordsVelLevel();
            my_loadbodyinertia->StubUpdate(n);
            std::shared_ptr<ChLoadAddedMass> copy(my_loadbodyinertia->Clone());  // (Chrono clones loads; the clone shares the context)
            copy->StubUpdate(n);
            TestHydro hydro_forces(bodies, h5fname, std::make_shared<NoWave>(static_cast<int>(nBodies)));
            auto theirs = my_system.containers.at(0)->loads.at(0);
            theirs->StubUpdate(n);
            ChVectorDynamic<> w(n), R1(n), R2(n), R3(n);
            for (long i = 0; i < n; ++i) {
                w(i)  = 0.3 * std::sin(1.0 + 0.7 * static_cast<double>(i));
                R1(i) = R2(i) = R3(i) = 1.0 - 0.01 * static_cast<double>(i);
            }
            my_loadbodyinertia->LoadIntLoadResidual_Mv(R1, w, -0.6);
            theirs->LoadIntLoadResidual_Mv(R2, w, -0.6);
            copy->LoadIntLoadResidual_Mv(R3, w, -0.6);
            int same_mv = 1, same_m = 1;
            for (long i = 0; i < n; ++i) same_mv = same_mv && R1(i) == R2(i) && R1(i) == R3(i);
            const auto& M1 = my_loadbodyinertia->m_jacobians->M;
            const auto& M2 = theirs->m_jacobians->M;
            for (long i = 0; i < n; ++i)
                for (long j = 0; j < n; ++j) same_m = same_m && M1(i, j) == M2(i, j);
            std::printf("ADDEDMASS %ld %d %d\n", static_cast<long>(M1.rows()), same_m, same_mv);
            const long D = 6 * static_cast<long>(nBodies);
            for (long i = 0; i < D; ++i) {
                std::printf("ROW");
                for (long j = 0; j < D; ++j) std::printf(" %.17g", M1(i, j));
                std::printf("\n");
            }
            std::printf("MV");
            for (long i = 0; i < n; ++i) std::printf(" %.17g", R1(i));
            std::printf("\nEnd\n");
            return 0;
        }
        if (mode == "api") {
            // the rest of the surface a reference program may touch: GetWave / GetForceAtTime, the Compute* entry points, GetProfileStats,
            // the pass schedule, and what a constructor that throws leaves behind
            std::string h5fname = argv[2];
            sphereBody->SetName("body1");
            sphereBody->SetPos(ChVector3d(0, 0, -2));
            system.Add(sphereBody);
            auto waves                     = std::make_shared<RegularWave>(1);
            waves->regular_wave_amplitude_ = 0.5;
            waves->regular_wave_omega_     = 1.1;
            std::vector<std::shared_ptr<ChBody>> bodies;
            bodies.push_back(sphereBody);
            {
                auto misnamed = chrono_types::make_shared<ChBody>();
                misnamed->SetName("floater");  // no "body<k>" name: the constructor throws (std::stoi, as in the reference) and must leave nothing behind
                system.Add(misnamed);
                bool threw = false;
                try {
                    std::vector<std::shared_ptr<ChBody>> bad;
                    bad.push_back(misnamed);
                    TestHydro never(bad, h5fname);
                } catch (const std::exception&) {
                    threw = true;
                }
                std::printf("THROWS %d %zu\n", threw ? 1 : 0, misnamed->forces.size());
            }
            TestHydro hydro_forces(bodies, h5fname, waves);
            hydro_forces.SetPassSchedule(-1);
            hydro_forces.SetPassSchedule(1, 2);
            hydro_forces.SetPassSchedule(0);
            system.time = 0.75;
            const std::vector<double> fw = hydro_forces.ComputeForceWaves(), fg = hydro_forces.GetWave()->GetForceAtTime(0.75);
            const std::vector<double> hs = hydro_forces.ComputeForceHydrostatics();
            const double total_z         = hydro_forces.CoordinateFuncForBody(1, 2);
            const std::vector<double> rad = hydro_forces.ComputeForceRadiationDampingConv();  // second evaluation at this time: the reference's duplicate-time error
            std::printf("UNREACHED %zu\n", rad.size());
            (void)fw; (void)fg; (void)hs; (void)total_z;
            return 3;
        }
    } catch (const std::runtime_error& e) {
        if (mode == "api") {
            std::printf("DUPLICATE %s\n", e.what());
            return 0;
        }
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 2;
}
