// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it.
#pragma once
#include "chrono/physics/ChLoadContainer.h"
namespace chrono {
class ChBody;
class ChSystem {  // ref: include/hydroc/chloadaddedmass.h:38
  public:
    void SetGravitationalAcceleration(const ChVector3d& g) { g_ = g; }     // ref: demos/sphere/demo_sphere_reg_waves.cpp:57 (driver)
    ChVector3d GetGravitationalAcceleration() const { return g_; }         // ref: src/hydro_forces.cpp:268
    long GetNumCoordsVelLevel() const { return ncoords_vel; }              // ref: src/chloadaddedmass.cpp:35
    void Add(std::shared_ptr<ChLoadContainer> c) { containers.push_back(std::move(c)); }  // ref: src/hydro_forces.cpp:233
    void AddBody(std::shared_ptr<ChBody> b);                               // ref: demos/sphere/demo_sphere_reg_waves.cpp:72 (driver)
    void Add(std::shared_ptr<ChBody> b) { AddBody(std::move(b)); }         // ref: demos/sphere/demo_sphere_reg_waves.cpp:97 (driver)
    double GetChTime() const { return time; }                              // ref: demos/sphere/demo_sphere_reg_waves.cpp:141 (driver)
    const std::vector<std::shared_ptr<ChBody>>& GetBodies() const { return bodies; }  // ref: src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp:447 (driver)
    void DoStepDynamics(double h);  // ref: demos/sphere/demo_sphere_reg_waves.cpp:145 (driver); heave-only stand-in, defined in ChBody.h
    // stub-only state
    double time       = 0.0;
    long ncoords_vel  = 0;
    std::vector<std::shared_ptr<ChLoadContainer>> containers;
    std::vector<std::shared_ptr<ChBody>> bodies;

  private:
    ChVector3d g_{0, 0, -9.81};
};
}  // namespace chrono
