// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/physics/ChLoadContainer.h"
namespace chrono {
class ChBody;
class ChSystem {
  public:
    void SetGravitationalAcceleration(const ChVector3d& g) { g_ = g; }
    ChVector3d GetGravitationalAcceleration() const { return g_; }
    long GetNumCoordsVelLevel() const { return ncoords_vel; }
    void Add(std::shared_ptr<ChLoadContainer> c) { containers.push_back(std::move(c)); }
    void AddBody(std::shared_ptr<ChBody> b);
    void Add(std::shared_ptr<ChBody> b) { AddBody(std::move(b)); }
    double GetChTime() const { return time; }
    const std::vector<std::shared_ptr<ChBody>>& GetBodies() const { return bodies; }
    void DoStepDynamics(double h);  // heave-only stand-in, defined in ChBody.h
    double time       = 0.0;
    long ncoords_vel  = 0;
    std::vector<std::shared_ptr<ChLoadContainer>> containers;
    std::vector<std::shared_ptr<ChBody>> bodies;

  private:
    ChVector3d g_{0, 0, -9.81};
};
}  // namespace chrono
