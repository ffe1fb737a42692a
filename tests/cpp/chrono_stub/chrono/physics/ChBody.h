// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/physics/ChForce.h"
#include "chrono/physics/ChLoad.h"
#include "chrono/physics/ChSystem.h"
namespace chrono {
class ChBody : public ChLoadable {
  public:
    void SetName(const std::string& n) { name_ = n; }
    const std::string& GetName() const { return name_; }
    double GetChTime() const { return system_ ? system_->time : 0.0; }
    ChVector3d GetPos() const { return pos; }
    ChQuaterniond GetRot() const { return rot; }
    ChVector3d GetPosDt() const { return pos_dt; }
    ChVector3d GetAngVelParent() const { return angvel; }
    void AddForce(std::shared_ptr<ChForce> f) { forces.push_back(std::move(f)); }
    ChSystem* GetSystem() const { return system_; }
    ChVector3d pos, pos_dt, angvel;
    ChQuaterniond rot;
    std::vector<std::shared_ptr<ChForce>> forces;
    ChSystem* system_ = nullptr;

  private:
    std::string name_;
};
inline void ChSystem::AddBody(std::shared_ptr<ChBody> b) {
    b->system_ = this;
    ncoords_vel += 6;
    bodies.push_back(std::move(b));
}
}  // namespace chrono
