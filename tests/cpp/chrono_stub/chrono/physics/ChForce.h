// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/functions/ChFunction.h"
namespace chrono {
class ChForce {
  public:
    enum class AlignmentFrame { BODY_DIR, WORLD_DIR };
    enum class ForceType { FORCE, TORQUE };
    void SetAlign(AlignmentFrame a) { align = a; }
    void SetMode(ForceType m) { mode = m; }
    void SetName(const std::string& n) { name = n; }
    void SetF_x(std::shared_ptr<ChFunction> f) { fx = std::move(f); }
    void SetF_y(std::shared_ptr<ChFunction> f) { fy = std::move(f); }
    void SetF_z(std::shared_ptr<ChFunction> f) { fz = std::move(f); }
    // what ChForce::UpdateTime does with its three modulation functions
    ChVector3d Evaluate(double t) const { return ChVector3d(fx->GetVal(t), fy->GetVal(t), fz->GetVal(t)); }
    AlignmentFrame align = AlignmentFrame::BODY_DIR;
    ForceType mode       = ForceType::FORCE;
    std::string name;
    std::shared_ptr<ChFunction> fx, fy, fz;
};
}  // namespace chrono
