// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it.
#pragma once
#include "chrono/functions/ChFunction.h"
namespace chrono {
class ChForce {  // ref: include/hydroc/hydro_forces.h:145
  public:
    enum class AlignmentFrame { BODY_DIR, WORLD_DIR };  // ref: src/hydro_forces.cpp:98 (ChForce::AlignmentFrame::WORLD_DIR)
    enum class ForceType { FORCE, TORQUE };             // ref: src/hydro_forces.cpp:162 (ChForce::ForceType::TORQUE)
    void SetAlign(AlignmentFrame a) { align = a; }                     // ref: src/hydro_forces.cpp:98
    void SetMode(ForceType m) { mode = m; }                            // ref: src/hydro_forces.cpp:162
    void SetName(const std::string& n) { name = n; }                   // ref: src/hydro_forces.cpp:100
    void SetF_x(std::shared_ptr<ChFunction> f) { fx = std::move(f); }  // ref: src/hydro_forces.cpp:150
    void SetF_y(std::shared_ptr<ChFunction> f) { fy = std::move(f); }  // ref: src/hydro_forces.cpp:151
    void SetF_z(std::shared_ptr<ChFunction> f) { fz = std::move(f); }  // ref: src/hydro_forces.cpp:152
    // stub-only: what ChForce::UpdateTime does with its three modulation functions (the tests' stand-in for Chrono's update pass)
    ChVector3d Evaluate(double t) const { return ChVector3d(fx->GetVal(t), fy->GetVal(t), fz->GetVal(t)); }
    // stub-only state (read by the tests)
    AlignmentFrame align = AlignmentFrame::BODY_DIR;
    ForceType mode       = ForceType::FORCE;
    std::string name;
    std::shared_ptr<ChFunction> fx, fy, fz;
};
}  // namespace chrono
