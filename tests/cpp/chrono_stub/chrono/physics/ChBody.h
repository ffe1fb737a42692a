// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it.
#pragma once
#include "chrono/physics/ChForce.h"
#include "chrono/physics/ChLoad.h"
#include "chrono/physics/ChSystem.h"
namespace chrono {
class ChBody : public ChLoadable {  // ref: include/hydroc/hydro_forces.h:104
  public:
    void SetName(const std::string& n) { name_ = n; }   // ref: demos/sphere/demo_sphere_reg_waves.cpp:98 (driver)
    void SetPos(const ChVector3d& p) { pos = p; }       // ref: demos/sphere/demo_sphere_reg_waves.cpp:73 (driver)
    void SetMass(double m) { mass = m; }                // ref: demos/sphere/demo_sphere_reg_waves.cpp:100 (driver)
    const std::string& GetName() const { return name_; }                 // ref: src/hydro_forces.cpp:106
    double GetChTime() const { return system_ ? system_->time : 0.0; }   // ref: src/hydro_forces.cpp:550
    ChVector3d GetPos() const { return pos; }                            // ref: src/hydro_forces.cpp:279
    ChQuaterniond GetRot() const { return rot; }                         // ref: src/hydro_forces.cpp:280
    ChVector3d GetPosDt() const { return pos_dt; }                       // ref: src/hydro_forces.cpp:567
    ChVector3d GetAngVelParent() const { return angvel; }                // ref: src/hydro_forces.cpp:568
    void AddForce(std::shared_ptr<ChForce> f) { forces.push_back(std::move(f)); }  // ref: src/hydro_forces.cpp:166
    ChSystem* GetSystem() const { return system_; }                      // ref: src/hydro_forces.cpp:231
    // stub-only state (set and read by the tests)
    ChVector3d pos, pos_dt, angvel;
    ChQuaterniond rot;
    std::vector<std::shared_ptr<ChForce>> forces;
    ChSystem* system_ = nullptr;
    double mass       = 1.0;
    double heave_damping = 0.0;  // stands in for a ChLinkTSDA damper between the body and the ground (the sphere demos' PTO)

  private:
    std::string name_;
};
inline void ChSystem::AddBody(std::shared_ptr<ChBody> b) {
    b->system_ = this;
    ncoords_vel += 6;
    bodies.push_back(std::move(b));
}
// What Chrono's default stepper (Euler implicit linearized) does for bodies held on a vertical prismatic joint, which is all
// the reference's sphere regression drivers need: the applied loads are evaluated at (q_n, v_n, t_n) -- every
// ChForce of every hydro body, i.e. six ChFunction::GetVal callbacks per body --, the added-mass load enters the mass matrix
// through its Jacobian block, then v_{n+1} = v_n + h F / (m + M_zz), z_{n+1} = z_n + h v_{n+1}, t_{n+1} = t_n + h.
inline void ChSystem::DoStepDynamics(double h) {
    for (auto& c : containers)
        for (auto& l : c->loads) l->StubUpdate(GetNumCoordsVelLevel());
    std::vector<double> fz(bodies.size(), 0.0);
    for (size_t k = 0; k < bodies.size(); ++k)
        for (auto& f : bodies[k]->forces) {
            const ChVector3d v = f->Evaluate(time);
            if (f->mode == ChForce::ForceType::FORCE) fz[k] += v.z();
        }
    for (size_t k = 0; k < bodies.size(); ++k) {
        ChBody& b = *bodies[k];
        if (b.forces.empty()) continue;  // ground and other bodies without applied forces stay where they are
        double m = b.mass;
        for (auto& c : containers)
            for (auto& l : c->loads) m += l->m_jacobians->M(6 * static_cast<long>(k) + 2, 6 * static_cast<long>(k) + 2);
        const double F = fz[k] + b.mass * g_.z() - b.heave_damping * b.pos_dt.z();
        const double v = b.pos_dt.z() + h * F / m;
        b.pos_dt       = ChVector3d(0, 0, v);
        b.pos          = ChVector3d(b.pos.x(), b.pos.y(), b.pos.z() + h * v);
    }
    time += h;
}
}  // namespace chrono
