// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/core/ChStubTypes.h"
namespace chrono {
class ChLoadable {
  public:
    virtual ~ChLoadable() = default;
};
struct ChLoadJacobians {
    ChMatrixDynamic<double> K, R, M;
};
class ChLoadBase {
  public:
    virtual ~ChLoadBase() = default;
    virtual void ComputeQ(ChState*, ChStateDelta*) = 0;
    virtual void ComputeJacobian(ChState*, ChStateDelta*) = 0;
    virtual void LoadIntLoadResidual_Mv(ChVectorDynamic<>& R, const ChVectorDynamic<>& w, const double c) = 0;
    virtual bool IsStiff() = 0;
    // ChLoadBase::Update -> CreateJacobianMatrices + ComputeJacobian in Chrono
    void StubUpdate(long n) {
        if (!m_jacobians) {
            m_jacobians = new ChLoadJacobians;
            m_jacobians->K.setZero(n, n);
            m_jacobians->R.setZero(n, n);
            m_jacobians->M.setZero(n, n);
        }
        ComputeJacobian(nullptr, nullptr);
    }
    ChLoadJacobians* m_jacobians = nullptr;
};
class ChLoadCustomMultiple : public ChLoadBase {
  public:
    explicit ChLoadCustomMultiple(std::vector<std::shared_ptr<ChLoadable>>& loadables) : loadables_(loadables) {}
    virtual ChLoadCustomMultiple* Clone() const = 0;
    std::vector<std::shared_ptr<ChLoadable>> loadables_;
};
}  // namespace chrono
