// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it.
#pragma once
#include "chrono/core/ChStubTypes.h"
namespace chrono {
class ChLoadable {  // ref: include/hydroc/chloadaddedmass.h:37
  public:
    virtual ~ChLoadable() = default;
};
struct ChLoadJacobians {  // ref: src/chloadaddedmass.cpp:44 (m_jacobians->M), :48 (->R), :52 (->K)
    ChMatrixDynamic<double> K, R, M;
};
class ChLoadBase {  // (base of ChLoadCustomMultiple in Chrono; the reference overrides its virtuals)
  public:
    virtual ~ChLoadBase() = default;
    virtual void ComputeQ(ChState*, ChStateDelta*) = 0;         // ref: include/hydroc/chloadaddedmass.h:54 (override)
    virtual void ComputeJacobian(ChState*, ChStateDelta*) = 0;  // ref: include/hydroc/chloadaddedmass.h:69 (override)
    virtual void LoadIntLoadResidual_Mv(ChVectorDynamic<>& R, const ChVectorDynamic<>& w, const double c) = 0;  // ref: include/hydroc/chloadaddedmass.h:82 (override)
    virtual bool IsStiff() = 0;                                 // ref: include/hydroc/chloadaddedmass.h:89 (override)
    // stub-only: ChLoadBase::Update -> CreateJacobianMatrices + ComputeJacobian in Chrono
    void StubUpdate(long n) {
        if (!m_jacobians) {
            m_jacobians = new ChLoadJacobians;
            m_jacobians->K.setZero(n, n);
            m_jacobians->R.setZero(n, n);
            m_jacobians->M.setZero(n, n);
        }
        ComputeJacobian(nullptr, nullptr);
    }
    ChLoadJacobians* m_jacobians = nullptr;  // ref: src/chloadaddedmass.cpp:44
};
class ChLoadCustomMultiple : public ChLoadBase {  // ref: include/hydroc/chloadaddedmass.h:22
  public:
    explicit ChLoadCustomMultiple(std::vector<std::shared_ptr<ChLoadable>>& loadables) : loadables_(loadables) {}  // ref: src/chloadaddedmass.cpp:15
    virtual ChLoadCustomMultiple* Clone() const = 0;  // ref: include/hydroc/chloadaddedmass.h:43 (override)
    std::vector<std::shared_ptr<ChLoadable>> loadables_;  // stub-only state
};
}  // namespace chrono
