// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it
// (tests/golden/chrono_usage.json, tests/test_chrono_stub_audit.py).
#pragma once
#include "chrono/core/ChStubTypes.h"
namespace chrono {
class ChFunction {  // ref: include/hydroc/hydro_forces.h:45 (ComponentFunc : public ChFunction)
  public:
    virtual ~ChFunction() = default;
    virtual ChFunction* Clone() const = 0;       // ref: include/hydroc/hydro_forces.h:73 (override)
    virtual double GetVal(double x) const = 0;   // ref: include/hydroc/hydro_forces.h:81 (override)
};
}  // namespace chrono
