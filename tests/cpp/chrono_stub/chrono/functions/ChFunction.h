// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/core/ChStubTypes.h"
namespace chrono {
class ChFunction {
  public:
    virtual ~ChFunction() = default;
    virtual ChFunction* Clone() const = 0;
    virtual double GetVal(double x) const = 0;
};
}  // namespace chrono
