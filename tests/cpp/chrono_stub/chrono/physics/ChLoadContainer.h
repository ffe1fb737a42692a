// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).  Every declaration cites the reference line that uses it.
#pragma once
#include "chrono/physics/ChLoad.h"
namespace chrono {
class ChLoadContainer {  // ref: src/hydro_forces.cpp:223
  public:
    void Add(std::shared_ptr<ChLoadBase> l) { loads.push_back(std::move(l)); }  // ref: src/hydro_forces.cpp:234
    std::vector<std::shared_ptr<ChLoadBase>> loads;  // stub-only state
};
}  // namespace chrono
