// TEST INFRASTRUCTURE ONLY (see chrono/core/ChStubTypes.h).
#pragma once
#include "chrono/physics/ChLoad.h"
namespace chrono {
class ChLoadContainer {
  public:
    void Add(std::shared_ptr<ChLoadBase> l) { loads.push_back(std::move(l)); }
    std::vector<std::shared_ptr<ChLoadBase>> loads;
};
}  // namespace chrono
