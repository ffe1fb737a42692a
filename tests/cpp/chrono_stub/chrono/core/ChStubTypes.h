// TEST INFRASTRUCTURE ONLY -- never shipped, never part of the product.
// Minimal stand-ins for the few Project Chrono types the adapter block of include/hydroc_amd/hydro_forces.h touches
// (the classes the reference subclasses / calls: include/hydroc/hydro_forces.h:18-33,45-148, include/hydroc/chloadaddedmass.h:
// 22-90, src/hydro_forces.cpp:96-101,146-168,223-234, src/chloadaddedmass.cpp:27-70).  Project Chrono is not installed in the
// build image; these headers exist so that the guarded adapter code is compiled and driven by a test instead of rotting.
// They pin nothing about Chrono's behaviour.
#pragma once
#include <cmath>
#include <cstddef>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace chrono_types {
template <class T, class... A>
std::shared_ptr<T> make_shared(A&&... a) {
    return std::make_shared<T>(std::forward<A>(a)...);
}
}  // namespace chrono_types

namespace chrono {

struct ChVector3d {  // ref: src/hydro_forces.cpp:279
    double v[3] = {0, 0, 0};  // stub-only state
    ChVector3d() = default;
    ChVector3d(double a, double b, double c) : v{a, b, c} {}  // ref: demos/sphere/demo_sphere_reg_waves.cpp:57 (driver)
    double x() const { return v[0]; }  // ref: src/hydro_forces.cpp:284
    double y() const { return v[1]; }  // ref: src/hydro_forces.cpp:285
    double z() const { return v[2]; }  // ref: src/hydro_forces.cpp:286
    double Length() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }  // ref: src/hydro_forces.cpp:269
};

struct ChQuaterniond {  // (the type of ChBody::GetRot(), src/hydro_forces.cpp:280)
    ChVector3d cardan;  // stub-only state: the stub stores the angles directly
    ChVector3d GetCardanAnglesXYZ() const { return cardan; }  // ref: src/hydro_forces.cpp:280
};

// just enough of an Eigen-like dynamic matrix / vector for ChLoadAddedMass (in Chrono both are Eigen::Matrix types: the members
// below are Eigen's API, used by the reference at src/chloadaddedmass.cpp:16-20,37,48 -- setZero, block, rows, size -- and, for
// element access and raw storage, by the binding: operator(), data(), cols())
template <class T = double>
class ChMatrixDynamic {  // ref: include/hydroc/chloadaddedmass.h:86
  public:
    struct Block {
        ChMatrixDynamic& m;
        long r0, c0, nr, nc;
        Block& operator=(const ChMatrixDynamic& src) {
            for (long i = 0; i < nr; ++i)
                for (long j = 0; j < nc; ++j) m(r0 + i, c0 + j) = src(i, j);
            return *this;
        }
    };
    long rows() const { return r_; }
    long cols() const { return c_; }
    void setZero() { std::fill(d_.begin(), d_.end(), T(0)); }
    void setZero(long r, long c) {
        r_ = r;
        c_ = c;
        d_.assign(static_cast<size_t>(r) * c, T(0));
    }
    T& operator()(long i, long j) { return d_[static_cast<size_t>(i) * c_ + j]; }
    const T& operator()(long i, long j) const { return d_[static_cast<size_t>(i) * c_ + j]; }
    Block block(long r0, long c0, long nr, long nc) { return Block{*this, r0, c0, nr, nc}; }

  private:
    long r_ = 0, c_ = 0;
    std::vector<T> d_;
};

template <class T = double>
class ChVectorDynamic {  // ref: include/hydroc/chloadaddedmass.h:82
  public:
    ChVectorDynamic() = default;
    explicit ChVectorDynamic(long n) : d_(n, T(0)) {}
    long size() const { return static_cast<long>(d_.size()); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    T& operator()(long i) { return d_[i]; }
    const T& operator()(long i) const { return d_[i]; }

  private:
    std::vector<T> d_;
};

class ChState {};       // ref: include/hydroc/chloadaddedmass.h:54
class ChStateDelta {};  // ref: include/hydroc/chloadaddedmass.h:55

}  // namespace chrono
