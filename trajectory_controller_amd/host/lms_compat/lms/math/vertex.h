// TEST-ONLY stand-in for lms/math/vertex.h (see ../../README.md): float 2-vector with the members
// the reference uses (src/trajectory_point_follower.cpp:117,220,427,432-434,449).
#pragma once
#include <cmath>
namespace lms { namespace math {
struct vertex2f {
    float x = 0, y = 0;
    vertex2f() = default;
    vertex2f(float x_, float y_) : x(x_), y(y_) {}
    float length() const { return std::sqrt(x * x + y * y); }
    float distance(const vertex2f& o) const { return (*this - o).length(); }
    float angle() const { return std::atan2(y, x); }
    vertex2f normalize() const { const float l = length(); return l > 0 ? vertex2f(x / l, y / l) : vertex2f(0, 0); }
    vertex2f operator-(const vertex2f& o) const { return vertex2f(x - o.x, y - o.y); }
    vertex2f operator+(const vertex2f& o) const { return vertex2f(x + o.x, y + o.y); }
    vertex2f operator*(float s) const { return vertex2f(x * s, y * s); }
};
template <typename T> int sgn(T v) { return (T(0) < v) - (v < T(0)); }
}}  // namespace lms::math
