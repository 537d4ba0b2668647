// TEST-ONLY stand-in for street_environment/trajectory.h (see ../README.md): usage at
// src/trajectory_point_follower.cpp:76,84-85,116-117,136,232-235,394-441.
#pragma once
#include <vector>
#include "lms/math/vertex.h"
namespace street_environment {
struct TrajectoryPoint {
    lms::math::vertex2f position, directory;
    float velocity = 0;
    bool right = true;
    bool isRight() const { return right; }
};
class Trajectory : public std::vector<TrajectoryPoint> {
public:
    using std::vector<TrajectoryPoint>::vector;
};
}  // namespace street_environment
