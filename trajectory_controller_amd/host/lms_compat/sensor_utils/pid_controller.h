// TEST-ONLY stand-in for sensor_utils/pid_controller.h (see ../README.md); interface from
// src/trajectory_point_follower.cpp:219-220,296-298,453,470.  Not pinned by any reference test:
// THIS build defines pid(e) as the textbook discrete PID on error e with step dt.
#pragma once
namespace sensor_utils {
class PID {
public:
    void set(float kp, float ki, float kd, float dt) { kp_ = kp; ki_ = ki; kd_ = kd; dt_ = dt; }
    float pid(float e) {
        integ_ += e * dt_;
        const float d = has_prev_ ? (e - prev_) / dt_ : 0.0f;
        prev_ = e; has_prev_ = true;
        return kp_ * e + ki_ * integ_ + kd_ * d;
    }
    void reset() { integ_ = 0; prev_ = 0; has_prev_ = false; }
private:
    float kp_ = 1, ki_ = 0, kd_ = 0, dt_ = 0.01f, integ_ = 0, prev_ = 0;
    bool has_prev_ = false;
};
}  // namespace sensor_utils
