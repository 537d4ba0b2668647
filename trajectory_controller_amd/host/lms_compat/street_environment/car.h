// TEST-ONLY stand-in for street_environment/car.h (see ../README.md): CarCommand with named,
// prioritised states (src/trajectory_point_follower.cpp:37-62,114-125,227-286).
#pragma once
#include <map>
#include <string>
namespace street_environment {
class CarCommand {
public:
    enum class StateType { NOT_DEFINED, IDLE, DRIVING };
    struct State {
        StateType state = StateType::NOT_DEFINED;
        int priority = 0;
        std::string name;
        double steering_front = 0, steering_rear = 0, targetSpeed = 0, targetDistance = 0;
        bool indicatorLeft = false, indicatorRight = false;
    };
    State* getState(const std::string& name) { auto it = states_.find(name); return it == states_.end() ? nullptr : &it->second; }
    void putState(const State& s) { states_[s.name] = s; }
    void removeState(const std::string& name) { states_.erase(name); }
    // measured quantities the controller reads
    float velocity() const { return velocity_; }
    void setVelocity(float v) { velocity_ = v; }
    float steeringFront() const { return 0; }
    float steeringRear() const { return 0; }
private:
    std::map<std::string, State> states_;
    float velocity_ = 0;
};
}  // namespace street_environment
