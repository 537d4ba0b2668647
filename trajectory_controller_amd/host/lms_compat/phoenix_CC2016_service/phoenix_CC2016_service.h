// TEST-ONLY stand-in for phoenix_CC2016_service (see ../README.md): driveMode() only
// (src/trajectory_point_follower.cpp:35-36,70).
#pragma once
namespace phoenix_CC2016_service {
enum class CCDriveMode { IDLE, FOH, FMH, PARKING };
class Phoenix_CC2016Service {
public:
    CCDriveMode driveMode() const { return mode; }
    CCDriveMode mode = CCDriveMode::FOH;
};
}  // namespace phoenix_CC2016_service
