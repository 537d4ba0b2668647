// Plugin export, as in the reference (src/interface.cpp:3).
#include "trajectory_point_controller.h"
LMS_MODULE_INTERFACE(TrajectoryPointController)
