// TEST-ONLY stand-in: the reference includes this header (include/trajectory_point_follower.h:10)
// but uses nothing from it on the tobiMPC path.
#pragma once
