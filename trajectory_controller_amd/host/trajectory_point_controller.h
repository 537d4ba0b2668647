// TrajectoryPointController -- the LMS module surface of lms-org/trajectory_controller, kept so the
// MI355X solver drops into the existing C++ host (reference: include/trajectory_point_follower.h:26-70,
// src/interface.cpp:3).  Same class name, the same four lms::Module virtuals, the same data channels
// (TRAJECTORY in; TRAJECTORY_POINT, TRAJECTORY_DEBUG, CAR out) and service (PHOENIX_SERVICE).
//
// What changed: mpcControllerTobi no longer instantiates dlib::mpc on the CPU; its body is one call
// into the C ABI (include/tpc_mpc.h), which solves the box-constrained MPC QP on the GPU with dlib's
// iteration sequence.  dlib is not included anywhere.  MPC_HORIZON is a run-time config key
// ("mpcHorizon", default 4 = the reference's compile-time constant) because the library carries
// kernels for several horizons.
#pragma once

#include "lms/module.h"
#include "lms/math/vertex.h"
#include "lms/math/lookup_table.h"
#include "lms/math/interpolation.h"
#include "street_environment/car.h"
#include "street_environment/trajectory.h"
#include "sensor_utils/pid_controller.h"
#include "phoenix_CC2016_service/phoenix_CC2016_service.h"

#include "tpc_mpc.h"

class TrajectoryPointController : public lms::Module {
public:
    bool initialize() override;
    bool deinitialize() override;
    bool cycle() override;
    void configsChanged() override;

    // reference: include/trajectory_point_follower.h:44.  Public here so host-side tests can call it.
    void mpcControllerTobi(double v, double delta_y, double delta_phi, double* steering_front,
                           double* steering_rear);
    // reference: include/trajectory_point_follower.h:36
    street_environment::TrajectoryPoint getTrajectoryPoint(const float distanceToPoint);

    static constexpr size_t MPC_HORIZON = 4;   // reference default (include/...follower.h:48)

private:
    bool cycleTobiMpc(street_environment::CarCommand::State& state);
    void applyIndicatorsAndCrossing(street_environment::CarCommand::State& state);

    lms::math::LookupTable<float, lms::math::LookupTableOrder::ASC> m_mpcLookupVelocity;
    lms::math::LookupTable<float, lms::math::LookupTableOrder::ASC> m_trajectoryPointDistanceLookup;
    sensor_utils::PID slowDownCar;   // crossing-stop velocity rule inside getTrajectoryPoint (reference :445-473)

    double l = 0.21;   // wheelbase (reference: include/...follower.h:47)
    struct MpcParameters {
        double weight_y, weight_phi, weight_steeringFront, weight_steeringRear, stepSize;
    } mpcParameters{20, 7, 0.0005, 10, 0.1};
    double lower[2] = {0, 0}, upper[2] = {0, 0};

    lms::ReadDataChannel<street_environment::Trajectory> trajectory;
    lms::WriteDataChannel<street_environment::CarCommand> car;
    lms::WriteDataChannel<street_environment::TrajectoryPoint> debugging_trajectoryPoint;
    lms::WriteDataChannel<street_environment::Trajectory> trajectoryDebug;

    tpc_mpc_handle solver_ = nullptr;   // owns device scratch; created in initialize()
};
