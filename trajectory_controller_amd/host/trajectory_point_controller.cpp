// Module shim: TrajectoryPointController::cycle() with the tobiMPC back-end
// (reference: src/trajectory_point_follower.cpp:8-126, :227-299, :392-476) and the QP solved by
// libtpc_mpc.so.  Written from the reference's observable behaviour, without dlib.
// The whole cycle() of the reference is here -- drive-mode IDLE state (:35-52), default state (:54-61),
// the tobiMPC branch (:64-126), turn indicators (:227-241), crossing rule (:277-283), putState (:286) --
// except its two other back-ends: mikMPC (:127-213) calls call_andromeda() from a submodule that is not
// vendored, and the PID back-end (:214-225) is outside this build's scope; both
// are refused with an error, never emulated.
#include "trajectory_point_controller.h"

#include <cmath>

bool TrajectoryPointController::initialize() {
    // channels: reference src/trajectory_point_follower.cpp:9-13
    trajectory = readChannel<street_environment::Trajectory>("TRAJECTORY");
    debugging_trajectoryPoint = writeChannel<street_environment::TrajectoryPoint>("TRAJECTORY_POINT");
    trajectoryDebug = writeChannel<street_environment::Trajectory>("TRAJECTORY_DEBUG");
    car = writeChannel<street_environment::CarCommand>("CAR");

    // steering-angle limits +-22 degrees: reference :16-18
    const double alpha_max = 22 * M_PI / 180;
    lower[0] = lower[1] = -alpha_max;
    upper[0] = upper[1] = alpha_max;

    configsChanged();

    // the one new piece of state: a solver handle on the configured GPU.  No CPU fallback: without
    // the device the module refuses to initialise.
    const int device = config().get<int>("gpuDevice", 0);
    // the library next to this module must speak the header it was compiled against: ABI 4 changed the meaning of a field
    // of tpc_mpc_params (`reserved` became `options` and must be zero), later versions only add symbols
    if (tpc_mpc_abi_version() < 4 || tpc_mpc_abi_version() > TPC_MPC_ABI_VERSION) {
        logger.error("trajectory_point_controller") << "libtpc_mpc.so speaks ABI " << tpc_mpc_abi_version()
                                                     << ", this module was built for " << TPC_MPC_ABI_VERSION;
        return false;
    }
    if (tpc_mpc_create(device, &solver_) != TPC_MPC_OK) {
        logger.error("trajectory_point_controller") << "tpc_mpc_create failed: " << tpc_mpc_last_error(nullptr);
        return false;
    }
    return true;
}

bool TrajectoryPointController::deinitialize() {
    tpc_mpc_destroy(solver_);
    solver_ = nullptr;
    return true;
}

void TrajectoryPointController::configsChanged() {
    // reference :291-299
    m_mpcLookupVelocity.vx = config().getArray<float>("mpcLookupVelocityX");
    m_mpcLookupVelocity.vy = config().getArray<float>("mpcLookupVelocityY");
    m_trajectoryPointDistanceLookup.vx = config().getArray<float>("trajectoryPointDistanceLookupX");
    m_trajectoryPointDistanceLookup.vy = config().getArray<float>("trajectoryPointDistanceLookupY");
    const float dt = config().get<float>("dt", 0.01f);
    slowDownCar.set(config().get<float>("PID_Kp", 1), config().get<float>("PID_Ki", 0), config().get<float>("PID_Kd", 0), dt);
}

bool TrajectoryPointController::cycle() {
    using street_environment::CarCommand;
    auto phx = getService<phoenix_CC2016_service::Phoenix_CC2016Service>("PHOENIX_SERVICE");

    // drive mode IDLE: a priority-100 stand-still state, removed again once the mode is left (reference :35-52)
    if (phx->driveMode() == phoenix_CC2016_service::CCDriveMode::IDLE) {
        CarCommand::State idle;
        if (CarCommand::State* prev = car->getState("IDLE")) idle = *prev;
        idle.state = CarCommand::StateType::IDLE;
        idle.priority = 100;
        idle.name = "IDLE";
        idle.steering_front = idle.steering_rear = idle.targetSpeed = 0;
        car->putState(idle);
        return true;
    }
    car->removeState("IDLE");

    CarCommand::State state;
    if (CarCommand::State* prev = car->getState("DEFAULT")) state = *prev;   // reference :55-61
    state.priority = 10;
    state.name = "DEFAULT";

    const std::string type = config().get<std::string>("type", "tobiMPC");   // reference :64
    if (type != "tobiMPC") {
        // mikMPC (:127-213) calls call_andromeda() from a submodule that is not vendored; the PID
        // back-end (:214-225) is host glue outside this build's scope
        logger.error("trajectory_point_controller") << "type=" << type << ": only the tobiMPC back-end is part of this build";
        return false;
    }
    if (!cycleTobiMpc(state)) return false;

    applyIndicatorsAndCrossing(state);
    car->putState(state);   // reference :286
    return true;
}

// reference :65-125
bool TrajectoryPointController::cycleTobiMpc(street_environment::CarCommand::State& state) {
    using street_environment::CarCommand;
    auto phx = getService<phoenix_CC2016_service::Phoenix_CC2016Service>("PHOENIX_SERVICE");

    float lookAhead = m_trajectoryPointDistanceLookup.linearSearch(car->velocity());
    if (phx->driveMode() == phoenix_CC2016_service::CCDriveMode::FOH)
        lookAhead = config().get<float>("regelpunktMin", 0.6f) + car->velocity() * config().get<float>("regelpunktSlope", 0.1f);

    const street_environment::TrajectoryPoint tp = getTrajectoryPoint(lookAhead);

    double v = car->velocity();
    if (std::fabs(v) < 0.1) v = 0.1;   // the model divides nothing by v, but the reference clamps (:79-82)

    // :84 calls unqualified atan2 on two floats into a double; whether that resolves to the float or
    // the double overload depends on which of <cmath>/<math.h> the (unvendored) LMS headers pull in.
    // This build takes the double overload of glibc's ::atan2 (the only one <cmath> alone exposes
    // globally); the two differ by ~1e-8 relative in phi_soll.
    const double phi_soll = std::atan2((double)tp.directory.y, (double)tp.directory.x);
    const double y_soll = tp.position.y;                                  // :85

    // weights are read every cycle so they can be tuned live (:91-96)
    mpcParameters.weight_y = config().get<double>("weight_y", 20);
    mpcParameters.weight_phi = config().get<double>("weight_phi", 7);
    mpcParameters.weight_steeringFront = config().get<double>("weight_steering_front", 0.0005);
    mpcParameters.weight_steeringRear = config().get<double>("weight_steering_rear", 10);
    mpcParameters.stepSize = 0.1;

    double steering_front = 0, steering_rear = 0;
    mpcControllerTobi(v, y_soll, phi_soll, &steering_front, &steering_rear);   // :97
    if (std::isnan(steering_front) || std::isnan(steering_rear) || std::isnan(tp.velocity))
        logger.error("trajectory_point_controller") << "invalid vals: " << steering_front << " " << steering_rear;

    state.steering_front = steering_front;              // :114-117
    state.steering_rear = steering_rear;
    state.targetSpeed = tp.velocity;
    state.targetDistance = tp.position.length();
    *debugging_trajectoryPoint = tp;                    // :120
    state.state = tp.velocity == 0 ? CarCommand::StateType::IDLE : CarCommand::StateType::DRIVING;   // :121-125
    return true;
}

// reference :227-241 and :277-283: the indicators are reset every cycle and set when the trajectory changes
// side ahead; below 0.5 m/s (probably standing at a crossing) no steering and no indicators
void TrajectoryPointController::applyIndicatorsAndCrossing(street_environment::CarCommand::State& state) {
    state.indicatorLeft = state.indicatorRight = false;
    if (!trajectory->empty()) {   // (the reference reads at(0) unguarded; an empty trajectory has no side to change)
        const bool startsRight = trajectory->at(0).isRight();
        for (const street_environment::TrajectoryPoint& p : *trajectory)
            if (p.isRight() != startsRight) {
                state.indicatorLeft = startsRight;
                state.indicatorRight = !startsRight;
                break;
            }
    }
    if (state.targetSpeed < 0.5) {
        state.indicatorLeft = state.indicatorRight = false;
        state.steering_front = state.steering_rear = 0;
    }
}

// reference :301-389.  The dlib controller of the reference is replaced by one C-ABI call; the model
// (A, B, C, Q, R, bounds, one target for all steps, x0 = 0, cold start) is built inside the library
// from exactly these arguments.
void TrajectoryPointController::mpcControllerTobi(double v, double delta_y, double delta_phi,
                                                  double* steering_front, double* steering_rear) {
    v = m_mpcLookupVelocity.linearSearch((float)v);   // reference :323 (float table, like lms_math)

    tpc_mpc_params p;
    tpc_mpc_default_params(&p, config().get<int>("mpcHorizon", (int)MPC_HORIZON));
    p.step_size = mpcParameters.stepSize;
    p.wheelbase = l;
    p.weight_y = mpcParameters.weight_y;
    p.weight_phi = mpcParameters.weight_phi;
    p.weight_steering_front = mpcParameters.weight_steeringFront;
    p.weight_steering_rear = mpcParameters.weight_steeringRear;
    for (int j = 0; j < 2; ++j) { p.lower[j] = lower[j]; p.upper[j] = upper[j]; }
    // the knobs the reference left commented out (:374-375) are live config keys here
    p.eps = config().get<double>("mpcEpsilon", p.eps);
    p.max_iter = (uint64_t)config().get<long>("mpcMaxIterations", (long)p.max_iter);

    const int rc = tpc_mpc_solve_one(solver_, &p, v, delta_y, delta_phi, steering_front, steering_rear);
    if (rc != TPC_MPC_OK) {
        logger.error("trajectory_point_controller") << "tpc_mpc_solve_one: " << tpc_mpc_last_error(solver_);
        *steering_front = *steering_rear = std::nan("");   // surfaces through the NaN check in cycle()
        return;
    }
    // dlib reports nothing; the reference only logs NaN outputs behind the call (:101-103).  A NaN / Inf input
    // comes back as the start point (0, 0) -- which that check cannot see -- and a solve cut off by
    // mpcMaxIterations as its last iterate: both are logged here, neither is fatal.
    uint32_t flags = 0;
    int32_t iters = 0;
    if (tpc_mpc_last_flags(solver_, &flags, &iters) == TPC_MPC_OK && flags != 0) {
        if (flags & TPC_MPC_FLAG_NONFINITE)
            logger.warn("mpcControllerTobi") << "non-finite input (v " << v << ", delta_y " << delta_y << ", delta_phi "
                                             << delta_phi << "): steering left at the start point";
        if (flags & TPC_MPC_FLAG_MAX_ITER)
            logger.warn("mpcControllerTobi") << "stopped by mpcMaxIterations after " << iters << " iterations";
    }
}

// reference :392-476
street_environment::TrajectoryPoint TrajectoryPointController::getTrajectoryPoint(const float distanceToPoint) {
    // default when nothing can be followed: idle straight ahead (:394-407)
    street_environment::TrajectoryPoint out;
    out.position = lms::math::vertex2f(distanceToPoint, 0);
    out.directory = lms::math::vertex2f(1, 0);
    out.velocity = 0;
    if (trajectory->size() == 0) {
        logger.warn("cycle") << "Can't follow anything";
        return out;
    }
    // walk the polyline until the accumulated arc length passes distanceToPoint, then step back
    // along the last segment (:423-438); the point inherits velocity and direction of the segment end
    float walked = 0;
    bool found = false;
    for (size_t i = 1; i < trajectory->size(); ++i) {
        const street_environment::TrajectoryPoint& bot = trajectory->at(i - 1);
        const street_environment::TrajectoryPoint& top = trajectory->at(i);
        walked += bot.position.distance(top.position);
        if (walked > distanceToPoint) {
            const float back = walked - distanceToPoint;
            out = top;
            out.position = top.position + (bot.position - top.position).normalize() * back;
            found = true;
            break;
        }
    }
    if (!found) out = trajectory->at(trajectory->size() - 1);   // :439-442

    // stop-at-crossing velocity rule (:445-473)
    float minVelocity = config().get<float>("maxVelocityCrossing", 1.0f);
    for (const street_environment::TrajectoryPoint& p : *trajectory) {
        if (p.velocity != 0) continue;
        const float distanceToStop = lms::math::sgn(p.position.x) * p.position.length() - config().get<float>("stoppingDistance", 0.35f);
        if (distanceToStop < config().get<float>("distanceToStop", 1)) {
            const float vmaxCrossing = config().get<float>("maxVelocityCrossing", 1.0f);
            float velocity = slowDownCar.pid(distanceToStop);
            if (std::isnan(velocity) || velocity >= vmaxCrossing) velocity = vmaxCrossing;
            if (distanceToStop <= config().get<float>("crossingSaftyZone", 0.05f) || velocity < 0) velocity = 0;
            if (velocity < minVelocity) minVelocity = velocity;
            out.velocity = minVelocity;
            if (minVelocity == 0) break;
        } else {
            slowDownCar.reset();
        }
    }
    return out;
}
