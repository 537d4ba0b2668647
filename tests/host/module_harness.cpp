// Test harness: loads the module shim the way an LMS runtime would (channels, service, config),
// ticks cycle() over a few synthetic trajectories and prints what the module wrote to the CAR
// channel, plus the (v, y_soll, phi_soll) it fed to mpcControllerTobi, as JSON lines.
// usage: module_harness [horizon]     exit code 3 = initialize() refused (no GPU)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "trajectory_point_controller.h"

int main(int argc, char** argv) {
    const int horizon = argc > 1 ? std::atoi(argv[1]) : 4;
    lms::ChannelStore channels;
    std::map<std::string, std::shared_ptr<void>> services;
    auto phx = std::make_shared<phoenix_CC2016_service::Phoenix_CC2016Service>();
    services["PHOENIX_SERVICE"] = phx;

    TrajectoryPointController mod;
    mod.attach(&channels, &services);
    mod.config().set("mpcHorizon", horizon);
    mod.config().set("trajectoryPointDistanceLookupX", "0,1,2");
    mod.config().set("trajectoryPointDistanceLookupY", "0.4,0.6,0.9");
    if (!mod.initialize()) {
        std::printf("{\"initialize\": false}\n");
        return 3;
    }
    auto traj = channels.get<street_environment::Trajectory>("TRAJECTORY");
    auto car = channels.get<street_environment::CarCommand>("CAR");

    const float speeds[] = {0.05f, 0.8f, 1.7f, 2.6f};
    for (int sc = 0; sc < 4; ++sc) {
        traj->clear();
        for (int i = 0; i < 12; ++i) {   // a gentle curve, 0.15 m between points
            street_environment::TrajectoryPoint p;
            const float s = 0.15f * i;
            p.position = lms::math::vertex2f(s, 0.05f * sc * s * s + 0.02f * sc);
            p.directory = lms::math::vertex2f(1.0f, 0.1f * sc * s).normalize();
            p.velocity = 1.0f + 0.2f * sc;
            p.right = true;
            traj->push_back(p);
        }
        car->setVelocity(speeds[sc]);
        phx->mode = sc == 3 ? phoenix_CC2016_service::CCDriveMode::FMH : phoenix_CC2016_service::CCDriveMode::FOH;
        const bool ok = mod.cycle();
        const street_environment::CarCommand::State* st = car->getState("DEFAULT");
        // what cycle() fed to the solver (recomputed the same way the module does)
        float lookAhead = sc == 3 ? (speeds[sc] <= 2 ? 0.6f + (speeds[sc] - 1) * 0.3f : 0.9f)
                                  : 0.6f + speeds[sc] * 0.1f;
        const street_environment::TrajectoryPoint tp = mod.getTrajectoryPoint(lookAhead);
        double v = speeds[sc];
        if (std::fabs(v) < 0.1) v = 0.1;
        v = (double)(float)v;   // the velocity lookup table is a float table (follower.h:33, follower.cpp:323)
        std::printf("{\"points\": %d, \"look_ahead\": %.9g, \"car_velocity\": %.9g, \"px\": [", sc, (double)lookAhead, (double)speeds[sc]);
        for (size_t i = 0; i < traj->size(); ++i) std::printf("%s%.9g", i ? "," : "", (double)(*traj)[i].position.x);
        std::printf("], \"py\": [");
        for (size_t i = 0; i < traj->size(); ++i) std::printf("%s%.9g", i ? "," : "", (double)(*traj)[i].position.y);
        std::printf("], \"dx\": [");
        for (size_t i = 0; i < traj->size(); ++i) std::printf("%s%.9g", i ? "," : "", (double)(*traj)[i].directory.x);
        std::printf("], \"dy\": [");
        for (size_t i = 0; i < traj->size(); ++i) std::printf("%s%.9g", i ? "," : "", (double)(*traj)[i].directory.y);
        std::printf("], \"vel\": [");
        for (size_t i = 0; i < traj->size(); ++i) std::printf("%s%.9g", i ? "," : "", (double)(*traj)[i].velocity);
        std::printf("], \"target_distance\": %.9g}\n", (double)st->targetDistance);
        std::printf("{\"scenario\": %d, \"ok\": %s, \"v\": %.17g, \"y_soll\": %.17g, \"phi_soll\": %.17g, "
                    "\"steering_front\": %.17g, \"steering_rear\": %.17g, \"targetSpeed\": %.9g, \"driving\": %d}\n",
                    sc, ok ? "true" : "false", v, (double)tp.position.y,
                    std::atan2((double)tp.directory.y, (double)tp.directory.x), st->steering_front, st->steering_rear,
                    st->targetSpeed, st->state == street_environment::CarCommand::StateType::DRIVING);
    }
    // a lane change ahead sets the indicators (reference :227-241); they are reset every cycle
    (*traj)[8].right = false;
    mod.cycle();
    const street_environment::CarCommand::State* ind = car->getState("DEFAULT");
    std::printf("{\"indicator_left\": %s, \"indicator_right\": %s", ind->indicatorLeft ? "true" : "false",
                ind->indicatorRight ? "true" : "false");
    (*traj)[8].right = true;
    mod.cycle();
    ind = car->getState("DEFAULT");
    std::printf(", \"reset_left\": %s, \"reset_right\": %s}\n", ind->indicatorLeft ? "true" : "false",
                ind->indicatorRight ? "true" : "false");
    // IDLE drive mode publishes the priority-100 stand-still state and leaves DEFAULT alone; leaving the mode
    // removes it again (reference :35-52)
    phx->mode = phoenix_CC2016_service::CCDriveMode::IDLE;
    mod.cycle();
    const street_environment::CarCommand::State* idle = car->getState("IDLE");
    std::printf("{\"idle_state\": %s, \"priority\": %d, \"idle_speed\": %.9g", idle ? "true" : "false",
                idle ? idle->priority : -1, idle ? idle->targetSpeed : -1.0);
    phx->mode = phoenix_CC2016_service::CCDriveMode::FOH;
    mod.cycle();
    std::printf(", \"removed_after\": %s}\n", car->getState("IDLE") ? "false" : "true");
    // a back-end outside this build's scope is refused, not emulated
    mod.config().set("type", "PID");
    std::printf("{\"other_backend_refused\": %s}\n", mod.cycle() ? "false" : "true");
    mod.deinitialize();
    return 0;
}
