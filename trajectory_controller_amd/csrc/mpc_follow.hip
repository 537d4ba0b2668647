// Batched front and back ends of the tobiMPC branch of TrajectoryPointController::cycle()
// (reference: src/trajectory_point_follower.cpp:76-117 and :277-283), so a host can hand raw
// TRAJECTORY points to the GPU and get CarCommand fields back:
//
//   traj_point_kernel   getTrajectoryPoint (:392-443): walk the polyline accumulating float segment
//                       lengths until the look-ahead distance is passed, step back along the last
//                       segment; then target extraction (:78-85): v clamp, the velocity lookup table
//                       (:323), y_soll = position.y, phi_soll = atan2(dir.y, dir.x).
//   follow_post_kernel  the crossing rule (:277-283): targetSpeed < 0.5 -> steering 0.
//
// One lane per instance, SoA trajectories ([point][instance]) so every load of a wavefront is 64
// consecutive floats.  HBM-bound: 20 B per polyline point read once, 24 B written; no reuse, no LDS.
// The crossing-stop velocity PID (:445-473) is stateful and stays on the host.
// float arithmetic follows the module shim (host/trajectory_point_controller.cpp), which defines
// this build's semantics of the unvendored lms_math helpers ("parity unpinned").
#include "mpc_internal.h"

namespace tpc {

struct FollowArgs {
    int64_t n, ld;
    int max_points;
    const float *px, *py, *dx, *dy, *vel;   // [max_points][ld]
    const int32_t* count;                   // [n]
    const float* car_velocity;              // [n]
    const float* look_ahead;                // [n]
    const float *lut_x, *lut_y;             // velocity lookup (ascending x), may be null
    int lut_n;
    double *v_out, *ysoll_out, *phisoll_out;   // [n] -> inputs of the compact solve
    float *target_speed, *target_distance;     // [n]
};

__device__ __forceinline__ float lut_search(const float* vx, const float* vy, int n, float x) {
    if (n <= 0) return x;
    if (x <= vx[0]) return vy[0];
    for (int i = 1; i < n; ++i)
        if (x <= vx[i]) {
            const float t = (x - vx[i - 1]) / (vx[i] - vx[i - 1]);
            return vy[i - 1] + t * (vy[i] - vy[i - 1]);
        }
    return vy[n - 1];
}

__global__ void traj_point_kernel(FollowArgs a) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int cnt = a.count[k] < a.max_points ? a.count[k] : a.max_points;
    const float want = a.look_ahead[k];
    // default when there is nothing to follow: idle straight ahead (:394-407)
    float ox = want, oy = 0.0f, odx = 1.0f, ody = 0.0f, ovel = 0.0f;
    if (cnt > 0) {
        float walked = 0.0f;
        bool found = false;
        float bx = a.px[k], by = a.py[k];
        for (int i = 1; i < cnt; ++i) {
            const int64_t o = (int64_t)i * a.ld + k;
            const float tx = a.px[o], ty = a.py[o];
            const float ex = bx - tx, ey = by - ty;                 // bot - top
            const float len = sqrtf(ex * ex + ey * ey);             // bot.distance(top)
            walked += len;
            if (walked > want) {
                const float back = walked - want;
                const float nx = len > 0.0f ? ex / len : 0.0f, ny = len > 0.0f ? ey / len : 0.0f;
                ox = tx + nx * back; oy = ty + ny * back;           // top + normalize(bot-top)*back
                odx = a.dx[o]; ody = a.dy[o]; ovel = a.vel[o];
                found = true;
                break;
            }
            bx = tx; by = ty;
        }
        if (!found) {                                               // :439-442 last point
            const int64_t o = (int64_t)(cnt - 1) * a.ld + k;
            ox = a.px[o]; oy = a.py[o]; odx = a.dx[o]; ody = a.dy[o]; ovel = a.vel[o];
        }
    }
    double v = (double)a.car_velocity[k];
    if (fabs(v) < 0.1) v = 0.1;                                     // :78-82
    v = (double)lut_search(a.lut_x, a.lut_y, a.lut_n, (float)v);    // :323 (float table)
    a.v_out[k] = v;
    a.ysoll_out[k] = (double)oy;                                    // :85
    a.phisoll_out[k] = atan2((double)ody, (double)odx);             // :84
    a.target_speed[k] = ovel;                                       // :116
    a.target_distance[k] = sqrtf(ox * ox + oy * oy);                // :117 position.length()
}

__global__ void follow_post_kernel(int64_t n, const float* target_speed, double* front, double* rear) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (target_speed[k] < 0.5f) { front[k] = 0.0; rear[k] = 0.0; }  // :277-283
}

hipError_t launch_traj_point(const FollowArgs& a, hipStream_t s) {
    const int block = 256;
    hipLaunchKernelGGL(traj_point_kernel, dim3((unsigned)((a.n + block - 1) / block)), dim3(block), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_follow_post(int64_t n, const float* target_speed, double* front, double* rear, hipStream_t s) {
    const int block = 256;
    hipLaunchKernelGGL(follow_post_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, s, n,
                       target_speed, front, rear);
    return hipGetLastError();
}

}  // namespace tpc
