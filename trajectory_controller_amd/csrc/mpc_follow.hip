// Batched front and back ends of the tobiMPC branch of TrajectoryPointController::cycle()
// (reference: src/trajectory_point_follower.cpp:76-117 and :277-283), so a host can hand raw
// TRAJECTORY points to the GPU and get CarCommand fields back:
//
//   traj_point_kernel   getTrajectoryPoint (:392-443): walk the polyline accumulating float segment
//                       lengths until the look-ahead distance is passed, step back along the last
//                       segment; then target extraction (:78-85): v clamp, the velocity lookup table
//                       (:323), y_soll = position.y, phi_soll = atan2(dir.y, dir.x).
//   traj_horizon_kernel the same walk continued to one trajectory point per horizon step
//                       (dlib::mpc::set_target(val, time), mpc.h:142-155), emitted as a general-form batch.
//   follow_post_kernel  the crossing rule (:277-283): targetSpeed < 0.5 -> steering 0.
//
// One lane per instance, SoA trajectories ([point][instance]) so every load of a wavefront is 64
// consecutive floats.  HBM-bound: 20 B per polyline point read once, 24 B written; no reuse, no LDS.
// The crossing-stop velocity PID (:445-473) is stateful and stays on the host.
// float arithmetic follows the module shim (host/trajectory_point_controller.cpp), which defines
// this build's semantics of the unvendored lms_math helpers ("parity unpinned").
#include "mpc_internal.h"

namespace tpc {

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

// One trajectory point per horizon step: the same walk, continued past the first look-ahead distance
// to look_ahead + t * spacing for t = 0 .. H-1 (the distances grow with t, so one pass over the
// polyline serves them all; a segment may hold several targets, and targets beyond the end of the
// polyline take its last point like getTrajectoryPoint's fall-through, :439-442).  Step t's point
// feeds dlib::mpc::set_target(val, t) (mpc.h:142-155) as (y_soll, phi_soll) = (position.y,
// atan2(directory)).  Step 0 is exactly traj_point_kernel's point, so target_speed / target_distance
// and the crossing rule do not change.  The model is the compact one written out in general form.
__global__ void traj_horizon_kernel(FollowArgs a, FollowHorizonArgs f) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int cnt = a.count[k] < a.max_points ? a.count[k] : a.max_points;
    const float first = a.look_ahead[k];
    double v = (double)a.car_velocity[k];
    if (fabs(v) < 0.1) v = 0.1;                                     // :78-82
    v = (double)lut_search(a.lut_x, a.lut_y, a.lut_n, (float)v);    // :323 (float table)
    const float spacing = f.step_spacing ? f.step_spacing[k] : (float)(fabs(v) * f.step);
    const int64_t ld = f.ldw;
    auto emit = [&](int t, float ox, float oy, float odx, float ody, float ovel) {
        const double y = (double)oy, phi = atan2((double)ody, (double)odx);   // :85, :84
        f.targets[(int64_t)(2 * t) * ld + k] = y;
        f.targets[(int64_t)(2 * t + 1) * ld + k] = phi;
        if (f.targets_copy) {
            f.targets_copy[(int64_t)(2 * t) * f.ld_copy + k] = y;
            f.targets_copy[(int64_t)(2 * t + 1) * f.ld_copy + k] = phi;
        }
        if (t == 0) {
            a.target_speed[k] = ovel;                                // :116
            a.target_distance[k] = sqrtf(ox * ox + oy * oy);         // :117
        }
    };
    int t = 0;
    if (cnt > 0) {
        float walked = 0.0f;
        float bx = a.px[k], by = a.py[k];
        for (int i = 1; i < cnt && t < f.H; ++i) {
            const int64_t o = (int64_t)i * a.ld + k;
            const float tx = a.px[o], ty = a.py[o];
            const float ex = bx - tx, ey = by - ty;
            const float len = sqrtf(ex * ex + ey * ey);
            walked += len;
            float want = t == 0 ? first : first + (float)t * spacing;
            while (t < f.H && walked > want) {
                const float back = walked - want;
                const float nx = len > 0.0f ? ex / len : 0.0f, ny = len > 0.0f ? ey / len : 0.0f;
                emit(t, tx + nx * back, ty + ny * back, a.dx[o], a.dy[o], a.vel[o]);
                ++t;
                want = first + (float)t * spacing;
            }
            bx = tx; by = ty;
        }
        const int64_t o = (int64_t)(cnt - 1) * a.ld + k;
        for (; t < f.H; ++t) emit(t, a.px[o], a.py[o], a.dx[o], a.dy[o], a.vel[o]);
    } else {
        // nothing to follow: idle straight ahead (:394-407)
        for (; t < f.H; ++t) emit(t, first + (float)t * spacing, 0.0f, 1.0f, 0.0f, 0.0f);
    }
    // A=[1,Tv;0,1]  B=[0,Tv;Tv/l,-Tv/l]  C=0  (:326-333), Q, R (:359-363), bounds (:16-18), x0 = 0 (:377-378)
    const double av = f.step * v, cv = f.step * v / f.wheelbase;
    f.A[k] = 1.0; f.A[ld + k] = av; f.A[2 * ld + k] = 0.0; f.A[3 * ld + k] = 1.0;
    f.B[k] = 0.0; f.B[ld + k] = av; f.B[2 * ld + k] = cv; f.B[3 * ld + k] = -cv;
    for (int j = 0; j < 2; ++j) {
        f.C[j * ld + k] = 0.0;
        f.Q[j * ld + k] = f.q[j];
        f.R[j * ld + k] = f.r[j];
        f.lo_out[j * ld + k] = f.lo[j];
        f.hi_out[j * ld + k] = f.hi[j];
        f.x0[j * ld + k] = 0.0;
    }
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
hipError_t launch_traj_horizon(const FollowArgs& a, const FollowHorizonArgs& f, hipStream_t s) {
    const int block = 256;
    hipLaunchKernelGGL(traj_horizon_kernel, dim3((unsigned)((a.n + block - 1) / block)), dim3(block), 0, s, a, f);
    return hipGetLastError();
}
hipError_t launch_follow_post(int64_t n, const float* target_speed, double* front, double* rear, hipStream_t s) {
    const int block = 256;
    hipLaunchKernelGGL(follow_post_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, s, n,
                       target_speed, front, rear);
    return hipGetLastError();
}

}  // namespace tpc
