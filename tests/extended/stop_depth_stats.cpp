// tests/extended/stop_depth_stats.cpp -- ANALYSIS TOOL, NOT PRODUCT CODE (and not a test).
//
// Question it answers: in the projected-gradient phase of the LANE_FMA arithmetic (mpc_ub_model.h), after how many steps
// of the backward sweep is the stop test of an iteration already DECIDED -- i.e. the running maximum of the stop-test terms
// has reached g eps, so that no later term can make the instance stop (mpc.h:310 needs EVERY term below eps)?  A kernel
// could then drop the three stop-test instructions of every remaining variable whenever all 64 lanes of the wavefront
// are decided.  Prints, per instance: iterations and the depth (steps visited, 1..H; H+1 = the stopping iteration) of
// every PG iteration, as bytes, for the wave-level simulation in stop_depth_sim.py.
//
//   g++ -O2 -mfma -ffp-contract=off -std=c++17 -o /tmp/stop_depth_stats tests/extended/stop_depth_stats.cpp
//   /tmp/stop_depth_stats inputs.bin n out.bin      (inputs: v[n] dy[n] dphi[n] doubles; H = 20, dlib's defaults)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../trajectory_controller_amd/csrc/mpc_ub_host.h"

using namespace tpc::ub;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const long n = atol(argv[2]);
    constexpr int H = 20;
    typedef double T;
    std::vector<double> in(3 * n);
    FILE* fi = fopen(argv[1], "rb");
    if (!fi || fread(in.data(), 8, 3 * n, fi) != (size_t)(3 * n)) return 3;
    fclose(fi);
    FILE* fo = fopen(argv[3], "wb");
    FILE* fx = getenv("STOP_DEPTH_FEATURES") ? fopen(getenv("STOP_DEPTH_FEATURES"), "wb") : nullptr;   // lambda, last CD max_df, PG iterations per instance
    const T q[2] = {1.0, 1.0}, r[2] = {1.0, 1.0};   // overwritten below from argv if given
    T qq[2] = {q[0], q[1]}, rr[2] = {r[0], r[1]};
    double step = 0.1, wb = 0.21, lo[2] = {-0.3665191429188092, -0.3665191429188092}, hi[2] = {0.3665191429188092, 0.3665191429188092};
    if (argc >= 12) { qq[0] = atof(argv[4]); qq[1] = atof(argv[5]); rr[0] = atof(argv[6]); rr[1] = atof(argv[7]); step = atof(argv[8]); wb = atof(argv[9]); lo[0] = lo[1] = atof(argv[10]); hi[0] = hi[1] = atof(argv[11]); }
    const double eps = 0.01;
    const unsigned long max_iter = 10000, smo_iters = 50;
    std::vector<unsigned char> depth;
    for (long k = 0; k < n; ++k) {
        const T v = in[k], ty = in[n + k], tphi = in[2 * n + k];
        Unit<T, true> m;
        m.set_uniform((T)1, qq, rr, lo, hi);
        m.set_instance((T)step, (T)wb, v, ty, tphi);
        T x[2 * H], vv[2 * H], wz[H], wy[H], dd[2 * H], iqd[2 * H];
        for (int i = 0; i < H; ++i) { x[2 * i] = m.xz0; x[2 * i + 1] = m.xz1; }
        const T lambda = ctor_lambda_qdiag<T, H>(m.a, m.c, qq[0], qq[1], rr[0], rr[1], [&](int i, int j, T val) {
            iqd[2 * i + j] = val != (T)0 ? (T)1 / (val * m.s(j)) : (T)0;
        });
        unsigned long iter = 0;
        bool stopped = false, vinit = false;
        double last_max_df = 0.0, first_max_df = 0.0;
        for (unsigned long it = 0; it < smo_iters && !stopped; ++it) {
            T Z, Y;
            m.fwd_init(Z, Y);
            for (int i = 0; i < H; ++i) { m.fwd(Z, Y, x[2 * i], x[2 * i + 1]); wz[i] = Z; wy[i] = Y; }
            T n0, n1;
            m.bwd_last(n0, n1, Z, Y);
            for (int i = H - 1; i >= 0; --i) {
                if (i < H - 1) m.bwd(n0, n1, wz[i], wy[i]);
                dd[2 * i] = m.df0(n1, x[2 * i]);
                dd[2 * i + 1] = m.df1(n0, n1, x[2 * i + 1]);
            }
            T max_df = (T)0;
            int best = 0;
            for (int qv = 0; qv < 2 * H; ++qv) {
                const T up = (x[qv] <= m.bl(qv & 1)) ? (T)0 : dd[qv];
                const T dn = (x[qv] >= m.bh(qv & 1)) ? (T)0 : -dd[qv];
                const T mag = max_(up, dn);
                if (mag > max_df) { max_df = mag; best = qv; }
            }
            last_max_df = (double)max_df;
            if (it == 0) first_max_df = (double)max_df;
            if (max_df < eps) { stopped = true; break; }
            if (iqd[best] != (T)0) {
                x[best] = m.project(fma_(-iqd[best], dd[best], x[best]), best & 1);
                vinit = (it + 1 == smo_iters);
            }
            ++iter;
        }
        double feat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double accs[256];
        for (int j = 0; j < 256; ++j) accs[j] = 0.0;
        {   // features of the state the PG phase starts from: sums of the gradient over the horizon (smooth modes), bound counts
            T Z, Y;
            m.fwd_init(Z, Y);
            for (int i = 0; i < H; ++i) { m.fwd(Z, Y, x[2 * i], x[2 * i + 1]); wz[i] = Z; wy[i] = Y; }
            T n0, n1;
            m.bwd_last(n0, n1, Z, Y);
            for (int i = H - 1; i >= 0; --i) {
                if (i < H - 1) m.bwd(n0, n1, wz[i], wy[i]);
                const double d0 = (double)m.df0(n1, x[2 * i]), d1 = (double)m.df1(n0, n1, x[2 * i + 1]);
                const bool b0 = x[2 * i] <= m.bl(0) || x[2 * i] >= m.bh(0), b1 = x[2 * i + 1] <= m.bl(1) || x[2 * i + 1] >= m.bh(1);
                feat[0] += b0 ? 0 : d0; feat[1] += b1 ? 0 : d1;
                feat[2] += b0 ? 0 : i * d0; feat[3] += b1 ? 0 : i * d1;
                feat[4] += b0; feat[5] += b1;
                feat[6] += b0 ? 0 : d0 * d0; feat[7] += b1 ? 0 : d1 * d1;
            }
        }
        depth.clear();
        if (!stopped) {
            const T g = GradScale<T>::g;
            m.set_uniform(g, qq, rr, lo, hi);
            m.set_instance((T)step, (T)wb, v, ty, tphi);
            const T geps = g * eps;
            T il0, il1, beta;
            pg_constants<T>(lambda, m.s0, m.s1, il0, il1, beta);
            for (int i = 0; i < H; ++i) { vv[2 * i] = vinit ? x[2 * i] : m.xz0; vv[2 * i + 1] = vinit ? x[2 * i + 1] : m.xz1; }
            unsigned long pgk = 0;
            while (true) {
                T Z, Y;
                m.fwd_init(Z, Y);
                for (int i = 0; i < H; ++i) { m.fwd(Z, Y, x[2 * i], x[2 * i + 1]); wz[i] = Z; wy[i] = Y; }
                T n0, n1, acc = (T)0;
                m.bwd_last(n0, n1, Z, Y);
                int decided = H + 1;
                for (int i = H - 1; i >= 0; --i) {
                    if (i < H - 1) m.bwd(n0, n1, wz[i], wy[i]);
                    for (int j = 0; j < 2; ++j) {
                        const int qv = 2 * i + j;
                        const T xx = x[qv];
                        const T d = j == 0 ? m.df0(n1, xx) : m.df1(n0, n1, xx);
                        const T xn = pg_update<true>(m, j, xx, d, j == 0 ? il0 : il1, beta, vv[qv]);
                        const T vn = vv[qv];
                        acc = max_(acc, min_(abs_(d), abs_(xx - vn)));
                        x[qv] = xn;
                    }
                    if (decided == H + 1 && !(acc < geps)) decided = H - i;
                }
                depth.push_back((unsigned char)decided);
                if (pgk < 256) accs[pgk] = (double)(acc / g);
                ++pgk;
                if (acc < geps) break;
                ++iter;
                if (iter >= max_iter) break;
            }
        }
        const unsigned int cnt = (unsigned int)depth.size(), cd = (unsigned int)iter - (cnt ? cnt - 1 : 0);
        fwrite(&cnt, 4, 1, fo);
        fwrite(&cd, 4, 1, fo);
        if (fx) { const double rec[3] = {(double)lambda, last_max_df, (double)cnt}; fwrite(rec, 8, 3, fx); fwrite(&first_max_df, 8, 1, fx); fwrite(feat, 8, 8, fx); fwrite(accs, 8, 256, fx); }
        if (cnt) fwrite(depth.data(), 1, cnt, fo);
    }
    fclose(fo);
    if (fx) fclose(fx);
    return 0;
}
