"""The device cores under AddressSanitizer + UBSan (CPU build only: GPU ASan is not available on
this pool).  A stand-alone driver links tests/hostsim/hostsim.cpp with -fsanitize=address,undefined
and steps Go1, a 2-leg/5-joint robot, a long-horizon 1-leg robot and two leg_odom_type = 1 shapes (foot positions as states)
through window fill, marginalisation and VO updates, two of them with osqp.polish on; any out-of-bounds index in the kernels'
LDS/HBM carving aborts it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

DRIVER = r'''
#include "hostsim.cpp"
#include <cstdio>
static int run(int L, int nj, int N, int steps, int ft = 0, int form = 0, int polish = 0) {
    dekf_params p; default_params(&p); p.ekf_rate = 200; p.num_legs = L; p.joints_per_leg = nj; p.N = N; p.leg_odom_type = ft; p.arrival_cost_form = form; p.polish = polish;
    const int ns = 9 + 3 * L * ft;
    int B = 2;
    void* h = hs_create(&p, B);
    if (!h) return 1;
    std::vector<double> t(B), acc(3 * B), gy(3 * B), pf(3 * L * B), J(3 * L * nj * B), qd(L * nj * B), c(L * B);
    std::vector<int> mask(B, 1); std::vector<double> tp(B), tn(B), dp(3 * B), q(4 * B);
    for (int T = 0; T < steps; ++T) {
        for (int b = 0; b < B; ++b) {
            t[b] = 0.005 * T + 1e-5 * b;
            acc[3*b] = 0.1; acc[3*b+1] = -0.05; acc[3*b+2] = 9.8; gy[3*b] = 0.01; gy[3*b+1] = 0.02; gy[3*b+2] = 0.2;
            for (int i = 0; i < 3 * L; ++i) pf[3*L*b + i] = 0.1 * (i % 3) - 0.25;
            for (int i = 0; i < 3 * L * nj; ++i) J[3*L*nj*b + i] = (i % (nj + 1) == 0) ? 0.2 : 0.03 * ((i + T) % 5);
            for (int i = 0; i < L * nj; ++i) qd[L*nj*b + i] = 0.1 * ((i + T) % 7) - 0.3;
            for (int i = 0; i < L; ++i) c[L*b + i] = ((T / 5 + i) % 2) ? 1.0 : 0.0;
            tp[b] = 0.005 * (T - 7); tn[b] = 0.005 * (T - 1); dp[3*b] = 0.003; dp[3*b+1] = 0; dp[3*b+2] = 0;
            q[4*b] = 1; q[4*b+1] = q[4*b+2] = q[4*b+3] = 0;
        }
        hs_push_imu(h, t.data(), acc.data(), gy.data());
        hs_push_leg(h, pf.data(), J.data(), qd.data(), c.data());
        if (T > 8 && T % 6 == 0) hs_push_vo(h, mask.data(), tp.data(), tn.data(), dp.data(), tn.data(), q.data());
        hs_ekf_step(h);
        if (T == 0) hs_initialize(h); else hs_update(h, T);
    }
    std::vector<double> x(ns * B); std::vector<int> st(B), it(B);
    hs_get(h, x.data(), nullptr, nullptr, nullptr, st.data(), it.data(), nullptr);
    std::printf("L=%d nj=%d N=%d leg_odom_type=%d arrival_cost_form=%d: status %d iters %d v=%g\n", L, nj, N, ft, form, st[0], it[0], x[3]);
    hs_destroy(h);
    return st[0] == 1 ? 0 : 2;
}
// leg_odom_type 1 (foot positions as states, 21-dim blocks on Go1: its own solve family, factor in the slab) through window fill,
// marginalisation with swinging feet and VO updates, with BOTH forms of the arrival cost: 0 = the reference's covariance form
// (the default: pivoted generic inverse in the enlarged AsmScratch), 1 = information form (marginalize_info); and a 2-leg shape
int main() {
    // ... and osqp.polish (a second factorisation, the refinement steps' residual vectors, the save / load / accumulate of the polished
    // point in the slab scratch: mhe_solve_core.h polish_swap_in / polish_accumulate) on a 9-state and a 21-state shape
    return run(4, 3, 20, 60) | run(2, 5, 8, 30) | run(1, 3, 40, 70) | run(4, 3, 20, 34, 1, 0) | run(4, 3, 20, 34, 1, 1) |
           run(2, 5, 6, 24, 1, 0) | run(2, 5, 6, 24, 1, 1) | run(4, 3, 20, 44, 0, 0, 1) | run(2, 5, 6, 24, 1, 0, 1);
}
'''


def test_cores_are_clean_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-DDEKF_HOSTSIM", "-w", "-I", os.path.join(HERE, "hostsim"),
                           "-o", str(exe), str(src)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
