"""Parameter block of the estimator: ctypes mirror of ``dekf_params`` (include/dekf.h).

Field-for-field the reference's ``robot_params`` (DecentralEst.hpp:18-63) plus the
``orien_ekf`` parameters (orien_ekf.cpp:13-25).  Defaults are
``src/go1_example/config/parameters_go1.yaml`` of the reference.
"""
import ctypes as C

DEKF_MAX_LEGS = 4
DEKF_MAX_JOINTS = 8
D3 = C.c_double * 3
D4 = C.c_double * 4
DJ = C.c_double * DEKF_MAX_JOINTS


class DekfParams(C.Structure):
    _fields_ = [
        ("p_init_std", D3), ("v_init_std", D3), ("foot_init_std", D3), ("accel_bias_init_std", D3),
        ("p_process_std", D3), ("accel_input_std", D3), ("gyro_input_std", D3), ("accel_bias_std", D3),
        ("quaternion_ib", D4), ("p_ib", D3),
        ("num_legs", C.c_int), ("joints_per_leg", C.c_int), ("leg_odom_type", C.c_int),
        ("joint_position_std", DJ), ("joint_velocity_std", DJ),
        ("foot_slide_std", D3), ("foot_swing_std", D3),
        ("contact_effort_threshold", C.c_double),
        ("vo_p_std", D3),
        ("rate", C.c_int), ("N", C.c_int), ("est_type", C.c_int),
        ("rho", C.c_double), ("alpha", C.c_double), ("delta", C.c_double), ("sigma", C.c_double),
        ("verbose", C.c_int), ("adapt_rho", C.c_int), ("polish", C.c_int), ("max_qp_iter", C.c_int),
        ("rel_tol", C.c_double), ("abs_tol", C.c_double), ("prim_tol", C.c_double), ("dual_tol", C.c_double),
        ("time_limit", C.c_double),
        ("scaling_iters", C.c_int), ("check_termination", C.c_int), ("adaptive_rho_interval", C.c_int),
        ("adaptive_rho_tolerance", C.c_double),
        ("ekf_init_std", D4), ("ekf_process_std", D3), ("ekf_gravity_meas_std", D3),
        ("ekf_vo_meas_std", D4), ("ekf_quaternion_init", D4),
        ("ekf_rate", C.c_int), ("ekf_history", C.c_int),
        ("polish_refine_iter", C.c_int), ("arrival_cost_form", C.c_int), ("solve_pipeline", C.c_int), ("solve_workgroups_per_cu", C.c_int),
        ("polish_accept_osqp", C.c_int),
    ]

    def copy(self):
        other = DekfParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(DekfParams))
        return other

    @property
    def dim_state(self):
        return 9 + 3 * self.leg_odom_type * self.num_legs

    @property
    def dim_meas(self):
        return 3 * self.num_legs


def _set(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = v


def go1_params():
    """parameters_go1.yaml:1-75 (est_sub + orien_sub), OSQP defaults made explicit."""
    p = DekfParams()
    _set(p.p_init_std, [0.001] * 3)
    _set(p.v_init_std, [0.001] * 3)
    _set(p.foot_init_std, [0.001] * 3)
    _set(p.accel_bias_init_std, [0.0001] * 3)
    _set(p.p_process_std, [0.001] * 3)
    _set(p.accel_input_std, [0.025, 0.025, 0.02])
    _set(p.gyro_input_std, [0.03] * 3)
    _set(p.accel_bias_std, [0.07, 0.02, 0.03])
    _set(p.quaternion_ib, [1.0, 0.0, 0.0, 0.0])
    _set(p.p_ib, [0.01592, 0.06659, 0.00617])
    p.num_legs, p.joints_per_leg, p.leg_odom_type = 4, 3, 0
    _set(p.joint_position_std, [0.04] * DEKF_MAX_JOINTS)
    _set(p.joint_velocity_std, [0.22] * DEKF_MAX_JOINTS)
    _set(p.foot_slide_std, [0.003] * 3)
    _set(p.foot_swing_std, [1.0e7] * 3)
    p.contact_effort_threshold = 150.0
    _set(p.vo_p_std, [0.000015] * 3)
    p.rate, p.N, p.est_type = 200, 20, 0
    p.rho, p.alpha, p.delta, p.sigma = 0.1, 1.6, 1e-5, 1e-5
    p.verbose, p.adapt_rho, p.polish, p.max_qp_iter = 0, 1, 0, 4000
    p.rel_tol = p.abs_tol = p.prim_tol = p.dual_tol = 1e-6
    p.time_limit = 0.0028
    p.scaling_iters, p.check_termination, p.adaptive_rho_interval = 10, 25, 25
    p.adaptive_rho_tolerance = 5.0
    _set(p.ekf_init_std, [0.001] * 4)
    _set(p.ekf_process_std, [0.1] * 3)
    _set(p.ekf_gravity_meas_std, [4.0] * 3)
    _set(p.ekf_vo_meas_std, [0.0001] * 4)
    _set(p.ekf_quaternion_init, [1.0, 0.0, 0.0, 0.0])
    p.ekf_rate, p.ekf_history = 500, 256
    p.polish_refine_iter = 3
    p.arrival_cost_form, p.solve_pipeline, p.solve_workgroups_per_cu, p.polish_accept_osqp = 0, 0, 0, 0
    return p


def cassie_params():
    """BASELINE config 3: biped, 5 joints per leg (generalisation; not in the reference)."""
    p = go1_params()
    p.num_legs, p.joints_per_leg = 2, 5
    return p


def pogox_params():
    """BASELINE config 5: one-legged hopper, 0.5 s window (N = 100)."""
    p = go1_params()
    p.num_legs, p.joints_per_leg, p.N = 1, 3, 100
    return p
