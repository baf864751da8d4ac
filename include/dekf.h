/*
 * dekf.h — C ABI of the batched decentralized EKF + MHE estimator for MI355X.
 *
 * This is the drop-in boundary for ONE hot path of well-robotics/Decentralized_EKF_MHE:
 * the orien_est quaternion EKF and the decentral_legged_est MHE/KF update.  The
 * reference has no FFI; its boundary is the C++ class
 *     DecentralizedEstimation::{initialize, update, reset}
 *         (src/decentral_legged_est/include/decentral_legged_est/DecentralEst.hpp:96-103)
 * and the EKF methods gyro_nonlinear_predict / gyro_nonlinear_correct /
 * vo_nonlinear_correct driven by orien_ekf::timerCallback
 *         (src/orien_est/include/orien_ekf.hpp:79-81, src/orien_est/src/orien_ekf.cpp:77-106).
 * Every entry point below names the reference interface it replaces.  The batch
 * dimension B (independent robot instances; on the GPU the solve gives each one a workgroup of four
 * wavefronts, the EKF one lane, term construction one wavefront) is new.
 *
 * Conventions
 *  - plain pointers + sizes, no C++/torch types; all arrays are instance-major
 *    ([B][...], row-major inside an instance), doubles like the reference's
 *    Eigen::*d / ROS float64 fields.
 *  - every pointer argument is either a HOST or a DEVICE (HBM) pointer, selected
 *    by the `dekf_mem` argument of the call.
 *  - all calls are asynchronous on the handle's HIP stream except dekf_create,
 *    dekf_destroy, dekf_sync and host-side dekf_get.
 *  - nothing throws; every call returns a dekf_status.
 *  - there is NO CPU fallback: dekf_create fails with DEKF_ERR_NO_DEVICE when no
 *    gfx950 device is usable.
 */
#ifndef DEKF_H
#define DEKF_H

#ifdef __cplusplus
extern "C" {
#endif

#define DEKF_ABI_VERSION 4 /* 4: dekf_params.polish_accept_osqp (appended), solve_workgroups_per_cu = 4 selects the four-per-CU kernels;
                            * 3: osqp.polish implemented: dekf_params.polish_refine_iter, dekf_get_polish_status;
                            * 2: dekf_params.solve_workgroups_per_cu, dekf_solve_kernel_name, dekf_launch_info; dim_state-sized rows */
#define DEKF_MAX_LEGS 4
#define DEKF_MAX_JOINTS 8 /* joints per leg */

typedef enum dekf_status {
    DEKF_OK = 0,
    DEKF_ERR_INVALID = 1,    /* bad argument / unsupported configuration */
    DEKF_ERR_NO_DEVICE = 2,  /* no usable HIP device (no CPU fallback exists) */
    DEKF_ERR_HIP = 3,        /* a HIP runtime call failed; see dekf_last_error */
    DEKF_ERR_ORDER = 4,      /* call sequence violated (e.g. update before initialize) */
    DEKF_ERR_COMM = 5        /* RCCL communicator error */
} dekf_status;

typedef enum dekf_mem { DEKF_HOST = 0, DEKF_DEVICE = 1 } dekf_mem;

/*
 * Parameter block.  Field-for-field mirror of `robot_params`
 * (DecentralEst.hpp:18-63; ROS names in EstSub.cpp:123-208, values in
 * go1_example/config/parameters_go1.yaml:1-50) plus the orien_ekf parameters
 * (orien_ekf.cpp:13-25, parameters_go1.yaml:68-75).  Three-element std vectors
 * stay three elements; the per-joint encoder stds are widened to
 * joints_per_leg entries so non-Go1 legs (BASELINE configs 3, 5) can be described.
 */
typedef struct dekf_params {
    /* prior.* */
    double p_init_std[3];
    double v_init_std[3];
    double foot_init_std[3];
    double accel_bias_init_std[3];
    /* process.* */
    double p_process_std[3];
    double accel_input_std[3];
    double gyro_input_std[3];
    double accel_bias_std[3]; /* process.accel_bias_process_std */
    /* leg_odom.* */
    double quaternion_ib[4]; /* w x y z */
    double p_ib[3];
    int num_legs;       /* leg_odom.num_leg */
    int joints_per_leg; /* 3 on Go1 (hard-coded block<3,3> in the reference) */
    int leg_odom_type;  /* 0: foot-velocity measurements, dim_state 9.  1: foot positions are states,
                         * dim_state = 9 + 3 num_legs (21 on Go1), DecentralEst.cpp:20.  How the arrival cost of type 1 is
                         * updated is chosen by `arrival_cost_form` below; its default (0) is the reference's covariance-form
                         * saddle inverse (MheSrb.cpp:527-651).
                         * CONTRACT of type 1: the BASE states (p, v, accel bias: what the node logs and publishes) are inside the
                         * 1e-4 relative tolerance of the reference formula (measured 0.16 x of it over 32 x 2000 ticks); the
                         * FOOT-POSITION states are OUTSIDE that contract: <= 3 x the tolerance (measured 1.83 x; form 1: <= 5 x,
                         * measured 3.24 x).  The reference formula itself moves by 1.5 x the tolerance on those states when its
                         * 33-dim saddle inverse is evaluated in another pivot order (a 1e20 - 1e20 cancellation at every
                         * touch-down, DecentralEst.cpp:432-451, 550-563; MheSrb.cpp:527-651) — INTEGRATION.md section 5. */
    double joint_position_std[DEKF_MAX_JOINTS];
    double joint_velocity_std[DEKF_MAX_JOINTS];
    double foot_slide_std[3];
    double foot_swing_std[3];
    double contact_effort_threshold;
    /* visual_odom.* */
    double vo_p_std[3];
    /* estimation.* */
    int rate;     /* Hz of update(T) calls; dt = 1/rate */
    int N;        /* horizon */
    int est_type; /* 0 MHE, 1 KF */
    /* osqp.* (DecentralEst.cpp:204-217) */
    double rho, alpha, delta, sigma;
    int verbose, adapt_rho, polish, max_qp_iter; /* polish: OSQP's solution polishing after a solved QP (DecentralEst.cpp:207;
                                                   * declared default true, EstSub.cpp:188; parameters_go1.yaml:44 sets false) */
    double rel_tol, abs_tol, prim_tol, dual_tol;
    double time_limit; /* accepted, NOT honoured: runs are deterministic */
    /* OSQP defaults the reference leaves implicit, made explicit here */
    int scaling_iters;            /* 10 */
    int check_termination;        /* 25 */
    int adaptive_rho_interval;    /* OSQP: wall-clock derived; here fixed, default 25 */
    double adaptive_rho_tolerance; /* 5 */
    /* orien_sub.* */
    double ekf_init_std[4];
    double ekf_process_std[3];      /* gyro */
    double ekf_gravity_meas_std[3]; /* accel */
    double ekf_vo_meas_std[4];
    double ekf_quaternion_init[4];  /* w x y z */
    int ekf_rate;                   /* 500 */
    int ekf_history;                /* depth of the rewind ring (reference: unbounded); default 256 = 0.5 s at 500 Hz (an ORB-SLAM3
                                     * relocalisation stall), 55 KB per instance; size it as
                                     * ceil(worst VO pose latency * ekf_rate) + 2, INTEGRATION.md section 5 */
    int polish_refine_iter;         /* OSQP's polish_refine_iter (refinement steps of the polishing solve; OSQP default 3, the
                                     * reference does not set it).  Used when `polish` is on. */
    int arrival_cost_form;          /* leg_odom_type 1 only.  0: the reference's covariance-form saddle inverse
                                     * (MheSrb.cpp:527-651).  1: information form (same arrival cost in exact arithmetic,
                                     * computed from gains only). leg_odom_type 0 always uses the reference form. */
    /* launch tuning (new) */
    int solve_pipeline;             /* 0 (default): every kernel of a step in order on the handle's stream.  1: the MHE solve of
                                     * step T runs on a second stream out of double-buffered inputs and outputs, so the pushes, the
                                     * EKF tick and the term construction of step T + 1 (and then its solve) start while the last
                                     * workgroups of step T's solve are still running; getters wait for the newest solve in stream
                                     * order, so results are bit-identical.  Measured on MI355X, Go1, three-workgroup kernels: +3.4 % at
                                     * B = 4096 (2.21 M against 2.14 M steps/s, round 6), +3 ... +10 % on the other shapes at 1024-4096, -4.5 % at 8192 (EXPERIMENTS.md round 5 section 8): it hides the 0.07 ms of EKF + term
                                     * construction + launch gaps in front of every solve and the partly empty last round.  The two solve streams are created
                                     * at the greatest stream priority (their own hardware-queue class). */
    int solve_workgroups_per_cu;    /* 0: the default residency (3 for full Go1 / Cassie windows when the batch exceeds the 512
                                     * slots of the two-workgroup kernels, 2 for full PogoX windows above 256); 1 or 2: cap — 2
                                     * keeps the two-workgroup solve kernels for full windows too; 4 (round 6, Go1 / Cassie full
                                     * windows, batches above 512): workgroups of THREE wavefronts at four per CU
                                     * (k_mhe_solve_r4_*): +36 / +12 / +7 % at 1024 / 2048 / 3072 robots, +-1 % from 4096 on, where
                                     * the hardware leaves 2-9 % of these workgroups unplaced until others finish
                                     * (profiles/r06_go1_four_per_cu_3waves.txt) — an opt-in for fleets of 800-3000 robots per GPU.
                                     * A launch-tuning knob only: every
                                     * kernel family produces the SAME BITS for a given robot log (since round 5 the iteration phases
                                     * are compiled with floating-point contraction off and explicit fma; tests/test_gpu_configs.py
                                     * holds r3 == ll, r3 == lg and rr == gg with array_equal), so a robot's estimate does not depend
                                     * on the size of the fleet it is batched with. */
    int polish_accept_osqp;         /* osqp.polish only.  0 (default): OSQP's acceptance test for a polished point with its third clause
                                     * made symmetric — `pol_dua < dua && pri < 1e-10 && pol_pri < 1e-10` — so that a polished point never
                                     * trades a primal residual of 1e-11 for one of 1e-7 (this solver's iterates
                                     * reach that clause where OSQP's own do not).  1: polish.c's test verbatim
                                     * (`pol_dua < dua && pri < 1e-10`), for strict OSQP parity. */
} dekf_params;

typedef struct dekf_handle_s* dekf_handle;

/* Fill `p` with go1_example/config/parameters_go1.yaml + OSQP defaults. */
void dekf_default_params(dekf_params* p);

int dekf_abi_version(void);
const char* dekf_last_error(void);
/* hipRuntimeGetVersion() of the HIP runtime this library is bound to (0 if the call fails).  The overlap of consecutive steps
 * (solve_pipeline = 1) and of the look-ahead arrival cost rests on stream scheduling of that runtime; bench.py records it. */
int dekf_hip_runtime_version(void);

/* Replaces: constructing DecentralizedEstimation + orien_ekf for `batch` robots.
 * device = HIP device ordinal; stream = hipStream_t to run on (NULL: own stream). */
dekf_status dekf_create(const dekf_params* p, int batch, int device, void* stream,
                        dekf_handle* out);
dekf_status dekf_destroy(dekf_handle h);
/* Replaces DecentralizedEstimation::reset (DecentralEst.cpp:1011-1015): clears the
 * QP window, the arrival cost, the measurement stacks and the EKF state. */
dekf_status dekf_reset(dekf_handle h);
dekf_status dekf_sync(dekf_handle h);
int dekf_batch(dekf_handle h);
void* dekf_stream(dekf_handle h);

/* ---- sensor latches: the writes the ROS callbacks make into robot_store ---------- */

/* go1Sub::imu_callback / orien_ekf::imu_callback (go1Sub.cpp:30-51, orien_ekf.cpp:62-75):
 * imu_time[B], accel_b[B][3], gyro_b[B][3]. */
dekf_status dekf_push_imu(dekf_handle h, const double* imu_time, const double* accel_b,
                          const double* gyro_b, dekf_mem where);

/* go1Sub::lo_callback outputs (go1Sub.cpp:53-126): p_imu_2_foot[B][L][3],
 * J_imu_2_foot[B][L][3][nj], joint_velocity[B][L][nj], contact[B][L] (0/1). */
dekf_status dekf_push_leg(dekf_handle h, const double* p_imu_2_foot, const double* J_imu_2_foot,
                          const double* joint_velocity, const double* contact, dekf_mem where);

/* Same callback one step earlier: raw Go1 joint_position[B][12],
 * joint_velocity[B][12], foot_force[B][4]; FK, Jacobian and the contact threshold
 * run on the device. Go1 only (num_legs 4, joints_per_leg 3). */
dekf_status dekf_push_go1_joints(dekf_handle h, const double* joint_position,
                                 const double* joint_velocity, const double* foot_force,
                                 dekf_mem where);

/* robotSub::vo_callback (EstSub.cpp:45-56) + orien_ekf::vo_pose_callback
 * (orien_ekf.cpp:48-60): mask[B] (int, 1 = a new VO sample for this instance),
 * t_pre[B], t_now[B], dp_body[B][3] (orb/vo), q_vo[B][4] wxyz and its stamp
 * t_pose[B] (orb/pos). Instances with mask 0 are untouched. */
dekf_status dekf_push_vo(dekf_handle h, const int* mask, const double* t_pre, const double* t_now,
                         const double* dp_body, const double* t_pose, const double* q_vo,
                         dekf_mem where);

/* robotSub::orien_filter_callback (EstSub.cpp:34-43): overrides the orientation the
 * MHE reads with an external quaternion[B][4] wxyz instead of the on-device EKF's. */
dekf_status dekf_push_quaternion(dekf_handle h, const double* quat, dekf_mem where);

/* ---- the hot path ------------------------------------------------------------- */

/* orien_ekf::timerCallback (orien_ekf.cpp:77-106): history push, VO rewind/replay
 * when a VO pose is pending, predict, accel-correct; result feeds the MHE latch. */
dekf_status dekf_ekf_step(dekf_handle h);

/* DecentralizedEstimation::initialize (DecentralEst.cpp:9-150), T = 0. */
dekf_status dekf_initialize(dekf_handle h);
/* DecentralizedEstimation::update(T) (DecentralEst.cpp:152-198), T = 1, 2, ... */
dekf_status dekf_update(dekf_handle h, int T);
/* One estimator-step of the benchmark metric: dekf_ekf_step then
 * (T == 0 ? dekf_initialize : dekf_update(T)). */
dekf_status dekf_step(dekf_handle h, int T);

/* ---- results: the public members read by EstSub.cpp:99-106 ---------------------- */
/* x_mhe[B][dim_state] (x_MHE_ / x_KF_; dim_state = 9 + 3 * leg_odom_type * num_legs: p_s, v_s, accel bias,
 * then the foot positions of leg_odom_type 1), v_b[B][3] (v_MHE_b_ / v_KF_b_), quat[B][4] (EKF
 * quaternion_, wxyz), p_vo[B][3] (p_vo_accmulate_), status[B] (int, see below).
 * Any pointer may be NULL. */
dekf_status dekf_get(dekf_handle h, double* x_mhe, double* v_b, double* quat, double* p_vo,
                     int* status, dekf_mem where);
/* EKF covariance Cov_q_[B][4][4]. */
dekf_status dekf_get_ekf_cov(dekf_handle h, double* cov, dekf_mem where);
/* Per-instance solver diagnostics of the last update: iters[B], rho_updates[B] (int),
 * pri_res[B], dua_res[B] (unscaled OSQP residuals). Any pointer may be NULL. */
dekf_status dekf_get_solver_info(dekf_handle h, int* iters, int* rho_updates, double* pri_res,
                                 double* dua_res, dekf_mem where);
/* Outcome of OSQP's polishing step per instance, polish_status[B] (int; OSQP's info->status_polish): 0 polishing off or the
 * solve did not end OSQP_SOLVED, 1 the polished point replaced the ADMM iterate (pri_res / dua_res are then its residuals),
 * -1 polishing ran and was rejected (the ADMM iterate is returned). */
dekf_status dekf_get_polish_status(dekf_handle h, int* polish_status, dekf_mem where);
/* KF covariance C_KF_[B][dim_state][dim_state] (est_type 1). */
dekf_status dekf_get_kf_cov(dekf_handle h, double* cov, dekf_mem where);

/* status[B] values written by dekf_update */
#define DEKF_SOLVE_NONE 0       /* no solve yet (T = 0) */
#define DEKF_SOLVE_OK 1         /* OSQP_SOLVED */
#define DEKF_SOLVE_MAX_ITER 2   /* OSQP_MAX_ITER_REACHED; iterate still returned, as osqp-eigen does */
#define DEKF_SOLVE_NUMERIC -1   /* non-finite value or zero pivot */

/* ---- measurement (the reference's tic/toc helpers, DecentralEst.cpp:1031-1044) ------- */
/* on = 1: every kernel launch of the hot path is bracketed by HIP events on the handle's stream; on = 2: only the
 * MHE solve launches (class 2) — an event pair costs the stream about 7 us, six of them per step are 1 % of a
 * 2 ms step, so a throughput measurement that only needs the dominant kernel's launch time asks for 2; on = 0: off.  dekf_timing_read synchronises and returns, per kernel class
 * (0 ekf tick, 1 MHE assemble/marginalise [or KF update], 2 MHE ADMM solve, 3 all-gather of v_b), the summed
 * device milliseconds and the number of launches since the last read. */
#define DEKF_TIMING_CLASSES 4 /* (ABI 4: class 3 = the RCCL all-gather of v_b, bracketed on the handle's communication stream in either
                               * timing mode — it is not on the step's critical path) */
dekf_status dekf_timing_enable(dekf_handle h, int on);
dekf_status dekf_timing_read(dekf_handle h, double* ms_sum, int* launches);

/* How the solve kernel is launched on this device: the number of persistent workgroups (each solves
 * ceil(B / solve_workgroups) instances back to back per launch), the compute units and the engine clock in Hz —
 * what a caller needs to turn a launch time into cycles per solve. Any pointer may be NULL. */
dekf_status dekf_launch_info(dekf_handle h, int* solve_workgroups, int* compute_units, double* clock_hz);
/* Name of the solve kernel this handle launches for full windows (full_window != 0) or for the window-fill ticks,
 * e.g. "k_mhe_solve_r3_4_n20": lets a caller match profiler output (rocprofv3 kernel names) to the run. NULL for a
 * KF handle. The string lives as long as the library. */
const char* dekf_solve_kernel_name(dekf_handle h, int full_window);

/* ---- multi-GPU (new: the reference is single-robot) ------------------------------ */
/* All-gather of the fused base velocity over RCCL: every rank contributes its
 * v_b[B][3] and receives v_b_all[world][B][3] (device pointer). The communicator is
 * created from an ncclUniqueId distributed by the caller (torch.distributed, MPI, ...).
 *
 * dekf_allgather_vb is asynchronous twice over: it snapshots v_b in stream order (after
 * the step that produced it) and runs the collective on a second stream owned by the
 * handle, so the exchange of step T overlaps the kernels of step T+1 and a slow rank
 * delays the others by at most one step of slack. v_b_all is complete after dekf_sync,
 * or in stream order after dekf_allgather_wait (which makes dekf_stream() wait for the
 * last all-gather without blocking the host). Use a different v_b_all buffer while the
 * previous one is still being read on another stream. */
#define DEKF_UNIQUE_ID_BYTES 128
dekf_status dekf_comm_unique_id(void* id_out);
dekf_status dekf_comm_init(dekf_handle h, int world, int rank, const void* id);
dekf_status dekf_allgather_vb(dekf_handle h, double* v_b_all_dev);
dekf_status dekf_allgather_wait(dekf_handle h);
/* What the COMMUNICATOR itself says about the exchange (ncclCommCount / ncclCommUserRank of the handle's communicator), not what
 * the caller passed to dekf_comm_init — so that a throughput line can prove how many ranks RCCL really connected.  And a device-side
 * proof of one exchange: every rank contributes its own rank number, dekf_comm_ranks_seen all-gathers them on the communication
 * stream, waits, and writes how many DISTINCT, in-range rank numbers arrived (== world on a working communicator).  Both are
 * collectives over the communicator's ranks where noted.  Any out pointer may be NULL. */
dekf_status dekf_comm_info(dekf_handle h, int* comm_world, int* comm_rank);
dekf_status dekf_comm_ranks_seen(dekf_handle h, int* ranks_seen); /* collective */

#ifdef __cplusplus
}
#endif
#endif /* DEKF_H */
