"""ROS2 parameter files <-> ``DekfParams``.

The reference's nodes read their settings with ``declare_parameter`` / ``get_parameter`` from a ROS2
parameter file (``src/go1_example/config/parameters_go1.yaml``; names and defaults:
``src/decentral_legged_est/src/EstSub.cpp:123-208`` for ``est_sub``,
``src/orien_est/src/orien_ekf.cpp:13-25`` for ``orien_sub``).  ``load_ros_params`` reads the same file
with the same names and the same defaults for missing entries; ``dump_ros_params`` writes one.
The C++ twin is ``cpp/ros_params.hpp`` + ``paramsWrapper`` in ``cpp/est_node_core.hpp``.
"""
import yaml

from .params import DEKF_MAX_JOINTS, DekfParams, go1_params

# (yaml name, DekfParams field, default of the reference's declare_parameter)
EST_SUB = [
    ("prior.p_init_std", "p_init_std", [0.001] * 3),
    ("prior.v_init_std", "v_init_std", [0.001] * 3),
    ("prior.foot_init_std", "foot_init_std", [0.001] * 3),
    ("prior.accel_bias_init_std", "accel_bias_init_std", [0.001] * 3),
    ("process.p_process_std", "p_process_std", [0.01] * 3),
    ("process.accel_input_std", "accel_input_std", [0.01, 0.04, 0.001]),
    ("process.gyro_input_std", "gyro_input_std", [0.01] * 3),
    ("process.accel_bias_process_std", "accel_bias_std", [1.0, 1.0, 0.1]),
    ("leg_odom.quaternion_ib", "quaternion_ib", [1.0, 0.0, 0.0, 0.0]),
    ("leg_odom.p_ib", "p_ib", [0.0] * 3),
    ("leg_odom.num_leg", "num_legs", 4),
    ("leg_odom.leg_odom_type", "leg_odom_type", 0),
    ("leg_odom.joint_position_std", "joint_position_std", [0.01] * 3),
    ("leg_odom.joint_velocity_std", "joint_velocity_std", [0.01] * 3),
    ("leg_odom.foot_slide_std", "foot_slide_std", [0.001] * 3),
    ("leg_odom.foot_swing_std", "foot_swing_std", [10000.0] * 3),
    ("leg_odom.contact_effort_theshold", "contact_effort_threshold", 150.0),
    ("visual_odom.vo_p_std", "vo_p_std", [0.001] * 3),
    ("estimation.rate", "rate", 50),
    ("estimation.N", "N", 50),
    ("estimation.est_type", "est_type", 0),
    ("osqp.rho", "rho", 0.1),
    ("osqp.alpha", "alpha", 1.6),
    ("osqp.delta", "delta", 0.00001),
    ("osqp.sigma", "sigma", 0.00001),
    ("osqp.verbose", "verbose", True),
    ("osqp.adaptRho", "adapt_rho", True),
    ("osqp.polish", "polish", True),
    ("osqp.maxQPIter", "max_qp_iter", 1000),
    ("osqp.primTol", "prim_tol", 0.000001),
    ("osqp.dualTol", "dual_tol", 0.000001),
    ("osqp.realtiveTol", "rel_tol", 1e-3),
    ("osqp.absTol", "abs_tol", 1e-3),
    ("osqp.timeLimit", "time_limit", 0.005),
]
ORIEN_SUB = [
    ("init_std", "ekf_init_std", [0.001] * 4),
    ("process_std", "ekf_process_std", [0.1] * 3),
    ("gravity_meas_std", "ekf_gravity_meas_std", [4.0] * 3),
    ("vo_meas_std", "ekf_vo_meas_std", [0.0001] * 4),
    ("quaternion_init", "ekf_quaternion_init", [1.0, 0.0, 0.0, 0.0]),
    ("rate", "ekf_rate", 500),
]
# node-level settings that are not part of the estimator's parameter block
NODE_ONLY = [("log_name", "exp"), ("estimation.interval", 20)]


def _node_section(doc, node):
    for key in (node, "/" + node):
        if key in doc and isinstance(doc[key], dict):
            return doc[key].get("ros__parameters", {}) or {}
    return {}


def _lookup(section, dotted):
    cur = section
    for part in dotted.split("."):
        if not isinstance(cur, dict) or part not in cur:
            return None
        cur = cur[part]
    return cur


def _coerce(value, default, name):
    """the file's value as the type of the declared default (ints are accepted for doubles;
    PyYAML reads `1e-6` as a string)"""
    if isinstance(default, bool):
        if not isinstance(value, bool):
            raise TypeError(f"parameter {name}: expected a bool, got {value!r}")
        return value
    if isinstance(default, int):
        if isinstance(value, bool) or not isinstance(value, int):
            raise TypeError(f"parameter {name}: expected an integer, got {value!r}")
        return value
    if isinstance(default, float):
        if isinstance(value, bool):
            raise TypeError(f"parameter {name}: expected a number, got {value!r}")
        return float(value)
    if isinstance(default, list):
        if not isinstance(value, (list, tuple)):
            raise TypeError(f"parameter {name}: expected a sequence, got {value!r}")
        return [float(v) for v in value]
    return str(value)


def _params_of(doc, node, table):
    wild = (doc.get("/**") or {}).get("ros__parameters", {}) or {}
    sec = _node_section(doc, node)
    out = {}
    for name, _, default in table:
        v = _lookup(sec, name)
        if v is None:
            v = _lookup(wild, name)
        out[name] = default if v is None else _coerce(v, default, name)
    return out


def _assign(p, field, value):
    cur = getattr(p, field)
    if isinstance(value, list):
        if field in ("joint_position_std", "joint_velocity_std"):
            # the reference indexes these per joint of a leg; repeat the pattern over the joint slots
            value = [value[i % len(value)] for i in range(DEKF_MAX_JOINTS)]
        if len(value) < len(cur):
            raise ValueError(f"parameter for {field} needs {len(cur)} entries, got {len(value)}")
        for i in range(len(cur)):
            cur[i] = value[i]
    else:
        setattr(p, field, int(value) if isinstance(value, bool) else value)


def load_ros_params(path, est_node="est_sub", orien_node="orien_sub", joints_per_leg=3):
    """Parameter file -> (DekfParams, node settings dict with ``log_name`` and ``interval_ms``).

    Entries the file does not set get the defaults the reference's nodes declare (which differ from
    ``parameters_go1.yaml``: N 50, rate 50, maxQPIter 1000, ...).  Fields the reference leaves to OSQP's
    defaults (scaling passes, termination cadence, adaptive-rho) come from ``go1_params()``."""
    with open(path) as f:
        doc = yaml.safe_load(f) or {}
    p = go1_params()
    p.joints_per_leg = joints_per_leg
    for table, node in ((EST_SUB, est_node), (ORIEN_SUB, orien_node)):
        vals = _params_of(doc, node, table)
        for name, field, _ in table:
            _assign(p, field, vals[name])
    node_vals = _params_of(doc, est_node, [(n, None, d) for n, d in NODE_ONLY])
    return p, {"log_name": node_vals["log_name"], "interval_ms": node_vals["estimation.interval"]}


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        return repr(v)
    if isinstance(v, str):
        return '"' + v + '"'
    return "[" + ", ".join(repr(float(x)) for x in v) + "]"


def dump_ros_params(p: DekfParams, path=None, log_name="exp", interval_ms=None, est_node="est_sub",
                    orien_node="orien_sub"):
    """DekfParams -> text of a ROS2 parameter file with both nodes' sections (written to ``path`` if given)."""
    bools = {"verbose", "adapt_rho", "polish"}

    def value(field, default):
        v = getattr(p, field)
        if isinstance(default, list):
            return [v[i] for i in range(len(default))]
        return bool(v) if field in bools else v

    lines = [f"{est_node}:", "  ros__parameters:", f"    log_name: {_fmt(log_name)}"]
    group = None
    for name, field, default in EST_SUB:
        g, key = name.split(".")
        if g != group:
            lines.append(f"    {g}:")
            group = g
        lines.append(f"      {key}: {_fmt(value(field, default))}")
        if name == "estimation.rate":
            iv = interval_ms if interval_ms is not None else max(1, round(1000 / max(p.rate, 1)))
            lines.append(f"      interval: {int(iv)}")
    lines += ["", f"{orien_node}:", "  ros__parameters:"]
    for name, field, default in ORIEN_SUB:
        lines.append(f"    {name}: {_fmt(value(field, default))}")
    text = "\n".join(lines) + "\n"
    if path is not None:
        with open(path, "w") as f:
            f.write(text)
    return text
