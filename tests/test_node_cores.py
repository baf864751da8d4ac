"""ROS2-free cores of the reference's two nodes: parameter files, callbacks, timer logic.

CPU: the parameter readers (Python and C++) agree with each other, with the defaults the reference's nodes
declare and — when the reference is mounted — with its own parameters_go1.yaml; the replay program compiles
and links.  GPU: a synthetic message sequence replayed through orien_sub -> est_sub (raw joint states,
kinematics on the device) reproduces the oracle's estimates and writes the reference's log format.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.logger import read_log
from decentralized_ekf_mhe_amd.ros_params import EST_SUB, ORIEN_SUB, dump_ros_params, load_ros_params
from decentralized_ekf_mhe_amd.streams import make_streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc")
REF_YAML = "/root/reference/src/go1_example/config/parameters_go1.yaml"


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "examples", name + ".cpp"),
                           "-o", exe, "-L" + CSRC, "-ldekf", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _fields(p):
    out = {}
    for name, ctype in p._fields_:
        v = getattr(p, name)
        out[name] = [float(x) for x in v] if hasattr(v, "__len__") else float(v)
    return out


def _cpp_dump(exe, path):
    r = subprocess.run([exe, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out, extra = {}, {"undeclared": []}
    for line in r.stdout.strip().splitlines():
        key, *vals = line.split()
        if key == "log_name":
            extra["log_name"] = vals[0]
        elif key == "interval_ms":
            extra["interval_ms"] = int(vals[0])
        elif key == "undeclared":
            extra["undeclared"].append(vals[0])
        else:
            out[key] = [float(v) for v in vals] if len(vals) > 1 else float(vals[0])
    return out, extra


TRICKY = """
# wildcard section, comments, a sequence that starts on the next line, integers where doubles are declared
/**:
  ros__parameters:
    osqp:
      maxQPIter: 77        # only used where the node's own section is silent
      rho: 0.5
est_sub:
  ros__parameters:
    log_name: 'run #7'     # the hash inside the quotes is not a comment
    leg_odom:
      p_ib:
        [0.1, 2,
         -3.5e-1]
      contact_effort_theshold: 120
      num_leg: 4
    estimation: {}
    osqp:
      rho: 0.25
      verbose: false
      primTol: 1e-7
    unknown_block:
      typo_param: 3
vo_sub:
  ros__parameters:
    R_ic:
      [1, 0, 0,
       0, 1, 0,
       0, 0, 1]
    image_topic_left: "/camera/infra1/image_rect_raw"
orien_sub:
  ros__parameters:
    rate: 250
    quaternion_init: [0.5, 0.5, 0.5, 0.5] # w x y z
"""


def test_python_dump_load_roundtrip(tmp_path):
    p = go1_params()
    p.N, p.rate, p.est_type = 33, 100, 1
    p.accel_bias_std[1] = 0.123
    path = tmp_path / "p.yaml"
    dump_ros_params(p, path, log_name="trial", interval_ms=10)
    q, node = load_ros_params(path)
    assert _fields(q) == _fields(p)
    assert node == {"log_name": "trial", "interval_ms": 10}


def test_missing_entries_get_the_declared_defaults(tmp_path):
    path = tmp_path / "empty.yaml"
    path.write_text("est_sub:\n  ros__parameters:\n    log_name: \"x\"\n")
    p, node = load_ros_params(path)
    f = _fields(p)
    for name, field, default in EST_SUB + ORIEN_SUB:
        want = [float(v) for v in default] if isinstance(default, list) else float(default)
        got = f[field]
        if isinstance(want, list):
            got = got[:len(want)]
        assert got == want, name
    # EstSub.cpp:177-181, 186-197: not the Go1 file's values
    assert (p.N, p.rate, p.max_qp_iter, p.polish, node["interval_ms"]) == (50, 50, 1000, 1, 20)


def test_cpp_reader_matches_python_reader(tmp_path):
    exe = _build(tmp_path, "ros_params_dump")
    p = go1_params()
    p.N, p.est_type = 27, 1
    a = tmp_path / "a.yaml"
    dump_ros_params(p, a, log_name="abc", interval_ms=5)
    b = tmp_path / "b.yaml"
    b.write_text(TRICKY)
    for path in (a, b):
        got, extra = _cpp_dump(exe, path)
        want, node = load_ros_params(path)
        assert got == _fields(want), path
        assert (extra["log_name"], extra["interval_ms"]) == (node["log_name"].split()[0], node["interval_ms"])
    got, extra = _cpp_dump(exe, b)
    assert got["rho"] == 0.25 and got["max_qp_iter"] == 77 and got["prim_tol"] == 1e-7
    assert got["p_ib"] == [0.1, 2.0, -0.35] and got["contact_effort_threshold"] == 120.0
    assert got["ekf_rate"] == 250 and got["ekf_quaternion_init"] == [0.5] * 4
    assert extra["undeclared"] == ["unknown_block.typo_param"]


def test_cpp_reader_rejects_a_wrong_type(tmp_path):
    exe = _build(tmp_path, "ros_params_dump")
    bad = tmp_path / "bad.yaml"
    bad.write_text("est_sub:\n  ros__parameters:\n    estimation:\n      N: [1, 2]\n")
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "estimation.N" in r.stderr
    with pytest.raises(TypeError):
        load_ros_params(bad)


def _emit(d, indent=0):
    """block-style YAML text of a nested dict, with the quirks the reader has to cope with"""
    out = []
    pad = " " * indent
    for k, v in d.items():
        if isinstance(v, dict):
            out.append(f"{pad}{k}:")
            out += _emit(v, indent + 2)
        elif isinstance(v, bool):
            out.append(f"{pad}{k}: {'true' if v else 'false'}")
        elif isinstance(v, list):
            items = ", ".join(repr(x) for x in v)
            if len(v) > 3:   # a sequence that starts on the next line and is broken in the middle
                half = len(v) // 2
                out.append(f"{pad}{k}:   # comment after the key")
                out.append(f"{pad}  [{', '.join(repr(x) for x in v[:half])},")
                out.append(f"{pad}   {', '.join(repr(x) for x in v[half:])}]")
            else:
                out.append(f"{pad}{k}: [{items}]  # trailing comment")
        elif isinstance(v, str):
            out.append(f'{pad}{k}: "{v}"')
        else:
            out.append(f"{pad}{k}: {v!r}")
    return out


def test_cpp_reader_agrees_with_pyyaml_on_generated_files(tmp_path):
    """property test of the hand-written reader: random nested parameter files, every leaf compared with PyYAML"""
    import yaml
    from hypothesis import HealthCheck, given, settings, strategies as st
    exe = _build(tmp_path, "ros_params_dump")
    reserved = {"null", "true", "false", "yes", "no", "on", "off", "y", "n"}  # YAML 1.1 gives these a type of their own
    names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_", min_size=1, max_size=8).filter(lambda t: t not in reserved)
    floats = st.floats(allow_nan=False, allow_infinity=False, width=64).filter(lambda x: x == 0 or 1e-300 < abs(x) < 1e300)
    leaves = st.one_of(st.booleans(), st.integers(-10**9, 10**9), floats,
                       st.text(alphabet="abcdefghijklmnopqrstuvwxyz /#:-", min_size=1, max_size=12).filter(lambda t: t.strip() == t),
                       st.lists(floats, min_size=1, max_size=9))
    trees = st.recursive(st.dictionaries(names, leaves, min_size=1, max_size=4),
                         lambda inner: st.dictionaries(names, st.one_of(leaves, inner), min_size=1, max_size=4), max_leaves=12)
    path = tmp_path / "gen.yaml"

    def flat(d, prefix=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, prefix + k + ".")
            else:
                yield prefix + k, v

    @settings(max_examples=120, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
    @given(trees)
    def check(tree):
        text = "\n".join(_emit({"node": {"ros__parameters": tree}})) + "\n"
        # the emitter is sound (PyYAML follows YAML 1.1 and reads a float without a dot, `1e-05`, as a string; ROS2's
        # own reader and ours read a number)
        def num(y, v):
            if isinstance(v, float) and isinstance(y, str):
                return float(y)
            if isinstance(v, list) and isinstance(y, list):
                return [num(a, b) for a, b in zip(y, v)]
            return y
        truth = dict(flat({"node": {"ros__parameters": tree}}))
        loaded = dict(flat(yaml.safe_load(text)))
        assert {k: num(loaded[k], v) for k, v in truth.items()} == truth and loaded.keys() == truth.keys()
        path.write_text(text)
        r = subprocess.run([exe, "--raw", str(path)], capture_output=True, text=True)
        assert r.returncode == 0, (r.stderr, text)
        got = {}
        for line in r.stdout.splitlines():
            name, typ, *vals = line.split("\t")
            got[name] = {"bool": lambda: bool(int(vals[0])), "int": lambda: int(vals[0]), "double": lambda: float(vals[0]),
                         "string": lambda: vals[0], "doubles": lambda: [float(v) for v in vals]}[typ]()
        want = dict(flat({"node": {"ros__parameters": tree}}))
        assert got.keys() == want.keys(), text
        for k, v in want.items():
            assert got[k] == v and type(got[k]) is type(v) or (isinstance(v, list) and got[k] == [float(x) for x in v]), (k, v, got[k], text)

    check()


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference not mounted")
def test_reference_go1_file_gives_go1_params(tmp_path):
    p, node = load_ros_params(REF_YAML)
    assert _fields(p) == _fields(go1_params())
    assert node == {"log_name": "go1", "interval_ms": 5}
    got, extra = _cpp_dump(_build(tmp_path, "ros_params_dump"), REF_YAML)
    assert got == _fields(go1_params()) and extra["undeclared"] == []


def test_replay_program_compiles_and_links(tmp_path):
    exe = _build(tmp_path, "go1_nodes_replay")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def _quat_to_euler(q):
    w, x, y, z = q
    sinp = 2 * (w * y - z * x)
    return np.array([np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)),
                     np.copysign(np.pi / 2, sinp) if abs(sinp) >= 1 else np.arcsin(sinp),
                     np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))])


@pytest.mark.gpu
@pytest.mark.parametrize("est_type,polish", [(0, 0), (1, 0), (0, 1)])
def test_replayed_nodes_reproduce_oracle_and_log(tmp_path, est_type, polish):
    """polish = 1: osqp.polish as the node DECLARES it (EstSub.cpp:188; the Go1 parameter file turns it off) — the node core used to
    refuse that at construction"""
    exe = _build(tmp_path, "go1_nodes_replay")
    p = go1_params()
    p.ekf_rate = p.rate  # one orien_sub tick per est_sub tick in this replay
    p.est_type = est_type
    p.polish = polish
    K, GATE, TIME_INIT = 50, 9, 100.0  # est_sub starts on the tick after its 10th IMU message
    s = make_streams(p, 1, K)
    yaml_path = tmp_path / "params.yaml"
    dump_ros_params(p, yaml_path, log_name="replay")
    p_ib = np.array(list(p.p_ib))

    ev = []

    def rec(kind, clock, *payload):
        r = np.zeros(32)
        flat = np.concatenate([np.ravel(np.asarray(x, float)) for x in payload]) if payload else np.zeros(0)
        r[0], r[1], r[2:2 + len(flat)] = kind, clock, flat
        ev.append(r)

    for k in range(K):
        clock = TIME_INIT + s["imu_t"][k, 0]
        rec(1, clock, s["accel"][k, 0], s["gyro"][k, 0])
        rec(2, clock, s["q_joint"][k, 0], s["foot_force"][k, 0], s["qdot"][k, 0])
        if s["vo_mask"][k, 0]:
            rec(4, clock, TIME_INIT + s["vo_t_pre"][k, 0], TIME_INIT + s["vo_t_now"][k, 0], s["vo_dp"][k, 0])
            q = s["vo_q"][k, 0]
            rec(6, clock, TIME_INIT + s["vo_t_pose"][k, 0], q[1], q[2], q[3], q[0])
        rec(5, clock, s["gt_p"][k, 0], s["gt_v_s"][k, 0], s["gt_quat"][k, 0])
        rec(7, clock)
        rec(0, clock)
    np.stack(ev).tofile(tmp_path / "events.bin")

    # oracle: the EKF over every sample, the estimator from the gate on, fed what the node cores latch
    ekf = O.Ekf(p)
    est = O.Est(p)
    quats, want, want_pv = [], {}, {}
    for k in range(K):
        ekf.set_imu(s["imu_t"][k, 0], s["accel"][k, 0], s["gyro"][k, 0])
        if s["vo_mask"][k, 0]:
            ekf.set_vo(s["vo_t_pose"][k, 0], s["vo_q"][k, 0])
            est.set_vo(s["vo_t_pre"][k, 0], s["vo_t_now"][k, 0], s["vo_dp"][k, 0])
        ekf.step()
        quats.append(ekf.get()[0].copy())
        if k < GATE:
            continue
        est.set_imu(s["imu_t"][k, 0], s["accel"][k, 0], s["gyro"][k, 0])
        est.set_quat(quats[k])
        est.set_leg(s["p_foot"][k, 0] + p_ib, s["J"][k, 0], s["qdot"][k, 0], (s["foot_force"][k, 0] >= p.contact_effort_threshold).astype(float))
        T = k - GATE
        est.initialize() if T == 0 else est.update(T)
        x, vb, pv = est.get()
        want[T], want_pv[T] = np.concatenate([x, vb]), pv.copy()

    r = subprocess.run([exe, str(yaml_path), str(tmp_path / "events.bin"), str(TIME_INIT), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got_q = np.array([[float(v) for v in ln.split()[2:6]] for ln in r.stdout.splitlines() if ln.startswith("q ")])
    got_x = {int(ln.split()[1]): np.array([float(v) for v in ln.split()[2:14]]) for ln in r.stdout.splitlines() if ln.startswith("x ")}
    assert len(got_q) == K and sorted(got_x) == list(range(K - GATE))
    # stamps go through (t + time_init) - time_init: 1e-13 s of rounding, so not bit-equal to the oracle's EKF
    assert np.abs(got_q - np.array(quats)).max() < 1e-9

    def close(a, b):
        return all(np.abs(a[blk] - b[blk]).max() <= 1e-4 * np.abs(b[blk]).max() + 1e-6 for blk in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 12)))

    for T in range(1, K - GATE):
        assert close(got_x[T], want[T]), T

    # EstSub.cpp:75-84: variables are registered after tick N (discrete_time_ == N + 1), rows from the next tick on
    log = read_log("replay", str(tmp_path))
    first = p.N + 1
    rows = K - GATE - first
    assert [log[n].shape for n in ("pose", "GT_v", "v_body", "x_MHE", "p_vo_accmulate_", "filter_euler_", "gt_euler_")] == \
        [(rows, 3)] * 3 + [(rows, 9)] + [(rows, 3)] * 3
    for i in range(rows):
        T = first + i
        k = T + GATE
        assert np.array_equal(log["x_MHE"][i], got_x[T][:9]) and np.array_equal(log["v_body"][i], got_x[T][9:])
        assert np.abs(log["p_vo_accmulate_"][i] - want_pv[T]).max() <= 1e-9 + 1e-6 * np.abs(want_pv[T]).max()
        assert np.allclose(log["pose"][i], s["gt_p"][k, 0] - s["gt_p"][GATE, 0], atol=1e-15)
        Rq = s["gt_R"][k, 0]
        assert np.allclose(log["GT_v"][i], Rq @ s["gt_v_s"][k, 0], atol=1e-9)
        assert np.allclose(log["gt_euler_"][i], _quat_to_euler(s["gt_quat"][k, 0]), atol=1e-12)
        assert np.allclose(log["filter_euler_"][i], _quat_to_euler(got_q[k]), atol=1e-12)
