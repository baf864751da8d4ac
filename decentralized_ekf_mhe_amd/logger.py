"""The reference's experiment-log format for batched runs.

``<name>_Name.csv`` lists ``name,type,len,`` per variable; ``<name>_Data`` is the raw concatenation,
tick after tick, of the variables in registration order (float64 for VectorXd / Quaterniond, float32
for VectorXf / int / VectorXi) — see decentralized_ekf_mhe_amd/cpp/data_logger.hpp for the citations.
``EstimatorLog`` writes the 27-double row of the reference's estimator node
(src/decentral_legged_est/src/EstSub.cpp:99-106) for one instance of a batch per file.
"""
import os

import numpy as np

_KINDS = {"double": "<f8", "VectorXd": "<f8", "Quaterniond": "<f8", "VectorXf": "<f4", "int": "<f4", "VectorXi": "<f4"}


class DataLogger:
    def __init__(self, file_name, file_location):
        os.makedirs(file_location, exist_ok=True)
        self._data = open(os.path.join(file_location, file_name + "_Data"), "wb")
        self._names = open(os.path.join(file_location, file_name + "_Name.csv"), "w")
        self._vars = []

    def add_data(self, name, length, kind="VectorXd"):
        if kind not in _KINDS:
            raise ValueError(f"unknown type {kind}")
        self._vars.append((name, int(length), kind))
        self._names.write(f"{name},{kind},{int(length)},\n")
        self._names.flush()

    def spin_logging(self, values):
        """values: dict name -> array of the registered length"""
        for name, length, kind in self._vars:
            v = np.asarray(values[name]).reshape(-1)
            if v.size != length:
                raise ValueError(f"{name}: expected {length} values, got {v.size}")
            self._data.write(v.astype(_KINDS[kind]).tobytes())

    def done_logging(self):
        self._data.close()
        self._names.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.done_logging()


ESTIMATOR_ROW = [("pose", 3), ("GT_v", 3), ("v_body", 3), ("x_MHE", 9), ("p_vo_accmulate_", 3), ("filter_euler_", 3), ("gt_euler_", 3)]


class EstimatorLog(DataLogger):
    """the estimator node's row: pose GT_v v_body x_MHE p_vo_accmulate_ filter_euler_ gt_euler_ (27 doubles)"""

    def __init__(self, file_name, file_location):
        super().__init__(file_name, file_location)
        for name, n in ESTIMATOR_ROW:
            self.add_data(name, n)

    def log_instance(self, out, b, gt_p=None, gt_v_b=None, filter_euler=None, gt_euler=None):
        """out = BatchedEstimator.get(); b = instance index; ground-truth fields default to zero"""
        z3 = np.zeros(3)
        self.spin_logging({"pose": z3 if gt_p is None else gt_p, "GT_v": z3 if gt_v_b is None else gt_v_b,
                           "v_body": out["v_b"][b], "x_MHE": out["x"][b], "p_vo_accmulate_": out["p_vo"][b],
                           "filter_euler_": z3 if filter_euler is None else filter_euler,
                           "gt_euler_": z3 if gt_euler is None else gt_euler})


def read_log(file_name, file_location):
    """parse a log pair back into {name: array[ticks, len]} (what the authors' plotting scripts do)"""
    cols = []
    with open(os.path.join(file_location, file_name + "_Name.csv")) as f:
        for line in f:
            parts = line.strip().split(",")
            if len(parts) >= 3 and parts[0]:
                cols.append((parts[0], parts[1], int(parts[2])))
    raw = open(os.path.join(file_location, file_name + "_Data"), "rb").read()
    row = sum(n * np.dtype(_KINDS[k]).itemsize for _, k, n in cols)
    if row == 0 or len(raw) % row:
        raise ValueError("data file does not hold a whole number of rows")
    ticks = len(raw) // row
    out = {name: np.zeros((ticks, n)) for name, _, n in cols}
    off = 0
    for t in range(ticks):
        for name, k, n in cols:
            dt = np.dtype(_KINDS[k])
            out[name][t] = np.frombuffer(raw, dtype=dt, count=n, offset=off)
            off += n * dt.itemsize
    return out
