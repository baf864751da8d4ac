"""Experiment-log format: the C++ writer of the shim and the Python writer produce the
same bytes, in the layout of the reference's Data_Logger (name,type,len, lines; raw float64 / float32
rows), and the 27-double estimator row round-trips."""
import os
import subprocess

import numpy as np

from decentralized_ekf_mhe_amd.logger import ESTIMATOR_ROW, DataLogger, EstimatorLog, read_log

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CPP = r'''
#include "decentralized_ekf_mhe_amd/cpp/data_logger.hpp"
int main(int argc, char** argv) {
    double pose[3] = {1, 2, 3}, x[9], q[4] = {0.5, -0.5, 0.5, -0.5};
    float f[2] = {1.5f, -2.25f};
    int n = 7, iv[3] = {1, -2, 3};
    for (int i = 0; i < 9; ++i) x[i] = 0.1 * i;
    Data_Logger lg("cpp", argv[1]);
    lg.add_data_vectorXd(pose, 3, "pose");
    lg.add_data_vectorXd(x, 9, "x_MHE");
    lg.add_data_quaternion(q, "quat");
    lg.add_data_vectorXf(f, 2, "f");
    lg.add_data(&n, "count");
    lg.add_data_vectorXi(iv, 3, "iv");
    for (int t = 0; t < 3; ++t) {
        pose[0] = t; x[8] = -t; n = 7 + t;
        lg.spin_logging();
    }
    return 0;
}
'''


def test_cpp_and_python_writers_agree(tmp_path):
    src = tmp_path / "lg.cpp"
    src.write_text(CPP)
    exe = str(tmp_path / "lg")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-I" + ROOT, str(src), "-o", exe])
    subprocess.check_call([exe, str(tmp_path)])
    with DataLogger("py", str(tmp_path)) as lg:
        lg.add_data("pose", 3)
        lg.add_data("x_MHE", 9)
        lg.add_data("quat", 4, "Quaterniond")
        lg.add_data("f", 2, "VectorXf")
        lg.add_data("count", 1, "int")
        lg.add_data("iv", 3, "VectorXi")
        for t in range(3):
            x = 0.1 * np.arange(9)
            x[8] = -t
            lg.spin_logging({"pose": [t, 2, 3], "x_MHE": x, "quat": [0.5, -0.5, 0.5, -0.5], "f": [1.5, -2.25], "count": [7 + t], "iv": [1, -2, 3]})
    assert (tmp_path / "cpp_Name.csv").read_text() == (tmp_path / "py_Name.csv").read_text()
    assert (tmp_path / "cpp_Name.csv").read_text().splitlines()[0] == "pose,VectorXd,3,"
    a, b = (tmp_path / "cpp_Data").read_bytes(), (tmp_path / "py_Data").read_bytes()
    assert a == b and len(a) == 3 * (8 * (3 + 9 + 4) + 4 * (2 + 1 + 3))
    back = read_log("cpp", str(tmp_path))
    assert np.array_equal(back["count"][:, 0], [7, 8, 9]) and np.array_equal(back["pose"][:, 0], [0, 1, 2])


def test_estimator_row_is_27_doubles(tmp_path):
    assert sum(n for _, n in ESTIMATOR_ROW) == 27
    out = {"v_b": np.arange(6.0).reshape(2, 3), "x": np.arange(18.0).reshape(2, 9), "p_vo": np.ones((2, 3))}
    with EstimatorLog("run", str(tmp_path)) as lg:
        lg.log_instance(out, 1, gt_v_b=[9, 9, 9])
        lg.log_instance(out, 0)
    assert os.path.getsize(tmp_path / "run_Data") == 2 * 27 * 8
    back = read_log("run", str(tmp_path))
    assert np.array_equal(back["x_MHE"][0], np.arange(9.0, 18.0)) and np.array_equal(back["GT_v"][0], [9, 9, 9])
    assert np.array_equal(back["v_body"][1], [0, 1, 2])
