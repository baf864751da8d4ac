#!/usr/bin/env python3
"""Prints the solve kernel's launch of a Go1 handle (workgroups, CUs): which residency a library variant (DEKF_LIB) really gets."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator
est = BatchedEstimator(go1_params(), int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
print(json.dumps({"lib": os.environ.get("DEKF_LIB", "libdekf.so"), **est.launch_info()}))
est.close()
