#!/usr/bin/env python3
"""Times k_mhe_assemble (and k_ekf_tick) of the Go1 benchmark shape with a one-iteration solve behind it: harness for the
early-exit builds of the assemble (-DDEKF_ASM_STOP=k, numerics garbage on purpose) that price its phases.
    DEKF_LIB=.../libdekf_asmK.so python tools/probes/asm_phase_probe.py [batch]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = go1_params()
p.ekf_rate = p.rate
p.max_qp_iter, p.adapt_rho, p.check_termination = 1, 0, 100000
W, steps = p.N + 6, 60
s = make_streams(p, B, W + steps)
sd = streams_to_device(s)
est = BatchedEstimator(p, B)
for k in range(W):
    est.push_stream_step(sd, k); est.step(k)
est.sync()
est.timing_enable(True)
for k in range(W, W + steps):
    est.push_stream_step(sd, k); est.step(k)
est.sync(); torch.cuda.synchronize()
t = est.timing_read()
print(json.dumps({"lib": os.path.basename(os.environ.get("DEKF_LIB", "product")), "batch": B,
                  "assemble_us": round(1e3 * t["assemble"][0] / max(t["assemble"][1], 1), 2), "ekf_us": round(1e3 * t["ekf"][0] / max(t["ekf"][1], 1), 2)}))
est.close()
