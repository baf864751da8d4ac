import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
B,K=4096,45
for form in (0,1):
    p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1; p.arrival_cost_form=form
    s=make_streams(p,64,K)
    big={k:(np.ascontiguousarray(np.tile(v,(1,B//64)+(1,)*(v.ndim-2))) if isinstance(v,np.ndarray) else v) for k,v in s.items()}
    sd=streams_to_device(big); est=BatchedEstimator(p,B)
    for k in range(K):
        if k==30: est.sync(); est.timing_enable(True)
        est.push_stream_step(sd,k); est.step(k)
    t=est.timing_read(); print("form",form,{k:round(v[0]/max(v[1],1),4) for k,v in t.items()}); est.close()
