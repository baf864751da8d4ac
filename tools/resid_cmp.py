import sys, os, ctypes as C, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params, capi
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
p=go1_params(); p.ekf_rate=p.rate
B,K=1000,p.N
s=make_streams(p,B,K); sd=streams_to_device(s)
res=[]
for cap in (0,2):
    q=p.copy(); q.solve_workgroups_per_cu=cap
    e=BatchedEstimator(q,B)
    for k in range(K):
        e.push_stream_step(sd,k); e.step(k)
    e.sync(); lib=capi.load(); out=np.zeros((B,32)); lib.dekf_debug_sections.argtypes=[C.c_void_p,C.c_void_p]
    capi.check(lib.dekf_debug_sections(e.h, C.c_void_p(out.ctypes.data))); res.append(out[:,16:30].copy()); print(e.solve_kernel_name(True)); e.close()
a,b=res
names=["pri/E","z/E","Ax/E","pri","z","Ax","dua/D","q/D","Aty/D","Px/D","dua","q","Aty","Px"]
for i,n in enumerate(names):
    d=np.abs(a[:,i]-b[:,i]); print(n, "max rel diff %.3e"%(d/np.maximum(np.abs(b[:,i]),1e-300)).max(), "equal frac", (d==0).mean(), a[0,i], b[0,i])
