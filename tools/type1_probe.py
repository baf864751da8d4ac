import sys, numpy as np, time
import torch; torch.cuda.init()
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as O
from decentralized_ekf_mhe_amd import go1_params, cassie_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_host, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
def run(p,B,K,label):
    s=make_streams(p,B,K)
    x,vb,q,secs,it=O.run_streams(p,s,nthreads=8,want_iters=True)
    est=BatchedEstimator(p,B); sh=streams_host(s)
    nb=p.dim_state//3
    worst=np.zeros(nb); same=0; tot=0
    for k in range(K):
        est.push_stream_step(sh,k); est.step(k)
        o=est.get(); info=est.solver_info()
        if k>0:
            d=np.abs(o["x"]-x[k]).reshape(B,nb,3).max(axis=2); ref=np.abs(x[k]).reshape(B,nb,3).max(axis=2)
            worst=np.maximum(worst,(d/(1e-4*ref+1e-6)).max(axis=0))
            same+=(info["iters"]==it[k]).sum(); tot+=B
            if not (o["status"]==(1 if p.est_type==0 else 0)).all(): print("status",k,o["status"]); 
    est.close()
    print(label,"worst err/tol",np.round(worst,3),"iters equal %d/%d"%(same,tot), flush=True)
p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1
run(p,8,75,"go1 type1 MHE")
p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1; p.est_type=1
run(p,8,40,"go1 type1 KF")
p=cassie_params(); p.ekf_rate=p.rate; p.leg_odom_type=1; p.N=8
run(p,4,30,"cassie type1 N=8 (lg)")
p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1; p.num_legs=1; p.N=30
run(p,4,50,"1 leg type1 N=30")
p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1; p.num_legs=3; p.N=5
run(p,4,20,"3 legs N=5")
# throughput
import torch
p=go1_params(); p.ekf_rate=p.rate; p.leg_odom_type=1
B,K=4096,80
s=make_streams(p,64,K)
big={k:(np.ascontiguousarray(np.tile(v,(1,B//64)+(1,)*(v.ndim-2))) if isinstance(v,np.ndarray) else v) for k,v in s.items()}
sd=streams_to_device(big); est=BatchedEstimator(p,B)
for k in range(60): est.push_stream_step(sd,k); est.step(k)
est.sync(); est.timing_enable(True); t=time.perf_counter()
for k in range(60,80): est.push_stream_step(sd,k); est.step(k)
est.sync(); el=time.perf_counter()-t
print("type1 B=4096: %.1f k steps/s"%(B*20/el/1e3), est.timing_read(), est.solver_info()["iters"].mean(), (est.get()["status"]==1).mean())
est.close()
