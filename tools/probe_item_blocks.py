import os, sys, time, json
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"prob-matrix-factorization_amd")]
import pmf_hip
from pmf_hip import ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER
from pmf_hip.synth import BASE_SEED, synth_ratings
K=64
for U,N in ((1_000_000,50_000_000),(500_000,25_000_000),(250_000,12_500_000),(125_000,6_250_000)):
    I=100_000
    u,i,r=synth_ratings(U,I,N,seed=BASE_SEED)
    ctx=pmf_hip.Context(U,I,K)
    rng=np.random.default_rng(1)
    ctx.set_ratings(u,i,r+1.0)
    ctx.set_array(USER,ARR_FACTOR,rng.gamma(1.0,0.3,(U,K))+0.1); ctx.set_array(ITEM,ARR_FACTOR,rng.gamma(1.0,0.3,(I,K))+0.1)
    ctx.set_array(USER,ARR_PRIOR_RATE,np.full(U,1.0)); ctx.set_array(ITEM,ARR_PRIOR_RATE,np.full(I,1.0))
    up=(0.3,0.0,True,0.3+K*0.3,1.0)
    for _ in range(2): ctx.gamma_sweep(USER,*up); ctx.gamma_sweep(ITEM,*up)
    ci=ctx.gather_ceiling_ms(ITEM,5); cu=ctx.gather_ceiling_ms(USER,5)
    print(json.dumps({"U":U,"N":N,"table_MB":U*K*4/1e6,"item_side_probe_ms":ci,"ns_per_Mgather":ci/N*1e9,"TBps":N*(4*K+8)/ci/1e9,"user_side_probe_ms":cu,"user_TBps":N*(4*K+8)/cu/1e9}),flush=True)
    ctx.close()
