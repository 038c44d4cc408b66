import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg,'GM',seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
for B,N in ((1,1024),(1,512),(4,1024)):
    pair = synthetic.make_correlated_pair(N,N,seed=1,batch=B)
    d = {k: torch.from_numpy(v).cuda() for k,v in pair.items() if k!='image_shape'}
    d['image0']=d['image1']=torch.zeros(pair['image_shape'],device='cuda')
    m.produce_matches(d,p=0.2,only_last=True)
    print(B,N,'sinkhorn us/iter', round(ctx.time_sinkhorn(B,N,50)*1e3,2), ctx.resident_status(), flush=True)
