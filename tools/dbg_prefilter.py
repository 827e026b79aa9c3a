import numpy as np, torch, sys
sys.path.insert(0,'.')
from emgraph_amd import device as D, _lib as L
from emgraph_amd.evaluation import ranking as RK
rs=np.random.RandomState(0)
n_ent,k=9000,200; ki=400; sc=float(np.float32(2/200))
E=(rs.randn(n_ent,ki)*0.3).astype(np.float32); R=(rs.randn(5,ki)*0.3).astype(np.float32)
T=np.stack([rs.randint(0,n_ent,200),rs.randint(0,5,200),rs.randint(0,n_ent,200)],1).astype(np.int32)
Et,Rt=torch.from_numpy(E).cuda(),torch.from_numpy(R).cuda()
Q,pos=D.eval_build_queries(4,Et,Rt,ki,sc,torch.from_numpy(T).cuda(),3)
kp=D.bf16_ld(ki); Eb=D.to_f16(Et,ki,ld_dst=kp); Qb=D.to_f16(Q,ki,ld_dst=kp)
b=RK.table_norm_bounds(Et,Eb,ki); band=RK.prefilter_band(Q,Qb,ki,b)
print('bounds',b,'band',band[:4], 'acc std', float((Q[:, :ki]@Et.T).std()), 'Qb max', float(Qb.float().abs().max()), 'inf?', bool(torch.isinf(Qb.float()).any()))
n_seg=D.eval_prefilter_segments(400,n_ent)
pairs,pc=RK._pair_buffer(Et.device, n_seg)
cnt=torch.zeros((2,400),dtype=torch.int32,device='cuda')
try:
    D.eval_prefilter_f16(4,Qb,pos,band,Eb,0,ki,sc,cnt[0],pairs,pc)
    torch.cuda.synchronize(); p=pc.cpu().numpy(); print('n_seg',n_seg,'cap',pairs.numel()//n_seg,'max',p[:n_seg].max(),'flag',p[n_seg],'total',p[:n_seg].sum())
except Exception as e: print('ERR',e)
