import sys, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import hostsim
from cvxpnpl_amd import synth
d=synth.make_pnpl(125000,10,0,2.0,seed=42)
r=hostsim.solve_batch(d["pts_2d"],d["pts_3d"],None,None,d["K"],opts=hostsim.default_opts(first_check=6))
it=r["iters"]; 
print("hist", {int(k):int(v) for k,v in zip(*np.unique(it,return_counts=True))})
idx=np.argsort(-it)[:40]; print(idx.tolist(), it[idx].tolist())
np.save("/tmp/exp/slow125_idx.npy", idx)
