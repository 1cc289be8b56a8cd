import sys, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/tmp/exp')
from cvxpnpl_amd import synth
import collect
from newton import unpack, family_basis, zperp, solve_max_t
d=synth.make_pnpl(125000,10,0,2.0,seed=42)
idx=np.load("/tmp/exp/slow125_idx.npy")[:14]
sub={k:(d[k][idx] if k in ("pts_2d","pts_3d") else d[k]) for k in ("pts_2d","pts_3d","K")}
sub["line_2d"]=np.zeros((14,0,2,2)); sub["line_3d"]=np.zeros((14,0,2,3))
import ctypes as C
collect.L.dr_policy.argtypes=[C.c_int,C.c_double,C.c_int,C.c_double,C.c_int,C.c_int,C.c_int]
collect.L.dr_policy(0,0.005,2,2.0,0,0,1)
st,it,R,out=collect.run(sub,10,first_check=6,dual_refine=0)
print("iters", it.tolist(), "status", st.tolist(), "records", len(out))
np.savez("/tmp/exp/dump_slow125.npz", st=st,it=it,R=R,out=out)
for b in range(14):
    rec=out[out[:,0]==b]
    rec=rec[np.argsort(rec[:,1])]
    line=[]
    for r in rec:
        i=int(r[1]); ok=int(r[2]); delta=r[3]; S=unpack(r[4:59])-delta*np.eye(10); Rr=r[59:68]
        same=np.abs(Rr-R[b]).max()<1e-6
        z=np.concatenate([Rr.reshape(3,3).T.reshape(-1),[1.0]])
        P=zperp(z); ev=np.linalg.eigvalsh(P.T@S@P)
        if i<=14 and not ok and same:
            U=family_basis(z); tb,_=solve_max_t(S,U,P,iters=25)
        else: tb=float('nan')
        line.append(f"it{i}:{'ok' if ok else 'F'}{'' if same else '*'} lam[{ev[0]:.1e},{ev[1]:.1e},{ev[2]:.1e}] tmax {tb:.1e}")
    print(b, idx[b], " | ".join(line))
