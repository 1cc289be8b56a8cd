import numpy as np, sys
import proto
from newton import unpack, zperp, family_basis, solve_max_t
d=np.load(sys.argv[1]); out=d["out"]; Rf=d["R"]
recs=[r for r in out if r[2]==0 and np.abs(r[59:68]-Rf[int(r[0])]).max()<1e-6]
n=0
for r in recs:
    delta=r[3]; S=unpack(r[4:59])-delta*np.eye(10); R=r[59:68].reshape(3,3)
    k=proto.newton_dual(S,R,delta,K=4)
    if k<0:
        k8=proto.newton_dual(S,R,delta,K=10)
        z=np.concatenate([R.T.reshape(-1),[1.0]]); P=zperp(z); ev=np.linalg.eigvalsh(P.T@S@P)
        tb,_=solve_max_t(S,family_basis(z),P,iters=25)
        print("it",int(r[1]),"code",k,"K=10:",k8,"lam",ev[:3],"tmax",tb)
        n+=1
        if n>=12: break
