import numpy as np, sys
from newton import unpack, family_basis, zperp
d=np.load("/tmp/exp/dump_slow125.npz"); out=d["out"]; Rf=d["R"]
recs=[r for r in out if r[2]==0 and r[1]<=14 and np.abs(r[59:68]-Rf[int(r[0])]).max()<1e-6]
print(len(recs),"records")
def prep(r):
    delta=r[3]; S=unpack(r[4:59])-delta*np.eye(10); R=r[59:68].reshape(3,3)
    z=np.concatenate([R.T.reshape(-1),[1.0]]); P=zperp(z); U=family_basis(z)
    M0=P.T@S@P; Mk=np.stack([P.T@u@P for u in U]); return delta,M0,Mk
def eiggrad(M0,Mk,delta,K,gain=2.0):
    v=np.zeros(len(Mk))
    for k in range(K):
        M=M0+np.tensordot(v,Mk,1); w,Q=np.linalg.eigh(M)
        if w[0]>-delta: return k
        n=Q[:,0]; g=np.array([n@m@n for m in Mk]); v=v+gain*abs(w[0])/(g@g)*g
    M=M0+np.tensordot(v,Mk,1)
    return K if np.linalg.eigvalsh(M)[0]>-delta else -1
def barrier(M0,Mk,delta,K,c=1.0,mu=None,tstep=True):
    nv=len(Mk); v=np.zeros(nv); lam=np.linalg.eigvalsh(M0)[0]; t=lam-c*abs(lam)
    for k in range(K):
        M=M0+np.tensordot(v,Mk,1)-t*np.eye(9)
        X=np.linalg.inv(M); XM=np.stack([X@m for m in Mk]+[-X])
        m_=mu if mu else 1.0/np.trace(X)   # mu so that dt direction balanced
        g=-np.array([np.trace(a) for a in XM]); g[-1]-=1.0/m_
        H=np.einsum('aij,bji->ab',XM,XM)
        dx=-np.linalg.solve(H,g)
        a=1.0
        for ls in range(8):
            vn=v+a*dx[:nv]; tn=t+a*dx[nv]
            lmn=np.linalg.eigvalsh(M0+np.tensordot(vn,Mk,1))[0]
            if lmn-tn>0: break
            a*=0.5
        v,t=vn,tn
        if lmn>-delta: return k+1
    return -1
def center(M0,Mk,delta,K,c=1.0):
    """analytic centre of {v: M(v) >= t0 I} with t0 fixed below lam_min: Newton steps; check lam_min after each"""
    nv=len(Mk); v=np.zeros(nv); lam=np.linalg.eigvalsh(M0)[0]; t=lam-c*abs(lam)
    for k in range(K):
        M=M0+np.tensordot(v,Mk,1)-t*np.eye(9); X=np.linalg.inv(M); XM=np.stack([X@m for m in Mk])
        g=-np.array([np.trace(a) for a in XM]); H=np.einsum('aij,bji->ab',XM,XM); dx=-np.linalg.solve(H,g)
        a=1.0
        for ls in range(8):
            vn=v+a*dx; lmn=np.linalg.eigvalsh(M0+np.tensordot(vn,Mk,1))[0]
            if lmn-t>0: break
            a*=0.5
        v=vn
        if lmn>-delta: return k+1
        # raise t toward new lam_min
        t=lmn-c*abs(lmn) if lmn<0 else t
    return -1
P=[prep(r) for r in recs]
for name,f in (("eiggrad gain2",lambda p:eiggrad(p[1],p[2],p[0],30)),("eiggrad gain1.5",lambda p:eiggrad(p[1],p[2],p[0],30,1.5)),("eiggrad gain3",lambda p:eiggrad(p[1],p[2],p[0],30,3.0)),
               ("barrier c=1",lambda p:barrier(p[1],p[2],p[0],12)),("barrier c=3",lambda p:barrier(p[1],p[2],p[0],12,3.0)),("barrier c=0.3",lambda p:barrier(p[1],p[2],p[0],12,0.3)),
               ("center c=1",lambda p:center(p[1],p[2],p[0],12)),("center c=3",lambda p:center(p[1],p[2],p[0],12,3.0)),("center c=0.3",lambda p:center(p[1],p[2],p[0],12,0.3))):
    res=[f(p) for p in P]
    print(f"{name:18s} steps to certify:", res)
