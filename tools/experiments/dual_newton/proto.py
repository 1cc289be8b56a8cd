import numpy as np, sys
from newton import unpack, AM
from basis_tab import TAB   # list of 15 matrices (upper entries): [(p,q,coef)], last = T
zI=np.array([1,0,0,0,1,0,0,0,1,1.0])
def dense(tab):
    M=np.zeros((10,10))
    for p,q,c in tab:
        M[p,q]+=c
        if p!=q: M[q,p]+=c
    return M
UD=[dense(t) for t in TAB]
def PR(R):
    P=np.zeros((10,10))
    for j in range(3): P[3*j:3*j+3,3*j:3*j+3]=R
    P[9,9]=1; return P
def hess_table(X):
    n=len(TAB); H=np.zeros((n,n)); g=np.zeros(n)
    for a in range(n):
        g[a]=sum(c*(1 if p==q else 2)*X[p,q] for p,q,c in TAB[a])
        for b in range(a,n):
            h=0.0
            for p,q,ca in TAB[a]:
                wa=1 if p==q else 2
                for r,s,cb in TAB[b]:
                    wb=1 if r==s else 2
                    h+=ca*cb*wa*wb*0.5*(X[p,r]*X[q,s]+X[p,s]*X[q,r])
            H[a,b]=H[b,a]=h
    return g,H
def newton_dual(S1, R, delta, K=4, c=0.3, kappa=1.0, verbose=False):
    """returns number of Newton steps used (>=1) or -1; S1 world frame (no delta)"""
    P=PR(R); Sp=P.T@S1@P
    v=np.zeros(15)  # last = t (coefficient of -T => stored as v[14] multiplies UD[14] = -T)
    lam=np.linalg.eigvalsh(Sp+kappa*np.outer(zI,zI)/4)[0]   # device: from failed LDL? use estimate
    t=lam-c*abs(lam)
    def Shat(v,t): return Sp+sum(v[a]*UD[a] for a in range(14))-t*UD[14]+kappa*np.outer(zI,zI)/4
    def feasible(M):
        try: np.linalg.cholesky(M); return True
        except np.linalg.LinAlgError: return False
    M=Shat(v,t)
    if not feasible(M): return -2
    for k in range(K):
        X=np.linalg.inv(M)
        g,H=hess_table(X)
        # check vs dense
        if verbose:
            Hd=np.array([[np.trace(X@UD[a]@X@UD[b]) for b in range(15)] for a in range(15)]); print("H err", np.abs(H-Hd).max()/np.abs(Hd).max())
        grad=-g.copy(); grad[14]=0.0          # d/dv of -logdet = -tr(X U_a); t-component: +tr(X T) - 1/mu = 0 by the mu rule
        # note basis 14 is T with Shat = ... - t T: derivative wrt t of -logdet(Shat) = +tr(X T); H_tt etc need sign: direction matrix for t is -T
        sgn=np.ones(15); sgn[14]=-1.0
        Hs=H*np.outer(sgn,sgn)
        dx=-np.linalg.solve(Hs,grad)
        a=1.0; ok=False
        for ls in range(5):
            vn=v.copy(); vn[:14]+=a*dx[:14]; tn=t+a*dx[14]
            Mn=Shat(vn,tn)
            if feasible(Mn): ok=True; break
            a*=0.5
        if not ok: return -1
        v,t,M=vn,tn,Mn
        Sv=Sp+sum(v[a_]*UD[a_] for a_ in range(14))
        if feasible(Sv+delta*np.eye(10)+0*np.outer(zI,zI)) or np.linalg.eigvalsh(Sv+np.outer(zI,zI))[0]>-delta: return k+1
    return -1
if __name__=="__main__":
    d=np.load(sys.argv[1]); out=d["out"]; Rf=d["R"]
    recs=[r for r in out if r[2]==0 and np.abs(r[59:68]-Rf[int(r[0])]).max()<1e-6]
    res=[]
    for i,r in enumerate(recs):
        delta=r[3]; S=unpack(r[4:59])-delta*np.eye(10); R=r[59:68].reshape(3,3)
        res.append(newton_dual(S,R,delta,verbose=(i==0)))
    res=np.array(res); print("records",len(res),"steps hist",{int(k):int(v) for k,v in zip(*np.unique(res,return_counts=True))})
