import numpy as np, sys
from scipy.optimize import minimize
from newton import unpack, family_basis, zperp
def tmax(M0, Mk):
    v = np.zeros(len(Mk)); best = -1e9
    for beta in (50, 200, 1000, 5000, 20000):
        def f(v):
            M = M0 + np.tensordot(v, Mk, 1)
            w, Q = np.linalg.eigh(M)
            a = -beta * (w - w[0]); e = np.exp(a); s = e.sum()
            val = -(w[0] - np.log(s) / beta)
            p = e / s
            G = (Q * p) @ Q.T
            g = -np.einsum('kij,ij->k', Mk, G)
            return val, g
        r = minimize(f, v, jac=True, method='BFGS', options={'maxiter': 300, 'gtol': 1e-9})
        v = r.x
        best = max(best, np.linalg.eigvalsh(M0 + np.tensordot(v, Mk, 1))[0])
    return best, v
if __name__ == "__main__":
    d = np.load(sys.argv[1]); out = d["out"]
    fail = out[out[:, 2] == 0]
    rs = np.random.RandomState(0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    sel = fail if len(fail) <= n else fail[rs.choice(len(fail), n, replace=False)]
    res = []
    for r in sel:
        it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
        z = np.concatenate([R.T.reshape(-1), [1.0]])
        assert np.abs(S @ z).max() < 1e-9, np.abs(S @ z).max()
        U = family_basis(z); assert U.shape[0] == 14
        P = zperp(z)
        M0 = P.T @ S @ P; Mk = np.stack([P.T @ u @ P for u in U])
        lam0 = np.linalg.eigvalsh(M0)
        tb, v = tmax(M0, Mk)
        res.append((r[0], it, lam0[0], lam0[1], lam0[2], tb, np.linalg.norm(v)))
    res = np.array(res)
    print(" b    it  lam1(S1)    lam2       lam3      t_max      |v|")
    for row in res[:50]: print("%5d %2d  %+.3e  %+.3e  %+.3e  %+.3e  %.3e" % tuple(row))
    print("fraction with t_max > 0:", (res[:, 5] > 0).mean(), " min t_max", res[:, 5].min(), "median t_max", np.median(res[:, 5]), "median lam_min", np.median(res[:, 2]))
    for it in np.unique(res[:, 1]):
        m = res[:, 1] == it
        print("it", it, "n", m.sum(), "t_max>0", (res[m, 5] > 0).mean(), "median tmax", np.median(res[m, 5]), "min tmax", res[m,5].min(), "median lam_min", np.median(res[m, 2]))
    np.save("/tmp/exp/tmax.npy", res)
