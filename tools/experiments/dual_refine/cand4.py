import numpy as np, sys
from newton import unpack, family_basis, zperp
d = np.load(sys.argv[1]); out = d["out"]; Rfin = d["R"]
fail = out[out[:, 2] == 0]
def lam_min(M): return np.linalg.eigvalsh(M)[0]
stats = {}
def rec(name, it, ok): stats.setdefault(name, {}).setdefault(it, []).append(ok)
for r in fail:
    b = int(r[0]); it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
    W = unpack(r[113:168])
    z = np.concatenate([R.T.reshape(-1), [1.0]])
    U = family_basis(z)
    def PU(E): return sum(np.tensordot(u, E) * u for u in U)
    def pz(x): return x - (z @ x) / 4 * z
    def test(Sn): return lam_min(Sn + np.outer(z, z)) > -delta
    ww, V = np.linalg.eigh(W); v2 = V[:, -2]
    D = PU(np.eye(10) - np.outer(z, z) / 4)
    okD = any(test(S + m * D) for m in (0.015, 0.015 / 4))
    def grad(Sc, x0, sig, nit, k):
        B = Sc + sig * np.eye(10)
        if lam_min(B + np.outer(z, z)) <= 0: return False, Sc, x0
        x = pz(x0)
        for _ in range(nit):
            x = pz(np.linalg.solve(B, x)); x /= np.linalg.norm(x)
        ray = x @ Sc @ x
        if ray >= 0: return False, Sc, x
        E = np.outer(x, x); G = PU(E); g2 = np.tensordot(G, E)
        Sn = Sc + k * (abs(ray)) / g2 * G
        return test(Sn), Sn, x
    for sig in (0.01, 0.005):
        for nit in (1, 2, 3):
            ok, Sn, x = grad(S, v2, sig, nit, 2.0)
            rec(f"grad sig={sig} nit={nit}", it, ok)
            if nit == 2:
                ok2 = ok
                if not ok:
                    ok2, _, _ = grad(Sn, x, sig, 1, 2.0)
                rec(f"grad sig={sig} nit=2 + 2nd step (1 it from x)", it, ok2)
                rec(f"Dshift, then grad sig={sig} nit=2", it, okD or ok)
                rec(f"grad sig={sig} nit=2, then Dshift", it, ok or okD)
                okk = ok or grad(S, v2, sig, 2, 4.0)[0] or grad(S, v2, sig, 2, 1.4)[0]
                rec(f"grad sig={sig} nit=2 ladder k=2,4,1.4", it, okk)
for name in stats:
    tot = sum(len(v) for v in stats[name].values()); okc = sum(sum(v) for v in stats[name].values())
    print(f"{name:50s} {okc}/{tot} = {okc/tot:.3f}   by it: " + "  ".join(f"{it}:{sum(v)}/{len(v)}" for it, v in sorted(stats[name].items())))
