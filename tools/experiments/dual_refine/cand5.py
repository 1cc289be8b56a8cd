import numpy as np, sys
from newton import unpack, family_basis
d = np.load(sys.argv[1]); out = d["out"]; Rfin = d["R"]
fail = out[out[:, 2] == 0]
def lam_min(M): return np.linalg.eigvalsh(M)[0]
stats = {}
def rec(name, it, ok): stats.setdefault(name, {}).setdefault(it, []).append(ok)
def newton_cg(S, PU, z, sig, ncg, damp=True):
    B = S + sig * np.eye(10)
    if lam_min(B + np.outer(z, z)) <= 0: return None
    X = np.linalg.inv(B)
    G = PU(X)                      # -gradient of psi = -logdet: d/dv logdet = <X, U_k>
    # solve H[V] = G with H[V] = PU(X V X) by CG
    V = np.zeros((10, 10)); r = G.copy(); p = r.copy(); rr = np.tensordot(r, r)
    for i in range(ncg):
        Hp = PU(X @ p @ X)
        a = rr / np.tensordot(p, Hp)
        V = V + a * p; r = r - a * Hp
        rr2 = np.tensordot(r, r)
        if rr2 < 1e-24: break
        p = r + (rr2 / rr) * p; rr = rr2
    if damp:
        dec = np.sqrt(max(np.tensordot(V, PU(X @ V @ X)), 0))   # Newton decrement of the (truncated) step
        if dec > 0.25: V = V / (1 + dec)
    return V
for r in fail:
    b = int(r[0]); it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
    z = np.concatenate([R.T.reshape(-1), [1.0]])
    U = family_basis(z)
    def PU(E): return sum(np.tensordot(u, E) * u for u in U)
    def test(Sn): return lam_min(Sn + np.outer(z, z)) > -delta
    lam1 = lam_min(S + np.outer(z, z))
    for signame, sigs in (("5e-3", [5e-3]), ("2e-3", [2e-3, 8e-3]), ("1e-3", [1e-3, 4e-3, 1.6e-2]), ("3|lam1|", [3 * abs(lam1)])):
        for ncg in (1, 2, 3, 14):
            for damp in (True, False):
                ok = False; Sc = S
                for K in (1, 2):
                    V = None
                    for sg in sigs:
                        V = newton_cg(Sc, PU, z, sg, ncg, damp)
                        if V is not None: break
                    if V is None: break
                    Sc = Sc + V
                    ok = ok or test(Sc)
                    rec(f"sig={signame:8s} ncg={ncg:2d} damp={int(damp)} K={K}", it, ok)
names = sorted(stats)
for name in names:
    tot = sum(len(v) for v in stats[name].values()); okc = sum(sum(v) for v in stats[name].values())
    print(f"{name:40s} {okc}/{tot} = {okc/tot:.3f}   by it: " + "  ".join(f"{it}:{sum(v)}/{len(v)}" for it, v in sorted(stats[name].items())))
