import numpy as np, sys
from newton import unpack, family_basis, zperp
d = np.load(sys.argv[1]); out = d["out"]; Rfin = d["R"]
fail = out[out[:, 2] == 0]
def lam_min(M): return np.linalg.eigvalsh(M)[0]
stats = {}
def rec(name, it, ok): stats.setdefault(name, {}).setdefault(it, []).append(ok)
cosv = []
for r in fail:
    b = int(r[0]); it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
    W = unpack(r[113:168]); Wp = unpack(r[168:223])
    z = np.concatenate([R.T.reshape(-1), [1.0]])
    U = family_basis(z)
    def PU(E): return sum(np.tensordot(u, E) * u for u in U)
    Pz = np.eye(10) - np.outer(z, z) / 4
    w, Q = np.linalg.eigh(S + np.outer(z, z)); nvec = Q[:, 0]
    def test(Sn): return lam_min(Sn + np.outer(z, z)) > -delta
    def step(x, k=2.0):
        x = Pz @ x; x = x / np.linalg.norm(x)
        ray = x @ S @ x
        if ray >= 0: return False
        G = PU(np.outer(x, x)); g2 = np.tensordot(G, np.outer(x, x))
        return test(S + k * abs(ray) / g2 * G)
    ww, V = np.linalg.eigh(W)  # ascending
    # candidates: runner-up of W (second largest), smallest |lambda|
    v2 = V[:, -2]
    j0 = np.argmin(np.abs(ww)); v0 = V[:, j0]
    cosv.append((abs((Pz@v2/np.linalg.norm(Pz@v2)) @ nvec), abs((Pz@v0/np.linalg.norm(Pz@v0)) @ nvec), ww[-2], ww[j0]))
    rec("n = runner-up eigvec of W", it, step(v2))
    rec("n = eigvec of W with smallest |lam|", it, step(v0))
    # best Rayleigh among all eigenvectors of W
    rays = []
    for j in range(10):
        x = Pz @ V[:, j]; nx = np.linalg.norm(x)
        rays.append((x @ S @ x) / nx**2 if nx > 0.3 else 1e9)
    jb = int(np.argmin(rays))
    rec("n = eigvec of W with best Rayleigh", it, step(V[:, jb]))
    # one inverse iteration from these with sigma = 0.01
    B = S + 0.01 * np.eye(10) + np.outer(z, z)
    if lam_min(B) > 0:
        for nm, x0 in (("runner-up", v2), ("best-ray", V[:, jb])):
            x = np.linalg.solve(B, Pz @ x0)
            rec(f"invit(0.01) from {nm}", it, step(x))
            x = np.linalg.solve(B, Pz @ x)
            rec(f"invit(0.01)x2 from {nm}", it, step(x))
    # subspace: Rayleigh-Ritz in the span of the 2-3 eigvecs of W with smallest |lam| (excluding top)
    idx = np.argsort(np.abs(ww))[:3]
    Bz = Pz @ V[:, idx]; qb, _ = np.linalg.qr(Bz)
    wr, Vr = np.linalg.eigh(qb.T @ S @ qb)
    rec("Ritz in 3 smallest-|lam| eigvecs of W", it, step(qb @ Vr[:, 0]))
    rec("exact", it, step(nvec))
for name in stats:
    tot = sum(len(v) for v in stats[name].values()); okc = sum(sum(v) for v in stats[name].values())
    print(f"{name:44s} {okc}/{tot} = {okc/tot:.3f}   by it: " + "  ".join(f"{it}:{sum(v)}/{len(v)}" for it, v in sorted(stats[name].items())))
cosv = np.array(cosv)
print("median |cos(n, v2)|", np.median(cosv[:, 0]), " |cos(n, v_small)|", np.median(cosv[:, 1]), "  lam2(W) median", np.median(cosv[:, 2]), " smallest |lam| median", np.median(np.abs(cosv[:, 3])))
print("quantiles cos v2", np.quantile(cosv[:, 0], [.1, .25, .5, .75, .9]))
print("quantiles cos vsmall", np.quantile(cosv[:, 1], [.1, .25, .5, .75, .9]))
