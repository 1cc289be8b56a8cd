import numpy as np, sys
from newton import unpack, family_basis, zperp
d = np.load(sys.argv[1]); out = d["out"]; Rfin = d["R"]; itfin = d["it"]
fail = out[out[:, 2] == 0]
def lam_min(M): return np.linalg.eigvalsh(M)[0]
def ldl(M):
    """LDL^T no pivoting of symmetric M; returns L, d"""
    n = len(M); A = M.copy(); L = np.eye(n); dd = np.zeros(n)
    for k in range(n):
        dd[k] = A[k, k]
        L[k+1:, k] = A[k+1:, k] / dd[k]
        A[k+1:, k+1:] -= np.outer(L[k+1:, k], L[k+1:, k]) * dd[k]
    return L, dd
stats = {}
def rec(name, it, ok): stats.setdefault(name, {}).setdefault(it, []).append(ok)
posewrong = 0
for r in fail:
    b = int(r[0]); it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
    same = np.abs(R.reshape(-1) - Rfin[b]).max() < 1e-6
    z = np.concatenate([R.T.reshape(-1), [1.0]])
    U = family_basis(z)
    def PU(E): return sum(np.tensordot(u, E) * u for u in U)
    Pz = np.eye(10) - np.outer(z, z) / 4
    w, Q = np.linalg.eigh(S + np.outer(z, z))  # lift z direction to 4
    nvec = Q[:, 0]; lam1 = w[0]
    def test(Sn): return lam_min(Sn + np.outer(z, z)) > -delta
    G = PU(np.outer(nvec, nvec)); g2 = np.tensordot(G, np.outer(nvec, nvec))
    for k in (1.25, 1.5, 2, 3):
        rec(f"exact n, k={k}", it, test(S + k * abs(lam1) / g2 * G))
    rec("exact n, ladder 2,4,1.25", it, any(test(S + k * abs(lam1) / g2 * G) for k in (2, 4, 1.25)))
    okA = test(S + 2 * abs(lam1) / g2 * G)
    rec("exact n k=2 | pose==final", it, okA) if same else rec("exact n k=2 | pose!=final", it, okA)
    # two successive gradient steps
    S2 = S + 2 * abs(lam1) / g2 * G
    if not test(S2):
        w2, Q2 = np.linalg.eigh(S2 + np.outer(z, z)); n2 = Q2[:, 0]
        G2 = PU(np.outer(n2, n2)); g22 = np.tensordot(G2, np.outer(n2, n2))
        ok2 = test(S2 + 2 * abs(w2[0]) / g22 * G2)
    else: ok2 = True
    rec("exact n k=2, two steps", it, ok2)
    # n from LDL^T of S + delta I + zz^T/4*c  (as the device factorises S + delta I; z direction has pivot ~0 -> it uses S+delta I directly)
    Sd = S + delta * np.eye(10)
    L, dd = ldl(Sd)
    kneg = int(np.argmin(dd))
    e = np.zeros(10); e[kneg] = 1
    nl = np.linalg.solve(L.T, e); nl = Pz @ nl; nl /= np.linalg.norm(nl)
    ray = nl @ S @ nl
    Gl = PU(np.outer(nl, nl)); gl2 = np.tensordot(Gl, np.outer(nl, nl))
    rec("ldl n, k=2 (rayleigh)", it, test(S + 2 * abs(ray) / gl2 * Gl) if ray < 0 else False)
    rec("ldl n cos", it, abs(nl @ nvec) > 0.95)
    # inverse iteration: sigma = 2|ray| or fixed 2e-3... start from nl
    for sig in (1e-3, 3e-3, 1e-2):
        B = S + sig * np.eye(10) + np.outer(z, z)
        if lam_min(B) <= 0: rec(f"invit sig={sig} PD", it, False); continue
        rec(f"invit sig={sig} PD", it, True)
        for start in ("ldl", "ones"):
            x = nl.copy() if start == "ldl" else Pz @ np.ones(10)
            for itn in range(2):
                x = np.linalg.solve(B, x); x = Pz @ x; x /= np.linalg.norm(x)
                ray = x @ S @ x
                Gx = PU(np.outer(x, x)); gx2 = np.tensordot(Gx, np.outer(x, x))
                rec(f"invit sig={sig} start={start} its={itn+1} k=2", it, (ray < 0) and test(S + 2 * abs(ray) / gx2 * Gx))
for name in stats:
    tot = sum(len(v) for v in stats[name].values()); okc = sum(sum(v) for v in stats[name].values())
    print(f"{name:44s} {okc}/{tot} = {okc/tot:.3f}   by it: " + "  ".join(f"{it}:{sum(v)}/{len(v)}" for it, v in sorted(stats[name].items())))
