import numpy as np, sys
from newton import unpack, family_basis, zperp
d = np.load(sys.argv[1]); out = d["out"]
fail = out[out[:, 2] == 0]
rs = np.random.RandomState(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 655
sel = fail if len(fail) <= n else fail[rs.choice(len(fail), n, replace=False)]
def lam_min(M): return np.linalg.eigvalsh(M)[0]
stats = {}
def rec(name, it, ok): stats.setdefault(name, {}).setdefault(it, []).append(ok)
for r in sel:
    it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
    z = np.concatenate([R.T.reshape(-1), [1.0]])
    U = family_basis(z); P = zperp(z)
    M0 = P.T @ S @ P; Mk = np.stack([P.T @ u @ P for u in U])
    # baseline D shift
    D = np.eye(10) - np.outer(z, z) / 4
    DU = sum(np.tensordot(u, D) * u for u in U)  # projection onto U (orthonormal basis)
    okD = any(lam_min(P.T @ (S + m * DU) @ P) > -delta for m in (0.015, 0.015 / 4))
    rec("Dshift", it, okD)
    # (2) second-order perturbation step, exact eigen
    w, Q = np.linalg.eigh(M0)
    nvec = Q[:, 0]
    g = np.array([nvec @ m @ nvec for m in Mk])
    Wk = np.stack([m @ nvec for m in Mk])  # 14 x 9
    Xp = (Q[:, 1:] / (w[1:] - w[0])) @ Q[:, 1:].T
    Cm = Wk @ Xp @ Wk.T
    vstar = 0.5 * np.linalg.solve(Cm + 1e-12 * np.eye(14), g)
    for th in (0.25, 0.5, 1.0):
        rec(f"pert2 th={th}", it, lam_min(M0 + np.tensordot(th * vstar, Mk, 1)) > -delta)
    # ladder: try th=1, .5, .25
    rec("pert2 ladder", it, any(lam_min(M0 + np.tensordot(th * vstar, Mk, 1)) > -delta for th in (1.0, 0.5, 0.25)))
    # gradient only: v = tau g, tau so that first-order gain = k*|lam|
    for k in (2, 4, 8):
        tau = k * abs(w[0]) / (g @ g)
        rec(f"grad k={k}", it, lam_min(M0 + np.tensordot(tau * g, Mk, 1)) > -delta)
    # (3) barrier newton K steps in (v, t)
    for (gap, mu) in ((1e-3, 1e-4), (1e-3, 3e-5), (3e-4, 3e-5)):
        v = np.zeros(14); t = w[0] - gap
        okK = {}
        for K in range(1, 7):
            M = M0 + np.tensordot(v, Mk, 1) - t * np.eye(9)
            X = np.linalg.inv(M)
            XM = np.stack([X @ m for m in Mk] + [-X])
            gr = -np.array([np.trace(a) for a in XM]); gr[-1] -= 1.0 / mu
            H = np.einsum('aij,bji->ab', XM, XM)
            dx = -np.linalg.solve(H, gr)
            # damped Newton: step 1/(1+lambda_newton)
            lamN = np.sqrt(max(dx @ H @ dx, 0))
            a = 1.0 / (1.0 + lamN) if lamN > 0.25 else 1.0
            v = v + a * dx[:14]; t = t + a * dx[14]
            okK[K] = lam_min(M0 + np.tensordot(v, Mk, 1)) > -delta
            rec(f"barrier gap={gap} mu={mu} K={K}", it, any(okK.values()))
for name in stats:
    tot = sum(len(v) for v in stats[name].values()); okc = sum(sum(v) for v in stats[name].values())
    print(f"{name:40s} rescued {okc}/{tot} = {okc/tot:.3f}   by it: " + "  ".join(f"{it}:{sum(v)}/{len(v)}" for it, v in sorted(stats[name].items())))
