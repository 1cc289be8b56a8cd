import numpy as np, sys, time
np.set_printoptions(linewidth=220, precision=4)
def sidx(i, j):
    if i > j: i, j = j, i
    return i * 10 - i * (i - 1) // 2 + (j - i)
def unpack(s):
    M = np.zeros((10, 10))
    for i in range(10):
        for j in range(i, 10):
            M[i, j] = M[j, i] = s[sidx(i, j)]
    return M
# 22 equality rows as symmetric matrices with <A,Z> = sum coef Z_ij (offdiag pair once => A_ij = coef/2)
ROWS = [
 [(9,9,1)],
 [(0,0,1),(3,3,1),(6,6,1),(9,9,-1)], [(0,1,1),(3,4,1),(6,7,1)], [(0,2,1),(3,5,1),(6,8,1)],
 [(1,1,1),(4,4,1),(7,7,1),(9,9,-1)], [(1,2,1),(4,5,1),(7,8,1)], [(2,2,1),(5,5,1),(8,8,1),(9,9,-1)],
 [(0,0,1),(1,1,1),(2,2,1),(9,9,-1)], [(0,3,1),(1,4,1),(2,5,1)], [(0,6,1),(1,7,1),(2,8,1)],
 [(3,3,1),(4,4,1),(5,5,1),(9,9,-1)], [(3,6,1),(4,7,1),(5,8,1)], [(6,6,1),(7,7,1),(8,8,1),(9,9,-1)],
 [(1,5,1),(2,4,-1),(6,9,-1)], [(2,3,1),(0,5,-1),(7,9,-1)], [(0,4,1),(1,3,-1),(8,9,-1)],
 [(4,8,1),(5,7,-1),(0,9,-1)], [(5,6,1),(3,8,-1),(1,9,-1)], [(3,7,1),(4,6,-1),(2,9,-1)],
 [(2,7,1),(1,8,-1),(3,9,-1)], [(0,8,1),(2,6,-1),(4,9,-1)], [(1,6,1),(0,7,-1),(5,9,-1)],
]
def amats():
    A = np.zeros((22, 10, 10))
    for k, row in enumerate(ROWS):
        for (i, j, c) in row:
            if i == j: A[k, i, i] += c
            else: A[k, i, j] += c / 2; A[k, j, i] += c / 2
    return A
AM = amats()
def family_basis(z):
    """orthonormal (Frobenius) basis of U = {X in span A_i : X z = 0}, as [14,10,10]"""
    # orthonormal basis of span A_i first
    F = AM.reshape(22, 100)
    u, s, vt = np.linalg.svd(F, full_matrices=False)
    r = (s > 1e-10).sum(); assert r == 21
    Bs = vt[:r].reshape(r, 10, 10)
    G = np.stack([b @ z for b in Bs], 1)  # 10 x 21
    u2, s2, vt2 = np.linalg.svd(G)
    rk = (s2 > 1e-10).sum()
    N = vt2[rk:]  # (21-rk) x 21
    U = np.tensordot(N, Bs, 1)
    return U
def zperp(z):
    q, _ = np.linalg.qr(np.concatenate([z[:, None], np.eye(10)], 1))
    return q[:, 1:10]
def solve_max_t(S1, U, P, iters=60, verbose=False):
    """max t: P^T (S1 + sum v_k U_k) P >= t I by barrier path following; returns t_best, v"""
    M0 = P.T @ S1 @ P; Mk = np.stack([P.T @ u @ P for u in U])
    nv = len(U); v = np.zeros(nv)
    lam = np.linalg.eigvalsh(M0)[0]
    t = lam - max(1e-3, abs(lam))
    mu = 1.0
    best = lam
    for it in range(iters):
        for inner in range(3):
            M = M0 + np.tensordot(v, Mk, 1) - t * np.eye(9)
            lmin_cur = np.linalg.eigvalsh(M)[0]
            X = np.linalg.inv(M)
            XM = np.stack([X @ m for m in Mk] + [-X])
            g = -np.array([np.trace(a) for a in XM]); g[-1] -= 1.0 / mu
            H = np.einsum('aij,bji->ab', XM, XM)
            dx = -np.linalg.solve(H, g)
            # line search for feasibility
            a = 1.0
            while True:
                vn = v + a * dx[:nv]; tn = t + a * dx[nv]
                lmn = np.linalg.eigvalsh(M0 + np.tensordot(vn, Mk, 1))[0]
                if lmn - tn > 0.05 * lmin_cur: break
                a *= 0.5
            v, t = vn, tn
        lm = np.linalg.eigvalsh(M0 + np.tensordot(v, Mk, 1))[0]
        best = max(best, lm)
        mu *= 0.3
    return best, v
if __name__ == "__main__":
    f = sys.argv[1]
    d = np.load(f); out = d["out"]
    fail = out[out[:, 2] == 0]
    print("failed records", len(fail), "by iteration", np.bincount(fail[:, 1].astype(int)))
    rs = np.random.RandomState(0)
    sel = fail if len(fail) <= 300 else fail[rs.choice(len(fail), 300, replace=False)]
    res = []
    for r in sel:
        it = int(r[1]); delta = r[3]; S = unpack(r[4:59]) - delta * np.eye(10); R = r[59:68].reshape(3, 3)
        z = np.concatenate([R.T.reshape(-1), [1.0]])  # z[3j+i] = R[i][j]
        assert np.abs(S @ z).max() < 1e-9
        U = family_basis(z)
        assert U.shape[0] == 14, U.shape
        P = zperp(z)
        lam0 = np.linalg.eigvalsh(P.T @ S @ P)
        tb, v = solve_max_t(S, U, P)
        res.append((it, lam0[0], lam0[1], tb, np.linalg.norm(v)))
    res = np.array(res)
    print("it  lam_min(S1)  lam2   t_max  |v|")
    for row in res[:40]: print("%2d  %+.3e  %+.3e  %+.3e  %.3e" % tuple(row))
    print("fraction with t_max > 0:", (res[:, 3] > 0).mean(), " min t_max", res[:, 3].min(), "median t_max", np.median(res[:, 3]), "median lam_min", np.median(res[:, 1]))
    np.save("/tmp/exp/tmax.npy", res)
