import numpy as np, sys, ctypes as C
from policy import go, L
from newton import unpack
sys.path.insert(0, "/root/repo")
from cvxpnpl_amd import synth
from collect import run
B = int(sys.argv[1]); thr = int(sys.argv[2])
d = synth.make_pnpl(B, 10, 0, 2.0, seed=42)
L.dr_policy(1, 0.005, 2, 2.0, 1, 1, 1)
st, it, R, out = run(d, 10, 0)
print("mean", it.mean(), "n>=9", (it >= 9).sum(), "n>=12", (it >= 12).sum(), "max", it.max())
slow = np.where(it >= thr)[0]
for b in slow:
    recs = out[out[:, 0] == b]
    recs = recs[np.argsort(recs[:, 1])]
    line = f"b={b} final it {it[b]} st {st[b]}: "
    for r in recs:
        Rr = r[59:68]; same = np.abs(Rr - R[b]).max() < 1e-6
        Wp = unpack(r[168:223]); w = np.linalg.eigvalsh(Wp)[::-1]
        S = unpack(r[4:59]) - r[3] * np.eye(10)
        z = np.concatenate([Rr.reshape(3,3).T.reshape(-1), [1.0]])
        ws = np.linalg.eigvalsh(S + np.outer(z, z))
        line += f"\n    [it {int(r[1])} ok {int(r[2])} pose{'==' if same else '!='}final eigZ {w[0]:.2f},{w[1]:.2f},{w[2]:.3f} eigS {ws[0]:.1e},{ws[1]:.1e},{ws[2]:.1e}] "
    print(line)
