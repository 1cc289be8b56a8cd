"""Host build of the scalar core (tests/hostsim): iteration histograms with the eigen-gradient step (dual_refine = 1) and with the barrier
Newton solve behind it (2), from the third / second / first attempt (dual_refine = mode + 16 * (1 + first attempt); needs a host build with -DCVX_REFINE_FROM_EXPERIMENT)."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hostsim
from cvxpnpl_amd import synth

def hist(it):
    return {int(k): int(v) for k, v in zip(*np.unique(it, return_counts=True))}

sets = [("pnp10 10k seed42", synth.make_pnpl(10000, 10, 0, 2.0, seed=42), {}), ("pnp10 10k seed1", synth.make_pnpl(10000, 10, 0, 2.0, seed=1), {}),
        ("pnp10 125k first_check 6", synth.make_pnpl(125000, 10, 0, 2.0, seed=42), {"first_check": 6}),
        ("pnpl 5+5 30k first_check 6", synth.make_pnpl(30000, 5, 5, 2.0, seed=42), {"first_check": 6}),
        ("pnp6 10k", synth.make_pnpl(10000, 6, 0, 2.0, seed=42), {}), ("pnp4 5k", synth.make_pnpl(5000, 4, 0, 2.0, seed=42), {})]
for name, d, kw in sets:
    ref = None
    for label, rf in (("off", 0), ("grad from 3rd", 1), ("newton from 3rd", 2), ("newton from 2nd", 2 + 16 * 2), ("newton from 1st", 2 + 16 * 1)):
        r = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], d.get("line_2d"), d.get("line_3d"), d["K"], opts=hostsim.default_opts(dual_refine=rf, **kw))
        it = r["iters"]
        if ref is None:
            ref = r
        both = (r["status"] == 0) & (ref["status"] == 0)
        geo = synth.geodesic(r["R"], ref["R"])[both].max()
        gap = (r["cost"][:, 0] - r["cost"][:, 1])[r["status"] == 0]
        print(f"{name:28s} {label:16s} certified {int((r['status'] == 0).sum()):6d} mean {it.mean():.3f} max {it.max():4d} hist {hist(it) if len(hist(it)) < 14 else '...'} "
              f"max geodesic vs off {geo:.1e} gap in [{gap.min():.1e}, {gap.max():.1e}]", flush=True)
