"""Kernel resource table of a build (container):  python tools/resource_table.py [remarks file]
Default: the remarks cvxpnpl_amd/build.py keeps beside the library (cvxpnpl_amd/libcvxpnpl_amd.resources.txt).  One line per kernel:
VGPRs, AGPRs, scratch bytes per lane, wavefronts per SIMD, spilled SGPRs / VGPRs, LDS bytes per block.
`--write-golden` rewrites tests/golden/kernel_resources.json (the table tests/test_kernel_resources.py holds a build against)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpnpl_amd import build as _b  # noqa: E402


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    table = _b.kernel_resources(args[0] if args else _b.RESOURCES)
    print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sgprS':>6s} {'vgprS':>6s} {'LDS':>7s}")
    for name, r in table.items():
        print(f"{name[:72]:72s} {r['vgpr']:>5d} {r['agpr']:>5d} {r['scratch']:>8d} {r['occupancy']:>4d} {r['sgpr_spill']:>6d} {r['vgpr_spill']:>6d} {r['lds']:>7d}")
    if "--write-golden" in argv:
        path = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
        json.dump(table, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1:])
