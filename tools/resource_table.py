"""Kernel resource table of a verbose build (container):  python -m cvxpnpl_amd.build --force -v > log 2>&1; python tools/resource_table.py log
One line per kernel: VGPRs, AGPRs, scratch bytes per lane, wavefronts per SIMD, spilled SGPRs / VGPRs, LDS bytes per block."""
import re
import subprocess
import sys


def main(path):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sgprS':>6s} {'vgprS':>6s} {'LDS':>7s}")
    for r, d in zip(rows, names):
        d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "")
        print(f"{d[:72]:72s} {r.get('VGPRs'):>5s} {r.get('AGPRs'):>5s} {r.get('ScratchSize [bytes/lane]'):>8s} {r.get('Occupancy [waves/SIMD]'):>4s} "
              f"{r.get('SGPRs Spill'):>6s} {r.get('VGPRs Spill'):>6s} {r.get('LDS Size [bytes/block]'):>7s}")


if __name__ == "__main__":
    main(sys.argv[1])
