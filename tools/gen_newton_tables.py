#!/usr/bin/env python3
"""Generates the constant tables of cvx::dual_newton (csrc/solver_core.h: kNt*): a sparse integer basis of the dual family in the frame of R,
U_I = { X in span A_i : X z_I = 0 }, z_I = [vec I3; 1] (14 matrices, each a signed sum of constraint matrices: 4, 6 or 13 upper entries), and
T_I = I - z_I z_I^T / 4 as the 15th.  The constraint matrices are the 22 rows of SURVEY.md A.5 (= the reference's _A, cvxpnpl.py:387-451, on
the unscaled vech).  tests/test_dual_retry.py checks the tables in the header against the same construction.   python tools/gen_newton_tables.py [--device]"""
import numpy as np
import sympy as sp

ROWS = [
    [(9, 9, 1)],
    [(0, 0, 1), (3, 3, 1), (6, 6, 1), (9, 9, -1)], [(0, 1, 1), (3, 4, 1), (6, 7, 1)], [(0, 2, 1), (3, 5, 1), (6, 8, 1)],
    [(1, 1, 1), (4, 4, 1), (7, 7, 1), (9, 9, -1)], [(1, 2, 1), (4, 5, 1), (7, 8, 1)], [(2, 2, 1), (5, 5, 1), (8, 8, 1), (9, 9, -1)],
    [(0, 0, 1), (1, 1, 1), (2, 2, 1), (9, 9, -1)], [(0, 3, 1), (1, 4, 1), (2, 5, 1)], [(0, 6, 1), (1, 7, 1), (2, 8, 1)],
    [(3, 3, 1), (4, 4, 1), (5, 5, 1), (9, 9, -1)], [(3, 6, 1), (4, 7, 1), (5, 8, 1)], [(6, 6, 1), (7, 7, 1), (8, 8, 1), (9, 9, -1)],
    [(1, 5, 1), (2, 4, -1), (6, 9, -1)], [(2, 3, 1), (0, 5, -1), (7, 9, -1)], [(0, 4, 1), (1, 3, -1), (8, 9, -1)],
    [(4, 8, 1), (5, 7, -1), (0, 9, -1)], [(5, 6, 1), (3, 8, -1), (1, 9, -1)], [(3, 7, 1), (4, 6, -1), (2, 9, -1)],
    [(2, 7, 1), (1, 8, -1), (3, 9, -1)], [(0, 8, 1), (2, 6, -1), (4, 9, -1)], [(1, 6, 1), (0, 7, -1), (5, 9, -1)],
]


def constraint_matrices_x2():
    """2 A_i as integer matrices (<A_i, Z> = sum coef Z_ij with every off-diagonal pair once)"""
    A = np.zeros((22, 10, 10), dtype=int)
    for k, row in enumerate(ROWS):
        for (i, j, c) in row:
            if i == j:
                A[k, i, i] += 2 * c
            else:
                A[k, i, j] += c
                A[k, j, i] += c
    return A


def tables():
    zI = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 1])
    A2 = constraint_matrices_x2()
    G = sp.Matrix([[int(v) for v in (A2[i] @ zI)] for i in range(22)]).T
    mats = []
    for v in G.nullspace():
        den = sp.ilcm(*[sp.fraction(x)[1] for x in v])
        mats.append(sum(int(x * den) * A2[i] for i, x in enumerate(v)))
    order = sorted(range(len(mats)), key=lambda k: (np.count_nonzero(mats[k]), k))
    sel = []
    for k in order:  # an independent subset, sparsest first
        F = np.array([mats[j].reshape(-1) for j in sel + [k]], dtype=float)
        if np.linalg.matrix_rank(F) == len(sel) + 1:
            sel.append(k)
    assert len(sel) == 14
    tabs = []
    for k in sorted(sel):
        m = mats[k]
        m = m // np.gcd.reduce(np.abs(m[m != 0]))
        tabs.append([(i, j, float(m[i, j])) for i in range(10) for j in range(i, 10) if m[i, j] != 0])
    T = np.eye(10) - np.outer(zI, zI) / 4.0
    tabs.append([(i, j, float(T[i, j])) for i in range(10) for j in range(i, 10) if T[i, j] != 0])
    return tabs


if __name__ == "__main__":
    tabs = tables()
    nmax = max(len(t) for t in tabs)
    print(f"constexpr int NT_N = {len(tabs)}, NT_MAX = {nmax};")
    print("constexpr int kNtCount[NT_N] = {" + ", ".join(str(len(t)) for t in tabs) + "};")
    for name, idx, fmt in (("kNtP", 0, "{:d}"), ("kNtQ", 1, "{:d}")):
        print(f"constexpr signed char {name}[NT_N][NT_MAX] = {{")
        for t in tabs:
            print("    {" + ", ".join(fmt.format(int(e[idx])) for e in t + [(0, 0, 0.0)] * (nmax - len(t))) + "},")
        print("};")
    print("constexpr double kNtC[NT_N][NT_MAX] = {")
    for t in tabs:
        print("    {" + ", ".join(repr(e[2]) for e in t + [(0, 0, 0.0)] * (nmax - len(t))) + "},")
    print("};")
    if "--device" not in __import__("sys").argv:
        raise SystemExit(0)
    # (device routine cvxw::coop_newton) the work items of the Hessian over the 14 basis matrices (T_I is handled in closed form): a pair (A, B)
    # loops over A's entries e0..e1 with B's slots unrolled in registers (width 6 if B has at most six entries, else 14); the orientation with
    # the smaller entries x width, the pairs of two 13-entry matrices split in two; heaviest first: lane l takes items l and NT_ITEMS - 1 - l
    nb14 = len(tabs) - 1
    items = []
    for a in range(nb14):
        for b in range(a, nb14):
            na, nb = len(tabs[a]), len(tabs[b])
            wid = lambda n: 6 if n <= 6 else 14
            A, B = (a, b) if na * wid(nb) <= nb * wid(na) else (b, a)
            nA = len(tabs[A])
            if nA > 6 and len(tabs[B]) > 6:
                items += [(A, B, 0, (nA + 1) // 2), (A, B, (nA + 1) // 2, nA)]
            else:
                items.append((A, B, 0, nA))
    cost = lambda it: (it[3] - it[2]) * (6 if len(tabs[it[1]]) <= 6 else 14)
    items.sort(key=lambda it: (-cost(it), it))
    print(f"constexpr int NT_ITEMS = {len(items)}; // heaviest {cost(items[0])} slots, lane maximum {max(cost(items[l]) + (cost(items[len(items) - 1 - l]) if len(items) - 1 - l >= 64 else 0) for l in range(64))}")
    for name, k in (("kNtItemA", 0), ("kNtItemB", 1), ("kNtItemE0", 2), ("kNtItemE1", 3)):
        print(f"constexpr signed char {name}[NT_ITEMS] = {{" + ", ".join(str(it[k]) for it in items) + "};")
