"""Search for the sweep ordering of the quad layout's one-sided Jacobi (quad_kernel.h, kATab; round 5).

Ten columns, one per lane of a 16-lane DPP row.  A step pairs the columns five by five; both lanes of a pair end the step holding both
columns (own + fetched), so they may swap them for free.  A step whose pairs sit on lanes (2k, 2k + 1) needs no exchange instruction at
all in single precision: the partner's element is a DPP quad_perm [1,0,3,2] operand of the FMA itself.  Any other pairing goes through
ds_bpermute.  Wanted: nine steps, every one of the 45 pairs exactly once, as many neighbour steps as possible.

Facts this script establishes (run it: a few seconds):
  * with single-instruction DPP pairings only (quad_perm involutions, row_mirror, row_half_mirror, row_ror:8) no two CONSECUTIVE steps
    can both pair all ten columns: the ten occupied lanes would have to be closed under two of them, i.e. a union of full quads (the
    xor-type controls generate 4-cycles) -- so an all-DPP nine-step sweep does not exist (the odd-even transposition ordering, ten
    steps with row_shl:1 / row_shr:1 for every second one, is all-DPP but needs two DPP operands per element and a tenth step:
    45 instead of 26 VALU instructions on those steps against the 15 + 11 ds_bpermute of a round-robin step);
  * alternating A X A X A X A X A (A = neighbour step, X = ds_bpermute step with free swaps) DOES exist: only the set of pairs matters
    at an A step, and P(t+2) = f(P(t)) with f any sub-involution of the matching M(t+1) of the X step between them; the search below
    finds P1, M2, P3, ..., P9, a 1-factorisation of K10, in a few hundred nodes.  LDS round trips per sweep: 9 -> 4.
Prints the column-level schedule and the lane-level table (partner lane + take bit per X step) that quad_kernel.h's make_atab() encodes.
"""
import itertools


def matchings(items):
    if not items:
        yield []
        return
    a = items[0]
    for i in range(1, len(items)):
        b = items[i]
        rest = items[1:i] + items[i + 1:]
        for m in matchings(rest):
            yield [(a, b)] + m


ALL = [m for m in matchings(list(range(10)))]  # 945


def search(n_a=5):
    used = set()

    def free(pairs):
        return all(p not in used for p in pairs)

    P1 = [(2 * k, 2 * k + 1) for k in range(5)]
    used.update(P1)
    seq = [("A", P1, None)]

    def dfs(P, na):
        if na == n_a:
            return True
        for M in ALL:
            if not free(M):
                continue
            used.update(M)
            for sm in range(1, 32):
                f = list(range(10))
                for k, (a, b) in enumerate(M):
                    if (sm >> k) & 1:
                        f[a], f[b] = b, a
                Q = sorted(tuple(sorted((f[a], f[b]))) for a, b in P)
                if len(set(Q)) == 5 and free(Q):
                    used.update(Q)
                    seq.append(("X", M, sm))
                    seq.append(("A", Q, None))
                    if dfs(Q, na + 1):
                        return True
                    seq.pop(); seq.pop()
                    used.difference_update(Q)
            used.difference_update(M)
        return False

    assert dfs(P1, 1)
    return seq


def lane_tables(seq):
    lane_of = list(range(10))
    words = [0] * 16
    for l in range(10, 16):
        words[l] = l | (l << 4) | (l << 8) | (l << 12)
    s = 0
    for kind, M, sm in seq:
        if kind != "X":
            continue
        for k, (a, b) in enumerate(M):
            la, lb = lane_of[a], lane_of[b]
            words[la] |= lb << (4 * s)
            words[lb] |= la << (4 * s)
            if (sm >> k) & 1:
                words[la] |= 1 << (16 + s)
                words[lb] |= 1 << (16 + s)
                lane_of[a], lane_of[b] = lb, la
        s += 1
    return words


def check(words):
    """from ANY placement the lane-level table rotates every pair exactly once (here: from two consecutive sweeps' placements)"""
    col = list(range(10))
    for sweep in range(3):
        met = set()
        for st in range(9):
            if st % 2 == 0:
                for k in range(5):
                    p = tuple(sorted((col[2 * k], col[2 * k + 1]))); assert p not in met; met.add(p)
            else:
                s = st // 2
                for l in range(10):
                    q = (words[l] >> (4 * s)) & 15
                    if q > l:
                        p = tuple(sorted((col[l], col[q]))); assert p not in met; met.add(p)
                        if (words[l] >> (16 + s)) & 1:
                            col[l], col[q] = col[q], col[l]
        assert len(met) == 45


if __name__ == "__main__":
    seq = search()
    for kind, M, sm in seq:
        print(kind, " ".join(f"({a},{b})" for a, b in M), "" if sm is None else f" swap mask 0x{sm:x}")
    w = lane_tables(seq)
    check(w)
    print("lane words:", ", ".join(f"0x{x:05x}" for x in w))
