// recover_core.h -- all poses of a rank > 1 SDP solution: the mathematics, host and device.
//
// Reference behaviour: cvxpnpl.py:499-513 with the rank > 1 branch :507 -> _constraint_ortho_det (:221-343) -> _re6q3
// (:156-218).  The SDP solution Z is not rank one (the relaxation is not tight: minimal / degenerate configurations), so
// r is sought in the span of the top 2 or 4 eigenvectors subject to the 21 quadratic identities of a rotation matrix.
// Plain scalar code without the standard library (own complex arithmetic, insertion sort), so that the same source is
// the host path (cvxpnpl_recover_multi, host_recover.cpp) and the device kernel (recover_kernel.h: one lane per flagged
// problem -- rare, branchy, a cold path that must not cost a D2H copy and a host thread per problem).
//
// Formulation: z = [r; 1] = Vt a with Vt (10 x k) the marginalised eigenbasis (last row (0,..,0,1)); the identities are
// z^T A_i z = 0 with the same 21 constraint matrices A_i as the SDP (cvxpnpl.py:404-435), hence the k x k forms
// P_i = Vt^T A_i Vt.
#pragma once
#include <math.h>

#include "solver_core.h"

namespace cvxr {

struct cd { // complex double
    double re, im;
};
CVX_HD cd cmul(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
CVX_HD cd cadd(cd a, cd b) { return cd{a.re + b.re, a.im + b.im}; }
CVX_HD cd csub(cd a, cd b) { return cd{a.re - b.re, a.im - b.im}; }
CVX_HD double cabs_(cd a) { return sqrt(a.re * a.re + a.im * a.im); }
CVX_HD cd cdiv(cd a, cd b)
{
    const double d = b.re * b.re + b.im * b.im;
    return cd{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}



// symmetric eigendecomposition (two-sided cyclic Jacobi), ascending eigenvalues
CVX_HD void eigh10(const double *Zsym, double *w, double (*V)[10])
{
    double A[10][10];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) { A[i][j] = Zsym[cvx::sidx(i, j)]; V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, all = 0;
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) { all += A[i][j] * A[i][j]; if (i != j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-34 * all) break;
        for (int p = 0; p < 9; ++p)
            for (int q = p + 1; q < 10; ++q) {
                if (A[p][q] == 0.0) continue;
                double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                double t = (th < 0 ? -1.0 : 1.0) / (fabs(th) + sqrt(th * th + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 10; ++k) { double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
                for (int k = 0; k < 10; ++k) { double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
                for (int k = 0; k < 10; ++k) { double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
            }
    }
    int order[10];
    for (int i = 0; i < 10; ++i) order[i] = i;
    for (int i = 1; i < 10; ++i) { // insertion sort, ascending eigenvalues
        const int o = order[i];
        int j = i - 1;
        while (j >= 0 && A[order[j]][order[j]] > A[o][o]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
    }
    double Vs[10][10];
    for (int j = 0; j < 10; ++j) { w[j] = A[order[j]][order[j]]; for (int i = 0; i < 10; ++i) Vs[i][j] = V[i][order[j]]; }
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) V[i][j] = Vs[i][j];
}

// the 21 homogeneous constraint matrices as sparse (i, j, coef) lists on z_i z_j
struct Term { int i, j; double c; };
CVX_HD int constraint_terms(int idx, Term *t)
{
    if (idx < 15) {
        for (int k = 0; k < 3; ++k) { t[k].i = cvx::tri_i(idx, k); t[k].j = cvx::tri_j(idx, k); t[k].c = cvx::tri_s(idx, k); }
        return 3;
    }
    int d = idx - 15; // 0..2 rows of R, 3..5 columns of R: squared norm minus z9^2
    for (int k = 0; k < 3; ++k) {
        int e = d < 3 ? 3 * k + d : 3 * (d - 3) + k;
        t[k].i = e; t[k].j = e; t[k].c = 1.0;
    }
    t[3].i = 9; t[3].j = 9; t[3].c = -1.0;
    return 4;
}

// roots of a polynomial (descending coefficients), Durand-Kerner; returns count
CVX_HD int poly_roots(const double *p, int deg, cd *roots)
{
    int lead = 0;
    while (lead < deg && p[lead] == 0.0) ++lead;
    int n = deg - lead;
    if (n <= 0) return 0;
    cd z[8];
    double scale = 0;
    for (int i = 1; i <= n; ++i) {
        const double s = pow(fabs(p[lead + i] / p[lead]), 1.0 / i);
        scale = s > scale ? s : scale;
    }
    scale = scale > 0 ? scale : 1.0;
    for (int i = 0; i < n; ++i) { const double ph = 2 * M_PI * i / n + 0.4; z[i] = cd{scale * cos(ph), scale * sin(ph)}; }
    for (int it = 0; it < 500; ++it) {
        double delta = 0;
        for (int i = 0; i < n; ++i) {
            cd num = cd{p[lead], 0.0};
            for (int k = 1; k <= n; ++k) num = cadd(cmul(num, z[i]), cd{p[lead + k], 0.0});
            cd den = cd{p[lead], 0.0};
            for (int j = 0; j < n; ++j) if (j != i) den = cmul(den, csub(z[i], z[j]));
            if (cabs_(den) == 0.0) den = cd{1e-300, 0.0};
            const cd step = cdiv(num, den);
            z[i] = csub(z[i], step);
            const double rel = cabs_(step) / (cabs_(z[i]) + 1e-300);
            delta = rel > delta ? rel : delta;
        }
        if (delta < 1e-16) break;
    }
    // Newton polish on the original polynomial
    for (int i = 0; i < n; ++i)
        for (int it = 0; it < 3; ++it) {
            cd f = cd{p[lead], 0.0}, df = cd{0.0, 0.0};
            for (int k = 1; k <= n; ++k) { df = cadd(cmul(df, z[i]), f); f = cadd(cmul(f, z[i]), cd{p[lead + k], 0.0}); }
            if (cabs_(df) > 0) z[i] = csub(z[i], cdiv(f, df));
        }
    for (int i = 0; i < n; ++i) roots[i] = z[i];
    return n;
}

struct Poly { double c[5]; }; // ascending powers
CVX_HD Poly mul(const Poly &a, const Poly &b)
{
    Poly r;
    for (int i = 0; i < 5; ++i) r.c[i] = 0.0;
    for (int i = 0; i < 5; ++i) for (int j = 0; i + j < 5; ++j) r.c[i + j] += a.c[i] * b.c[j];
    return r;
}
CVX_HD Poly axpy(const Poly &a, double s, const Poly &b) { Poly r; for (int i = 0; i < 5; ++i) r.c[i] = a.c[i] + s * b.c[i]; return r; }
CVX_HD Poly lin(double c0, double c1) { Poly r; r.c[0] = c0; r.c[1] = c1; r.c[2] = 0; r.c[3] = 0; r.c[4] = 0; return r; }

// E6Q3 (cvxpnpl.py:156-218): rows [a^2 b^2 c^2 ab ac bc a b c 1] . coef = 0.  Reduce the six
// quadratic monomials by least squares; keep b^2, c^2, bc as linear forms in (b, c, 1) with
// coefficients linear in a; the three syzygies b(bc)=c(b^2), c(bc)=b(c^2), (b^2)(c^2)=(bc)^2
// give M(a) [b c 1]^T = 0; det M(a) is a quartic in a.
CVX_HD int e6q3(const double (*A)[10], int rows, double *a_out, double *b_out, double *c_out)
{
    double G[6][6], H[6][4];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) G[i][j] = 0.0; for (int j = 0; j < 4; ++j) H[i][j] = 0.0; }
    for (int r = 0; r < rows; ++r)
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) G[i][j] += A[r][i] * A[r][j];
            for (int j = 0; j < 4; ++j) H[i][j] += A[r][i] * A[r][6 + j];
        }
    // Gauss-Jordan with partial pivoting on [G | H]
    for (int k = 0; k < 6; ++k) {
        int p = k;
        for (int i = k + 1; i < 6; ++i) if (fabs(G[i][k]) > fabs(G[p][k])) p = i;
        if (G[p][k] == 0.0) return -1;
        if (p != k) {
            for (int j = 0; j < 6; ++j) { const double t_ = G[k][j]; G[k][j] = G[p][j]; G[p][j] = t_; }
            for (int j = 0; j < 4; ++j) { const double t_ = H[k][j]; H[k][j] = H[p][j]; H[p][j] = t_; }
        }
        double inv = 1 / G[k][k];
        for (int j = 0; j < 6; ++j) G[k][j] *= inv;
        for (int j = 0; j < 4; ++j) H[k][j] *= inv;
        for (int i = 0; i < 6; ++i) {
            if (i == k) continue;
            double f = G[i][k];
            for (int j = 0; j < 6; ++j) G[i][j] -= f * G[k][j];
            for (int j = 0; j < 4; ++j) H[i][j] -= f * H[k][j];
        }
    }
    // monomial m = -H[m] . [a, b, c, 1]; keep m in {b^2 (1), c^2 (2), bc (5)} as forms over (b, c, 1)
    const int pick[3] = {1, 2, 5};
    Poly q[3][3];
    for (int m = 0; m < 3; ++m) {
        const double *h = H[pick[m]];
        q[m][0] = lin(-h[1], 0); q[m][1] = lin(-h[2], 0); q[m][2] = lin(-h[3], -h[0]);
    }
    auto times_b = [&](const Poly *L, Poly *out) { // b * (l0 b + l1 c + l2) = l0 b^2 + l1 bc + l2 b
        for (int j = 0; j < 3; ++j) out[j] = axpy(mul(L[0], q[0][j]), 1.0, mul(L[1], q[2][j]));
        out[0] = axpy(out[0], 1.0, L[2]);
    };
    auto times_c = [&](const Poly *L, Poly *out) { // c * L = l0 bc + l1 c^2 + l2 c
        for (int j = 0; j < 3; ++j) out[j] = axpy(mul(L[0], q[2][j]), 1.0, mul(L[1], q[1][j]));
        out[1] = axpy(out[1], 1.0, L[2]);
    };
    Poly M[3][3], u[3], v[3];
    times_b(q[2], u); times_c(q[0], v);
    for (int j = 0; j < 3; ++j) M[0][j] = axpy(u[j], -1.0, v[j]);
    times_c(q[2], u); times_b(q[1], v);
    for (int j = 0; j < 3; ++j) M[1][j] = axpy(u[j], -1.0, v[j]);
    {
        const Poly *X = q[0], *Y = q[1], *W = q[2];
        Poly kb2 = axpy(mul(X[0], Y[0]), -1.0, mul(W[0], W[0]));
        Poly kc2 = axpy(mul(X[1], Y[1]), -1.0, mul(W[1], W[1]));
        Poly kbc = axpy(axpy(mul(X[0], Y[1]), 1.0, mul(X[1], Y[0])), -2.0, mul(W[0], W[1]));
        Poly kb = axpy(axpy(mul(X[0], Y[2]), 1.0, mul(X[2], Y[0])), -2.0, mul(W[0], W[2]));
        Poly kc = axpy(axpy(mul(X[1], Y[2]), 1.0, mul(X[2], Y[1])), -2.0, mul(W[1], W[2]));
        Poly k1 = axpy(mul(X[2], Y[2]), -1.0, mul(W[2], W[2]));
        for (int j = 0; j < 3; ++j) M[2][j] = axpy(axpy(mul(kb2, q[0][j]), 1.0, mul(kc2, q[1][j])), 1.0, mul(kbc, q[2][j]));
        M[2][0] = axpy(M[2][0], 1.0, kb); M[2][1] = axpy(M[2][1], 1.0, kc); M[2][2] = axpy(M[2][2], 1.0, k1);
    }
    Poly det = mul(M[0][0], axpy(mul(M[1][1], M[2][2]), -1.0, mul(M[1][2], M[2][1])));
    det = axpy(det, -1.0, mul(M[0][1], axpy(mul(M[1][0], M[2][2]), -1.0, mul(M[1][2], M[2][0]))));
    det = axpy(det, 1.0, mul(M[0][2], axpy(mul(M[1][0], M[2][1]), -1.0, mul(M[1][1], M[2][0]))));
    double p[5] = {det.c[4], det.c[3], det.c[2], det.c[1], det.c[0]};
    cd roots[8];
    int nr = poly_roots(p, 4, roots);
    for (int k = 0; k < nr; ++k) {
        double a = roots[k].re; // the reference keeps Re() of complex roots too (cvxpnpl.py:186)
        double Mv[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0, pw = 1;
                for (int e = 0; e < 5; ++e) { s += M[i][j].c[e] * pw; pw *= a; }
                Mv[i][j] = s;
            }
        double g00 = 0, g01 = 0, g11 = 0, r0 = 0, r1 = 0;
        for (int i = 0; i < 3; ++i) {
            g00 += Mv[i][0] * Mv[i][0]; g01 += Mv[i][0] * Mv[i][1]; g11 += Mv[i][1] * Mv[i][1];
            r0 += Mv[i][0] * Mv[i][2]; r1 += Mv[i][1] * Mv[i][2];
        }
        double dt = g00 * g11 - g01 * g01;
        a_out[k] = a;
        b_out[k] = -(g11 * r0 - g01 * r1) / dt;
        c_out[k] = -(g00 * r1 - g01 * r0) / dt;
    }
    return nr;
}


// All poses of one solution: Z55 = vech(Z) (cvxpnpl.py:346-370 order), B27 (t = -B r), Q45 (optional: packed A^T A; every
// pose is then Newton-polished on SO(3)).  R_out [4][9], t_out [4][3].  Returns the number of poses (2 or 4), 1 for a
// rank-1 Z, -1 if the rank is 0 / Z is not finite (reference: NotImplementedError / NaN sentinel).
CVX_HD int recover_multi(const double *Z55, const double *B27, const double *Q45, double *R_out, double *t_out)
{
    for (int i = 0; i < 55; ++i) if (!(Z55[i] == Z55[i])) return -1;
    double w[10], V[10][10];
    eigh10(Z55, w, V);
    int rank = 0;
    for (int i = 0; i < 10; ++i) rank += w[i] > 1e-3; // cvxpnpl.py:502
    double rc[4][9];
    int np = 0;
    if (rank == 1) {
        for (int i = 0; i < 9; ++i) rc[0][i] = V[i][9] / V[9][9];
        np = 1;
    } else {
        int k = 2 * ((rank + 1) / 2) < 4 ? 2 * ((rank + 1) / 2) : 4; // cvxpnpl.py:231
        if (k != 2 && k != 4) return -1;
        // marginalised basis Vt (10 x k): last column v0 = top eigenvector / its last entry,
        // the others have their last entry eliminated (cvxpnpl.py:234-236)
        // Deviation from the reference for robustness: v0 is built from the eigenvector (among the k
        // used) with the largest last entry, not blindly from the top one -- for an exact two-fold
        // ambiguity the eigenvalues are equal, the basis of the eigenspace is arbitrary and the top
        // vector's last entry can be ~0 (the reference divides by it and returns NaN poses).
        // (the top one, like the reference, whenever its last entry is usable: results then agree with the
        // reference's to rounding also for an inconsistent system, where the least-squares steps of the rank-4
        // branch depend on the parametrisation)
        int big = 9;
        for (int c = 10 - k; c < 10; ++c) if (fabs(V[9][c]) > fabs(V[9][big])) big = c;
        const int piv = fabs(V[9][9]) > 1e-3 * fabs(V[9][big]) ? 9 : big;
        double Vt[10][4];
        for (int i = 0; i < 10; ++i) Vt[i][k - 1] = V[i][piv] / V[9][piv];
        int a = 0;
        for (int col = 10 - k; col < 10; ++col) {
            if (col == piv) continue;
            for (int i = 0; i < 10; ++i) Vt[i][a] = V[i][col] - V[9][col] * Vt[i][k - 1];
            ++a;
        }
        double P[21][4][4];
        for (int q = 0; q < 21; ++q) for (int a_ = 0; a_ < 4; ++a_) for (int b_ = 0; b_ < 4; ++b_) P[q][a_][b_] = 0.0;
        for (int q = 0; q < 21; ++q) {
            Term t[4];
            int nt = constraint_terms(q, t);
            for (int e = 0; e < nt; ++e)
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) {
                        double s = 0.5 * t[e].c * (Vt[t[e].i][a] * Vt[t[e].j][b] + Vt[t[e].j][a] * Vt[t[e].i][b]);
                        P[q][a][b] += s;
                    }
        }
        double alpha[4][4];
        if (k == 2) { // cvxpnpl.py:303-315: mean coefficients, general quadratic formula
            double c0 = 0, c1 = 0, c2 = 0;
            for (int q = 0; q < 21; ++q) { c0 += P[q][0][0]; c1 += 2 * P[q][0][1]; c2 += P[q][1][1]; }
            c0 /= 21; c1 /= 21; c2 /= 21;
            double root = sqrt((c1 * c1 - 4 * c0 * c2 > 0.0 ? c1 * c1 - 4 * c0 * c2 : 0.0));
            alpha[0][0] = (-c1 + root) / (2 * c0); alpha[0][1] = 1;
            alpha[1][0] = (-c1 - root) / (2 * c0); alpha[1][1] = 1;
            np = 2;
        } else { // cvxpnpl.py:317-338
            double A[21][10];
            for (int q = 0; q < 21; ++q) {
                A[q][0] = P[q][0][0]; A[q][1] = P[q][1][1]; A[q][2] = P[q][2][2];
                A[q][3] = 2 * P[q][0][1]; A[q][4] = 2 * P[q][0][2]; A[q][5] = 2 * P[q][1][2];
                A[q][6] = 2 * P[q][0][3]; A[q][7] = 2 * P[q][1][3]; A[q][8] = 2 * P[q][2][3]; A[q][9] = P[q][3][3];
            }
            double a[4], b[4], c[4];
            np = e6q3(A, 21, a, b, c);
            if (np <= 0) return -1;
            for (int s = 0; s < np; ++s) { alpha[s][0] = a[s]; alpha[s][1] = b[s]; alpha[s][2] = c[s]; alpha[s][3] = 1; }
        }
        for (int s = 0; s < np; ++s)
            for (int i = 0; i < 9; ++i) {
                double acc = 0;
                for (int a = 0; a < k; ++a) acc += alpha[s][a] * Vt[i][a];
                rc[s][i] = acc;
            }
    }
    for (int s = 0; s < np; ++s) {
        // U V^T of the 3x3 (no determinant correction, cvxpnpl.py:510-511), t = -B r (:513)
        double M0[9], R[9], r[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M0[i * 3 + j] = rc[s][3 * j + i];
        cvx::polar3(M0, R, 40);
        // optional: Newton-polish r^T Q r on SO(3) from this candidate (each pose of an ambiguous problem is
        // a local minimiser; Z from a first-order solver only locates it to ~1e-5)
        if (Q45 && cvx::det3(R) > 0) {
            double tr = 0, Qs[45];
            for (int i = 0; i < 9; ++i) tr += Q45[cvx::qidx(i, i)];
            if (tr > 0) {
                for (int i = 0; i < 45; ++i) Qs[i] = Q45[i] / tr;
                double Rp[9];
                for (int i = 0; i < 9; ++i) Rp[i] = R[i];
                cvx::so3_newton(Qs, Rp, 8);
                bool fin = true;
                for (int i = 0; i < 9; ++i) fin &= (Rp[i] == Rp[i]);
                if (fin) for (int i = 0; i < 9; ++i) R[i] = Rp[i];
            }
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[3 * j + i] = R[i * 3 + j];
        for (int i = 0; i < 9; ++i) R_out[9 * s + i] = R[i];
        for (int i = 0; i < 3; ++i) {
            double acc = 0;
            for (int j = 0; j < 9; ++j) acc += B27[i * 9 + j] * r[j];
            t_out[3 * s + i] = -acc;
        }
    }
    return np;
}


} // namespace cvxr
