/*
 * oracle.c -- CPU restatement of the cvxpnpl hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library
 * under cvxpnpl_amd/csrc and fails loudly without a GPU.
 *
 * What it restates (reference = /root/reference/cvxpnpl.py, v1.1.0):
 *   orc_point_constraints   cvxpnpl.py:20-104   rows of [p]x (R P + t) = 0
 *   orc_line_constraints    cvxpnpl.py:107-153  rows of n^T (R P + t) = 0
 *   orc_eliminate           cvxpnpl.py:545-549  B = (N^T N)^-1 N^T C,  A = C - N B
 *   orc_vech10 / _inv       cvxpnpl.py:346-384  column-major lower-triangle pack/unpack
 *   orc_sdp_constraints     cvxpnpl.py:387-451  static 77x55 SCS data matrix, b = e0
 *   orc_scs_solve           call site cvxpnpl.py:485-489 -- the third-party solver.
 *   orc_solve_relaxation    cvxpnpl.py:454-520  Q, solve, eigh, rank, rank-1 ratio,
 *                                               SVD projection (no det fix), t = -B r,
 *                                               certificate |  ||Ar||^2 - dobj | > eps
 *   orc_constraint_ortho_det cvxpnpl.py:221-343 rank>1 multi-solution recovery
 *   orc_re6q3               cvxpnpl.py:156-218  6 quadrics / 3 unknowns via a quartic
 *   orc_pnp / orc_pnl / orc_pnpl  cvxpnpl.py:523-627
 *
 * Third-party arithmetic: the SDP is solved in the reference by `scs` (PyPI, cvxgrp/scs),
 * constrained only as scs>=2.0.0 (requirements.txt:4), absent from /root/reference and
 * from this image.  orc_scs_solve restates SCS's PUBLISHED algorithm (O'Donoghue, Chu,
 * Parikh, Boyd, "Conic optimization via operator splitting and homogeneous self-dual
 * embedding", JOTA 2016): ADMM on the homogeneous self-dual embedding
 *     u~ = (I+Q)^-1 (u+v);  u = Pi_C(a u~ + (1-a) u - v);  v = v - a u~ - (1-a) u + u_new
 * with over-relaxation a = 1.5 and SCS-2.x stopping rule (relative primal / dual /
 * gap residuals < eps).  Data equilibration and Anderson acceleration -- which change
 * the iterates but not the fixed point -- are not restated; a scalar `scale` between
 * the primal and dual blocks (SCS's `scale` setting) is.  The SDP optimum is solver
 * independent, so parity is anchored on it (see DESIGN.md "oracle").
 *
 * Pinning: tests/golden/ holds vectors produced by importing the reference itself
 * (stub `scs` module) in the build container -- tests/golden/make_golden.py -- and
 * tests/test_oracle_golden.py checks every function here against them.
 *
 * Everything is float64, scalar, single threaded (orc_*_batch loops use OpenMP when
 * compiled with -fopenmp; the cpu_baseline reports the thread count it used).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "oracle.h"

#define NV 55
#define NM 77
#define NU (NV + NM + 1)

/* ------------------------------------------------------------------ small dense LA */

/* Solve A X = B (A n x n, B n x nrhs, row-major) by LU with partial pivoting, the
 * method behind np.linalg.solve (LAPACK gesv).  Overwrites A and B.  -1 if singular. */
static int lu_solve(int n, double *A, int nrhs, double *B)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (best == 0.0 || !(best == best)) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < nrhs; ++j) { double t = B[k * nrhs + j]; B[k * nrhs + j] = B[p * nrhs + j]; B[p * nrhs + j] = t; }
        }
        for (int i = k + 1; i < n; ++i) {
            double f = A[i * n + k] / A[k * n + k];
            if (f == 0.0) continue;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] -= f * B[k * nrhs + j];
        }
    }
    for (int k = n - 1; k >= 0; --k)
        for (int j = 0; j < nrhs; ++j) {
            double s = B[k * nrhs + j];
            for (int i = k + 1; i < n; ++i) s -= A[k * n + i] * B[i * nrhs + j];
            B[k * nrhs + j] = s / A[k * n + k];
        }
    return 0;
}

/* Symmetric eigendecomposition, cyclic two-sided Jacobi.  A (n x n row-major, symmetric)
 * is destroyed; w ascending (LAPACK eigh order), V columns = eigenvectors. n <= 16. */
void orc_eigh(int n, double *A, double *w, double *V)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < n; ++i) {
            dg += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-34 * (dg + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
    /* selection sort ascending, permuting columns of V */
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { t = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = t; }
        }
    }
}

/* U Vh of the SVD of a 3x3 matrix M (row-major): the orthogonal factor the reference
 * forms as U @ Vh (cvxpnpl.py:510-511).  No determinant correction, like the reference. */
void orc_svd3_uvh(const double *M, double *R)
{
    double MtM[9], w[3], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += M[k * 3 + i] * M[k * 3 + j];
            MtM[i * 3 + j] = s;
        }
    orc_eigh(3, MtM, w, V); /* ascending: w[2] largest */
    double U[9];
    /* u_j = M v_j / sigma_j for the two largest, third by cross product with the sign
     * that makes M v_3 . u_3 >= 0 (so that U S Vh = M with S >= 0). */
    for (int jj = 0; jj < 2; ++jj) {
        int j = 2 - jj;
        double u[3], nrm = 0;
        for (int i = 0; i < 3; ++i) {
            u[i] = M[i * 3 + 0] * V[0 * 3 + j] + M[i * 3 + 1] * V[1 * 3 + j] + M[i * 3 + 2] * V[2 * 3 + j];
        }
        if (jj == 1) { /* Gram-Schmidt against the first */
            double d = u[0] * U[0 * 3 + 2] + u[1] * U[1 * 3 + 2] + u[2] * U[2 * 3 + 2];
            for (int i = 0; i < 3; ++i) u[i] -= d * U[i * 3 + 2];
        }
        for (int i = 0; i < 3; ++i) nrm += u[i] * u[i];
        nrm = sqrt(nrm);
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = u[i] / nrm;
    }
    double a[3] = {U[0 * 3 + 2], U[1 * 3 + 2], U[2 * 3 + 2]}, b[3] = {U[0 * 3 + 1], U[1 * 3 + 1], U[2 * 3 + 1]};
    double c3[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    double mv[3], dot = 0;
    for (int i = 0; i < 3; ++i) {
        mv[i] = M[i * 3 + 0] * V[0 * 3 + 0] + M[i * 3 + 1] * V[1 * 3 + 0] + M[i * 3 + 2] * V[2 * 3 + 0];
        dot += mv[i] * c3[i];
    }
    double sg = dot < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i) U[i * 3 + 0] = sg * c3[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[i * 3 + k] * V[j * 3 + k];
            R[i * 3 + j] = s;
        }
}

/* ------------------------------------------------------------------ constraint rows */

/* bearings p = K^-1 [u v 1]^T via a general solve, cvxpnpl.py:37 / :123-126 */
static int bearings(int n, const double *uv, const double *K, double *p /* n x 3 */)
{
    double *rhs = (double *)malloc(sizeof(double) * 3 * (n > 0 ? n : 1));
    double Kc[9];
    memcpy(Kc, K, sizeof(Kc));
    for (int i = 0; i < n; ++i) { rhs[0 * n + i] = uv[2 * i]; rhs[1 * n + i] = uv[2 * i + 1]; rhs[2 * n + i] = 1.0; }
    int rc = n > 0 ? lu_solve(3, Kc, n, rhs) : 0;
    for (int i = 0; i < n; ++i) { p[3 * i] = rhs[0 * n + i]; p[3 * i + 1] = rhs[1 * n + i]; p[3 * i + 2] = rhs[2 * n + i]; }
    free(rhs);
    return rc;
}

/* cvxpnpl.py:20-104.  C is (3n x 9) stacked [C1;C2;C3], N is (3n x 3) stacked [N1;N2;N3]
 * -- the order pnp() stacks them in (cvxpnpl.py:545-546). */
int orc_point_constraints(int n, const double *pts_2d, const double *pts_3d, const double *K, double *C, double *N)
{
    double *p = (double *)malloc(sizeof(double) * 3 * (n > 0 ? n : 1));
    int rc = bearings(n, pts_2d, K, p);
    for (int i = 0; i < n; ++i) {
        double px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
        double X = pts_3d[3 * i], Y = pts_3d[3 * i + 1], Z = pts_3d[3 * i + 2];
        double *c1 = C + (size_t)(0 * n + i) * 9, *c2 = C + (size_t)(1 * n + i) * 9, *c3 = C + (size_t)(2 * n + i) * 9;
        double *n1 = N + (size_t)(0 * n + i) * 3, *n2 = N + (size_t)(1 * n + i) * 3, *n3 = N + (size_t)(2 * n + i) * 3;
        c1[0] = 0;       c1[1] = -X * pz; c1[2] = X * py;  c1[3] = 0;       c1[4] = -Y * pz; c1[5] = Y * py;  c1[6] = 0;       c1[7] = -Z * pz; c1[8] = Z * py;
        c2[0] = X * pz;  c2[1] = 0;       c2[2] = -X * px; c2[3] = Y * pz;  c2[4] = 0;       c2[5] = -Y * px; c2[6] = Z * pz;  c2[7] = 0;       c2[8] = -Z * px;
        c3[0] = -X * py; c3[1] = X * px;  c3[2] = 0;       c3[3] = -Y * py; c3[4] = Y * px;  c3[5] = 0;       c3[6] = -Z * py; c3[7] = Z * px;  c3[8] = 0;
        n1[0] = 0;   n1[1] = -pz; n1[2] = py;
        n2[0] = pz;  n2[1] = 0;   n2[2] = -px;
        n3[0] = -py; n3[1] = px;  n3[2] = 0;
    }
    free(p);
    return rc;
}

/* cvxpnpl.py:107-153.  line_2d (n,2,2), line_3d (n,2,3); C (2n x 9), N (2n x 3). */
int orc_line_constraints(int n, const double *line_2d, const double *line_3d, const double *K, double *C, double *N)
{
    double *l = (double *)malloc(sizeof(double) * 6 * (n > 0 ? n : 1));
    int rc = bearings(2 * n, line_2d, K, l);
    for (int i = 0; i < n; ++i) {
        const double *a = l + 6 * i, *b = l + 6 * i + 3;
        double nx = a[1] * b[2] - a[2] * b[1], ny = a[2] * b[0] - a[0] * b[2], nz = a[0] * b[1] - a[1] * b[0];
        double nrm = sqrt(nx * nx + ny * ny + nz * nz);
        nx /= nrm; ny /= nrm; nz /= nrm;
        for (int e = 0; e < 2; ++e) {
            const double *P = line_3d + (size_t)(2 * i + e) * 3;
            double *c = C + (size_t)(2 * i + e) * 9, *nn = N + (size_t)(2 * i + e) * 3;
            c[0] = P[0] * nx; c[1] = P[0] * ny; c[2] = P[0] * nz;
            c[3] = P[1] * nx; c[4] = P[1] * ny; c[5] = P[1] * nz;
            c[6] = P[2] * nx; c[7] = P[2] * ny; c[8] = P[2] * nz;
            nn[0] = nx; nn[1] = ny; nn[2] = nz;
        }
    }
    free(l);
    return rc;
}

/* cvxpnpl.py:548-549: B = solve(N^T N, N^T C) (3x9), A = C - N B (m x 9) */
int orc_eliminate(int m, const double *C, const double *N, double *B, double *A)
{
    double NtN[9] = {0}, NtC[27] = {0};
    for (int r = 0; r < m; ++r)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) NtN[i * 3 + j] += N[r * 3 + i] * N[r * 3 + j];
            for (int j = 0; j < 9; ++j) NtC[i * 9 + j] += N[r * 3 + i] * C[r * 9 + j];
        }
    int rc = lu_solve(3, NtN, 9, NtC);
    memcpy(B, NtC, sizeof(double) * 27);
    for (int r = 0; r < m; ++r)
        for (int j = 0; j < 9; ++j)
            A[r * 9 + j] = C[r * 9 + j] - (N[r * 3] * B[j] + N[r * 3 + 1] * B[9 + j] + N[r * 3 + 2] * B[18 + j]);
    return rc;
}

/* ------------------------------------------------------------------ vech */

/* cvxpnpl.py:346-370: columns of the lower triangle, off-diagonals times `scale` */
void orc_vech10(const double *A, double scale, double *v)
{
    int k = 0;
    for (int j = 0; j < 10; ++j)
        for (int i = j; i < 10; ++i) v[k++] = (i == j ? 1.0 : scale) * A[i * 10 + j];
}

/* cvxpnpl.py:373-384 */
void orc_vech10_inv(const double *v, double *A)
{
    int k = 0;
    for (int j = 0; j < 10; ++j)
        for (int i = j; i < 10; ++i) { A[i * 10 + j] = v[k]; A[j * 10 + i] = v[k]; ++k; }
}

/* ------------------------------------------------------------------ static SDP data */

static void sym_vech2(const double *P, double *row)
{
    double S[100];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) S[i * 10 + j] = 0.5 * (P[i * 10 + j] + P[j * 10 + i]);
    orc_vech10(S, 2.0, row);
}

/* cvxpnpl.py:387-451: dense 77x55 matrix (row-major) and b (77). */
void orc_sdp_constraints(double *Ad, double *b)
{
    memset(Ad, 0, sizeof(double) * NM * NV);
    memset(b, 0, sizeof(double) * NM);
    Ad[0 * NV + 54] = 1.0; /* Z[9,9] = 1 */
    static const int rc_ab[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    static const double rc_c[6] = {1, 0, 0, 1, 0, 1};
    for (int i = 0; i < 6; ++i) {
        int a = rc_ab[i][0], bb = rc_ab[i][1];
        double P[100];
        /* kron(I3, E_ab^T): entries (3k+b, 3k+a) -- rows of R (R R^T = I) */
        memset(P, 0, sizeof(P));
        for (int k = 0; k < 3; ++k) P[(3 * k + bb) * 10 + (3 * k + a)] = 1.0;
        P[99] = -rc_c[i];
        sym_vech2(P, Ad + (size_t)(i + 1) * NV);
        /* kron(E_ab, I3): entries (3a+k, 3b+k) -- columns of R (R^T R = I) */
        memset(P, 0, sizeof(P));
        for (int k = 0; k < 3; ++k) P[(3 * a + k) * 10 + (3 * bb + k)] = 1.0;
        P[99] = -rc_c[i];
        sym_vech2(P, Ad + (size_t)(i + 7) * NV);
    }
    /* determinant rows: kron(E_ab, [e_l]x) on the 9x9 block, -kron(e_k, e_l) in row 9 */
    static const int det_ab[3][2] = {{1, 0}, {2, 1}, {0, 2}};
    static const int det_k[3] = {2, 0, 1};
    for (int g = 0; g < 3; ++g)
        for (int l = 0; l < 3; ++l) {
            double P[100], e[3] = {0, 0, 0};
            memset(P, 0, sizeof(P));
            e[l] = 1.0;
            double S[9] = {0, -e[2], e[1], e[2], 0, -e[0], -e[1], e[0], 0};
            int a = det_ab[g][0], bb = det_ab[g][1];
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) P[(3 * a + p) * 10 + (3 * bb + q)] = S[p * 3 + q];
            P[9 * 10 + (3 * det_k[g] + l)] = -1.0;
            sym_vech2(P, Ad + (size_t)(13 + 3 * g + l) * NV);
        }
    /* cone block: -vech(ones, sqrt 2) on the diagonal of rows 22.. */
    double ones[100], d[NV];
    for (int i = 0; i < 100; ++i) ones[i] = 1.0;
    orc_vech10(ones, sqrt(2.0), d);
    for (int k = 0; k < NV; ++k) Ad[(size_t)(22 + k) * NV + k] = -d[k];
    b[0] = 1.0;
}

/* ------------------------------------------------------------------ SCS restated */

/* projection of an svec-scaled 55-vector onto the PSD cone (SCS's "s" cone) */
static void proj_psd_svec(double *s)
{
    const double is2 = 1.0 / sqrt(2.0), s2 = sqrt(2.0);
    double M[100], w[10], V[100];
    int k = 0;
    for (int j = 0; j < 10; ++j)
        for (int i = j; i < 10; ++i) {
            double v = (i == j) ? s[k] : s[k] * is2;
            M[i * 10 + j] = v; M[j * 10 + i] = v; ++k;
        }
    orc_eigh(10, M, w, V);
    k = 0;
    for (int j = 0; j < 10; ++j)
        for (int i = j; i < 10; ++i) {
            double acc = 0;
            for (int e = 0; e < 10; ++e) if (w[e] > 0) acc += w[e] * V[i * 10 + e] * V[j * 10 + e];
            s[k++] = (i == j) ? acc : acc * s2;
        }
}

static int chol(int n, double *A) /* lower, in place */
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(int n, const double *L, double *x)
{
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

typedef struct {
    double A[NM * NV], b[NM];
    double L[NV * NV];   /* chol(I + A^T A) */
    int nm, nz;          /* rows in use, rows of the zero cone (77 / 22, or 71 / 16 for the rc variant) */
    int ready;
} scs_static_t;
static scs_static_t g_scs_var[2];
#define g_scs g_scs_var[0]

/* benchmarks/toolkit/methods/rc.py:9-64: the same construction as cvxpnpl.py:387-451 without the six rows
 * kron(I3, E_ij^T) (row orthonormality): Z99 = 1, the six column rows kron(E_ij, I3), the nine determinant rows,
 * then the cone block -- 71 x 55 (row-major), b (71). */
void orc_sdp_constraints_rc(double *Ad, double *b)
{
    static double full[NM * NV], fb[NM];
    orc_sdp_constraints(full, fb);
    memset(Ad, 0, sizeof(double) * 71 * NV);
    memset(b, 0, sizeof(double) * 71);
    memcpy(Ad, full, sizeof(double) * NV);                                        /* rc.py:24   Ad[0, -1] = 1        */
    for (int i = 0; i < 6; ++i) memcpy(Ad + (size_t)(1 + i) * NV, full + (size_t)(7 + i) * NV, sizeof(double) * NV);  /* rc.py:27-36 */
    for (int i = 0; i < 9; ++i) memcpy(Ad + (size_t)(7 + i) * NV, full + (size_t)(13 + i) * NV, sizeof(double) * NV); /* rc.py:39-53 */
    for (int k = 0; k < NV; ++k) Ad[(size_t)(16 + k) * NV + k] = full[(size_t)(22 + k) * NV + k];                       /* rc.py:56-57 */
    b[0] = 1.0;                                                                                                       /* rc.py:63    */
}

static void scs_static_init_var(int variant);
static void scs_static_init(void) { scs_static_init_var(0); }
static void scs_static_init_var(int variant)
{
#undef g_scs
#define g_scs g_scs_var[variant]
    if (g_scs.ready) return;
#ifdef _OPENMP
#pragma omp critical(orc_scs_init)
#endif
    {
        if (!g_scs.ready) {
            static scs_static_t tmp;
            memset(&tmp, 0, sizeof(tmp));
            if (variant) { orc_sdp_constraints_rc(tmp.A, tmp.b); tmp.nm = 71; tmp.nz = 16; }
            else { orc_sdp_constraints(tmp.A, tmp.b); tmp.nm = NM; tmp.nz = 22; }
            for (int i = 0; i < NV; ++i)
                for (int j = 0; j < NV; ++j) {
                    double s = (i == j);
                    for (int r = 0; r < tmp.nm; ++r) s += tmp.A[r * NV + i] * tmp.A[r * NV + j];
                    tmp.L[i * NV + j] = s;
                }
            chol(NV, tmp.L);
            memcpy(g_scs.A, tmp.A, sizeof(tmp.A));
            memcpy(g_scs.b, tmp.b, sizeof(tmp.b));
            memcpy(g_scs.L, tmp.L, sizeof(tmp.L));
            g_scs.nm = tmp.nm; g_scs.nz = tmp.nz;
#ifdef _OPENMP
#pragma omp flush
#endif
            g_scs.ready = 1;
        }
    }
#undef g_scs
#define g_scs g_scs_var[0]
}

/* [x;y] = M^-1 [wx; wy],  M = [[I, A^T], [-A, I]]:  x = (I + A^T A)^-1 (wx - A^T wy), y = wy + A x */
static void solve_M(const scs_static_t *G, const double *wx, const double *wy, double *x, double *y)
{
    const double *A = G->A;
    const int NMv = G->nm;
    for (int j = 0; j < NV; ++j) {
        double s = wx[j];
        for (int r = 0; r < NMv; ++r) s -= A[r * NV + j] * wy[r];
        x[j] = s;
    }
    chol_solve(NV, G->L, x);
    for (int r = 0; r < NMv; ++r) {
        double s = wy[r];
        for (int j = 0; j < NV; ++j) s += A[r * NV + j] * x[j];
        y[r] = s;
    }
}

/*
 * min c^T x  s.t.  A x + s = b,  s in {0}^22 x PSD_10      (cvxpnpl.py:485-489)
 * Returns the SCS fields the reference reads: x (55) and info.dobj = -b^T y; plus
 * iterations used and the three residuals.  `scale` rescales (b, c) -> (b/scale*?): we
 * apply SCS's primal/dual balance by solving the problem with c' = c * cscale, which
 * leaves x unchanged and multiplies y, dobj by cscale (undone on return).
 */
int orc_scs_solve_var(int variant, const double *c_in, double eps, int max_iters, double cscale, double *x_out, double *y_out,
                      double *dobj, double *pobj, int *iters_out, double *res_out)
{
    scs_static_init_var(variant);
    const scs_static_t *G = &g_scs_var[variant];
    const double *A = G->A, *b = G->b;
    const int NMv = G->nm, NZv = G->nz; /* rows, zero-cone rows: 77 / 22 (cvxpnpl.py:448) or 71 / 16 (rc.py:91) */
    const double alpha = 1.5;
    double c[NV];
    for (int j = 0; j < NV; ++j) c[j] = c_in[j] * cscale;
    /* M^-1 h, h = [c; b] */
    double ghx[NV], ghy[NM];
    solve_M(G, c, b, ghx, ghy);
    double hgh = 0;
    for (int j = 0; j < NV; ++j) hgh += c[j] * ghx[j];
    for (int r = 0; r < NMv; ++r) hgh += b[r] * ghy[r];
    double nb = 0, nc = 0;
    for (int r = 0; r < NMv; ++r) nb += b[r] * b[r];
    for (int j = 0; j < NV; ++j) nc += c[j] * c[j];
    nb = sqrt(nb); nc = sqrt(nc);

    double ux[NV] = {0}, uy[NM] = {0}, ut = 1.0;      /* u = (x, y, tau) */
    double vs[NM] = {0}, vk = 1.0;                     /* v = (0, s, kappa) */
    double tx[NV], ty[NM], tt, wx[NV], wy[NM], wt;
    int it, status = 1;
    double pres = 0, dres = 0, gap = 0;
    for (it = 1; it <= max_iters; ++it) {
        /* u~ = (I+Q)^-1 (u + v) */
        for (int j = 0; j < NV; ++j) wx[j] = ux[j];
        for (int r = 0; r < NMv; ++r) wy[r] = uy[r] + vs[r];
        wt = ut + vk;
        solve_M(G, wx, wy, tx, ty);
        double hmw = 0;
        for (int j = 0; j < NV; ++j) hmw += c[j] * tx[j];
        for (int r = 0; r < NMv; ++r) hmw += b[r] * ty[r];
        tt = (wt + hmw) / (1.0 + hgh);
        for (int j = 0; j < NV; ++j) tx[j] -= ghx[j] * tt;
        for (int r = 0; r < NMv; ++r) ty[r] -= ghy[r] * tt;
        /* u = Pi_C(alpha u~ + (1-alpha) u - v), v += u - alpha u~ - (1-alpha) u_old */
        for (int j = 0; j < NV; ++j) ux[j] = alpha * tx[j] + (1 - alpha) * ux[j]; /* free cone, v_x = 0 */
        double ry[NM], un[NM];
        for (int r = 0; r < NMv; ++r) { ry[r] = alpha * ty[r] + (1 - alpha) * uy[r]; un[r] = ry[r] - vs[r]; }
        proj_psd_svec(un + NZv); /* dual cone: free (22 / 16) x PSD */
        for (int r = 0; r < NMv; ++r) { vs[r] = vs[r] - ry[r] + un[r]; uy[r] = un[r]; }
        for (int r = 0; r < NZv; ++r) vs[r] = 0.0; /* s in the zero cone */
        double rt = alpha * tt + (1 - alpha) * ut, utn = rt - vk;
        if (utn < 0) utn = 0;
        vk = vk - rt + utn; ut = utn;

        if (ut > 1e-12) {
            double px = 0, py = 0, rp = 0, rd = 0;
            for (int r = 0; r < NMv; ++r) {
                double s = vs[r] / ut - b[r];
                for (int j = 0; j < NV; ++j) s += A[r * NV + j] * ux[j] / ut;
                rp += s * s;
                py += b[r] * uy[r] / ut;
            }
            for (int j = 0; j < NV; ++j) {
                double s = c[j];
                for (int r = 0; r < NMv; ++r) s += A[r * NV + j] * uy[r] / ut;
                rd += s * s;
                px += c[j] * ux[j] / ut;
            }
            pres = sqrt(rp) / (1 + nb); dres = sqrt(rd) / (1 + nc);
            gap = fabs(px + py) / (1 + fabs(px) + fabs(py));
            if (pres < eps && dres < eps && gap < eps) { status = 0; break; }
        }
    }
    if (it > max_iters) it = max_iters;
    double px = 0, py = 0;
    for (int j = 0; j < NV; ++j) { x_out[j] = ut > 0 ? ux[j] / ut : NAN; px += c_in[j] * x_out[j]; }
    for (int r = 0; r < NMv; ++r) { double y = ut > 0 ? uy[r] / ut / cscale : NAN; if (y_out) y_out[r] = y; py += b[r] * y; }
    if (dobj) *dobj = -py;
    if (pobj) *pobj = px;
    if (iters_out) *iters_out = it;
    if (res_out) { res_out[0] = pres; res_out[1] = dres; res_out[2] = gap; }
    return status;
}

int orc_scs_solve(const double *c_in, double eps, int max_iters, double cscale, double *x_out, double *y_out,
                  double *dobj, double *pobj, int *iters_out, double *res_out)
{
    return orc_scs_solve_var(0, c_in, eps, max_iters, cscale, x_out, y_out, dobj, pobj, iters_out, res_out);
}

/* ------------------------------------------------------------------ rank > 1 recovery */

/* real parts of the roots of p[0] x^4 + p[1] x^3 + p[2] x^2 + p[3] x + p[4], the way
 * np.roots does it: eigenvalues of the companion matrix (here by shifted QR on the
 * Hessenberg companion; cvxpnpl.py:185-186 keeps Re() of complex roots too).
 * Output order follows decreasing real part then decreasing imaginary part; callers
 * must not depend on np.roots' order. */
static void hqr4(double *H, int n, double *wr, double *wi)
{
    /* Francis double-shift QR for a small upper Hessenberg matrix (EISPACK hqr, 0-based) */
    int nn = n - 1, its = 0;
    double t = 0, anorm = 0;
    for (int i = 0; i < n; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += fabs(H[i * n + j]);
    while (nn >= 0) {
        int l;
        double p = 0, q = 0, r = 0, s, x, y, z, w;
        for (;;) {
            for (l = nn; l >= 1; --l) {
                s = fabs(H[(l - 1) * n + l - 1]) + fabs(H[l * n + l]);
                if (s == 0.0) s = anorm;
                if (fabs(H[l * n + l - 1]) + s == s) { H[l * n + l - 1] = 0.0; break; }
            }
            x = H[nn * n + nn];
            if (l == nn) { wr[nn] = x + t; wi[nn--] = 0.0; its = 0; break; }
            y = H[(nn - 1) * n + nn - 1];
            w = H[nn * n + nn - 1] * H[(nn - 1) * n + nn];
            if (l == nn - 1) {
                p = 0.5 * (y - x); q = p * p + w; z = sqrt(fabs(q)); x += t;
                if (q >= 0.0) {
                    z = p + (p >= 0 ? fabs(z) : -fabs(z));
                    wr[nn - 1] = wr[nn] = x + z;
                    if (z != 0.0) wr[nn] = x - w / z;
                    wi[nn - 1] = wi[nn] = 0.0;
                } else {
                    wr[nn - 1] = wr[nn] = x + p;
                    wi[nn - 1] = -(wi[nn] = z);
                }
                nn -= 2; its = 0; break;
            }
            if (its == 60) { for (int i = 0; i <= nn; ++i) { wr[i] = NAN; wi[i] = NAN; } return; }
            if (its == 10 || its == 20) {
                t += x;
                for (int i = 0; i <= nn; ++i) H[i * n + i] -= x;
                s = fabs(H[nn * n + nn - 1]) + fabs(H[(nn - 1) * n + nn - 2]);
                y = x = 0.75 * s; w = -0.4375 * s * s;
            }
            ++its;
            int m;
            for (m = nn - 2; m >= l; --m) {
                z = H[m * n + m]; r = x - z; s = y - z;
                p = (r * s - w) / H[(m + 1) * n + m] + H[m * n + m + 1];
                q = H[(m + 1) * n + m + 1] - z - r - s;
                r = H[(m + 2) * n + m + 1];
                s = fabs(p) + fabs(q) + fabs(r);
                p /= s; q /= s; r /= s;
                if (m == l) break;
                double u = fabs(H[m * n + m - 1]) * (fabs(q) + fabs(r));
                double v = fabs(p) * (fabs(H[(m - 1) * n + m - 1]) + fabs(z) + fabs(H[(m + 1) * n + m + 1]));
                if (u + v == v) break;
            }
            for (int i = m + 2; i <= nn; ++i) { H[i * n + i - 2] = 0.0; if (i != m + 2) H[i * n + i - 3] = 0.0; }
            for (int k = m; k <= nn - 1; ++k) {
                if (k != m) {
                    p = H[k * n + k - 1]; q = H[(k + 1) * n + k - 1]; r = 0.0;
                    if (k != nn - 1) r = H[(k + 2) * n + k - 1];
                    if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                }
                double sg = sqrt(p * p + q * q + r * r);
                s = p >= 0 ? sg : -sg;
                if (s != 0.0) {
                    if (k == m) { if (l != m) H[k * n + k - 1] = -H[k * n + k - 1]; }
                    else H[k * n + k - 1] = -s * x;
                    p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                    for (int j = k; j <= nn; ++j) {
                        p = H[k * n + j] + q * H[(k + 1) * n + j];
                        if (k != nn - 1) { p += r * H[(k + 2) * n + j]; H[(k + 2) * n + j] -= p * z; }
                        H[(k + 1) * n + j] -= p * y; H[k * n + j] -= p * x;
                    }
                    int mmin = nn < k + 3 ? nn : k + 3;
                    for (int i = l; i <= mmin; ++i) {
                        p = x * H[i * n + k] + y * H[i * n + k + 1];
                        if (k != nn - 1) { p += z * H[i * n + k + 2]; H[i * n + k + 2] -= p * r; }
                        H[i * n + k + 1] -= p * q; H[i * n + k] -= p;
                    }
                }
            }
        }
    }
}

int orc_poly_roots_real(int deg, const double *p, double *re)
{
    /* strip leading zeros like np.roots */
    int lead = 0;
    while (lead < deg && p[lead] == 0.0) ++lead;
    int n = deg - lead;
    if (n <= 0) return 0;
    int trail = 0;
    while (trail < n && p[deg - trail] == 0.0) ++trail; /* zero roots */
    int m = n - trail;
    double H[16] = {0}, wr[4], wi[4];
    for (int j = 0; j < m; ++j) H[0 * m + j] = -p[lead + 1 + j] / p[lead];
    for (int i = 1; i < m; ++i) H[i * m + i - 1] = 1.0;
    if (m > 0) hqr4(H, m, wr, wi);
    for (int i = 0; i < m; ++i) re[i] = wr[i];
    for (int i = m; i < n; ++i) re[i] = 0.0;
    /* sort: decreasing real part */
    for (int i = 0; i < n - 1; ++i)
        for (int j = i + 1; j < n; ++j)
            if (re[j] > re[i]) { double t = re[i]; re[i] = re[j]; re[j] = t; }
    return n;
}

/* polynomial helpers on coefficient arrays in ascending powers of a (degree <= 4) */
typedef struct { double c[5]; } poly_t;
static poly_t pmul(poly_t x, poly_t y)
{
    poly_t r = {{0, 0, 0, 0, 0}};
    for (int i = 0; i < 5; ++i)
        for (int j = 0; i + j < 5; ++j) r.c[i + j] += x.c[i] * y.c[j];
    return r;
}
static poly_t padd(poly_t x, poly_t y, double s) { poly_t r; for (int i = 0; i < 5; ++i) r.c[i] = x.c[i] + s * y.c[i]; return r; }

/*
 * cvxpnpl.py:156-218 (E6Q3).  A is (N x 10), columns [a^2,b^2,c^2,ab,ac,bc,a,b,c,1].
 * The reference reduces to [I6 | G] by least squares (:163-165), keeps the rows for
 * b^2, c^2, bc: with D = -G[[1,2,5]] (:168)
 *      b^2 = D0.[a,b,c,1]   c^2 = D1.[a,b,c,1]   bc = D2.[a,b,c,1]
 * and eliminates b, c through the identities  b(bc) = c(b^2),  c(bc) = b(c^2),
 * (b^2)(c^2) = (bc)^2, each of which reduces to a form linear in (b, c, 1) whose
 * coefficients are polynomials in a:  M(a) [b, c, 1]^T = 0  (:190-203).  det M(a) is the
 * quartic of :176-181.  The reference ships that determinant expanded symbolically;
 * here the same M(a) is derived from the identities and its determinant is expanded by
 * polynomial arithmetic at run time, so no generated expression is reproduced.
 * Then b, c per root by least squares on the first two columns of M (:206-216).
 */
int orc_re6q3(int nrows, const double *A, double *a_out, double *b_out, double *c_out)
{
    /* G = solve(B^T B, B^T C), B = A[:, :6], C = A[:, 6:] */
    double BtB[36] = {0}, BtC[24] = {0};
    for (int r = 0; r < nrows; ++r)
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) BtB[i * 6 + j] += A[r * 10 + i] * A[r * 10 + j];
            for (int j = 0; j < 4; ++j) BtC[i * 4 + j] += A[r * 10 + i] * A[r * 10 + 6 + j];
        }
    if (lu_solve(6, BtB, 4, BtC)) return -1;
    double d[3][4];
    static const int rows[3] = {1, 2, 5};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) d[i][j] = -BtC[rows[i] * 4 + j];
    /* linear forms in (b, c, 1) with polynomial-in-a coefficients:
     *   b^2 = P.b + Q.c + (d00 a + d03)   with P = d01, Q = d02
     *   c^2 = d11 b + d12 c + (d10 a + d13)
     *   bc  = d21 b + d22 c + (d20 a + d23)                                            */
    poly_t Z = {{0, 0, 0, 0, 0}};
    #define CST(v) ((poly_t){{(v), 0, 0, 0, 0}})
    #define LIN(c0, c1) ((poly_t){{(c0), (c1), 0, 0, 0}})
    /* a quadratic-monomial vector [b^2, c^2, bc] -> rows of coefficients over (b, c, 1) */
    poly_t q[3][3] = {
        {CST(d[0][1]), CST(d[0][2]), LIN(d[0][3], d[0][0])},
        {CST(d[1][1]), CST(d[1][2]), LIN(d[1][3], d[1][0])},
        {CST(d[2][1]), CST(d[2][2]), LIN(d[2][3], d[2][0])}};
    /* multiply a linear form L = l0 b + l1 c + l2 by b or by c and reduce again:
     *   b * L = l0 b^2 + l1 bc + l2 b ;  c * L = l0 bc + l1 c^2 + l2 c                 */
    poly_t M[3][3];
    /* identity 1:  b*(bc) - c*(b^2) = 0 */
    {
        poly_t *L = q[2], *K = q[0];
        for (int j = 0; j < 3; ++j) {
            poly_t t = padd(pmul(L[0], q[0][j]), pmul(L[1], q[2][j]), 1.0);       /* b*(bc) quad part */
            poly_t u = padd(pmul(K[0], q[2][j]), pmul(K[1], q[1][j]), 1.0);       /* c*(b^2) quad part */
            M[0][j] = padd(t, u, -1.0);
        }
        M[0][0] = padd(M[0][0], L[2], 1.0);   /* + l2 b */
        M[0][1] = padd(M[0][1], K[2], -1.0);  /* - k2 c */
    }
    /* identity 2:  c*(bc) - b*(c^2) = 0 */
    {
        poly_t *L = q[2], *K = q[1];
        for (int j = 0; j < 3; ++j) {
            poly_t t = padd(pmul(L[0], q[2][j]), pmul(L[1], q[1][j]), 1.0);       /* c*(bc) */
            poly_t u = padd(pmul(K[0], q[0][j]), pmul(K[1], q[2][j]), 1.0);       /* b*(c^2) */
            M[1][j] = padd(t, u, -1.0);
        }
        M[1][1] = padd(M[1][1], L[2], 1.0);
        M[1][0] = padd(M[1][0], K[2], -1.0);
    }
    /* identity 3:  (b^2)(c^2) - (bc)^2 = 0.  Product of two linear forms
     *   (x0 b + x1 c + x2)(y0 b + y1 c + y2) = x0y0 b^2 + x1y1 c^2 + (x0y1+x1y0) bc
     *                                          + (x0y2+x2y0) b + (x1y2+x2y1) c + x2y2  */
    {
        poly_t *X = q[0], *Y = q[1], *W = q[2];
        poly_t cb2 = padd(pmul(X[0], Y[0]), pmul(W[0], W[0]), -1.0);
        poly_t cc2 = padd(pmul(X[1], Y[1]), pmul(W[1], W[1]), -1.0);
        poly_t cbc = padd(padd(pmul(X[0], Y[1]), pmul(X[1], Y[0]), 1.0), pmul(W[0], W[1]), -2.0);
        poly_t lb = padd(padd(pmul(X[0], Y[2]), pmul(X[2], Y[0]), 1.0), pmul(W[0], W[2]), -2.0);
        poly_t lc = padd(padd(pmul(X[1], Y[2]), pmul(X[2], Y[1]), 1.0), pmul(W[1], W[2]), -2.0);
        poly_t l1 = padd(pmul(X[2], Y[2]), pmul(W[2], W[2]), -1.0);
        for (int j = 0; j < 3; ++j)
            M[2][j] = padd(padd(pmul(cb2, q[0][j]), pmul(cc2, q[1][j]), 1.0), pmul(cbc, q[2][j]), 1.0);
        M[2][0] = padd(M[2][0], lb, 1.0);
        M[2][1] = padd(M[2][1], lc, 1.0);
        M[2][2] = padd(M[2][2], l1, 1.0);
    }
    (void)Z;
    /* det M(a) by cofactor expansion in polynomial arithmetic */
    poly_t det = pmul(M[0][0], padd(pmul(M[1][1], M[2][2]), pmul(M[1][2], M[2][1]), -1.0));
    det = padd(det, pmul(M[0][1], padd(pmul(M[1][0], M[2][2]), pmul(M[1][2], M[2][0]), -1.0)), -1.0);
    det = padd(det, pmul(M[0][2], padd(pmul(M[1][0], M[2][1]), pmul(M[1][1], M[2][0]), -1.0)), 1.0);
    double p[5] = {det.c[4], det.c[3], det.c[2], det.c[1], det.c[0]};
    double roots[4];
    int nr = orc_poly_roots_real(4, p, roots);
    for (int k = 0; k < nr; ++k) {
        double a = roots[k], Mv[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0, pw = 1;
                for (int e = 0; e < 5; ++e) { s += M[i][j].c[e] * pw; pw *= a; }
                Mv[i][j] = s;
            }
        /* [b c] = -solve(M2^T M2, M2^T m3), M2 = first two columns */
        double G[4] = {0, 0, 0, 0}, g[2] = {0, 0};
        for (int i = 0; i < 3; ++i) {
            G[0] += Mv[i][0] * Mv[i][0]; G[1] += Mv[i][0] * Mv[i][1];
            G[3] += Mv[i][1] * Mv[i][1];
            g[0] += Mv[i][0] * Mv[i][2]; g[1] += Mv[i][1] * Mv[i][2];
        }
        G[2] = G[1];
        double dt = G[0] * G[3] - G[1] * G[2];
        a_out[k] = a;
        b_out[k] = -(G[3] * g[0] - G[1] * g[1]) / dt;
        c_out[k] = -(G[0] * g[1] - G[2] * g[0]) / dt;
    }
    return nr;
}

/*
 * cvxpnpl.py:221-343.  vecs (10x10 row-major, columns = eigenvectors ascending),
 * rank = #eigenvalues > 1e-3.  Output r_c (k x 9), k = 2 or 4; returns k (or -1).
 */
int orc_constraint_ortho_det(const double *vecs, int rank, double *rc_out)
{
    int rk = 2 * ((rank + 1) / 2);
    if (rk > 4) rk = 4;
    if (rk != 2 && rk != 4) return -1;
    /* V rows = last rk eigenvectors (:234); v0 = V[-1]/V[-1,-1]; marginalise (:235-236) */
    double Vr[4][10], v0[10], V[9][4];
    for (int a = 0; a < rk; ++a)
        for (int i = 0; i < 10; ++i) Vr[a][i] = vecs[i * 10 + (10 - rk + a)];
    for (int i = 0; i < 10; ++i) v0[i] = Vr[rk - 1][i] / Vr[rk - 1][9];
    for (int i = 0; i < 9; ++i) {
        for (int a = 0; a < rk - 1; ++a) V[i][a] = Vr[a][i] - Vr[a][9] * v0[i];
        V[i][rk - 1] = v0[i];
    }
    /* 21 quadratic forms P (rk x rk): 6 column, 6 row, 9 determinant (:239-301) */
    double P[21][4][4];
    memset(P, 0, sizeof(P));
    int k = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            /* Vci = rows 3i..3i+2 of V (column i of R), Vri = rows i, i+3, i+6 (row i of R) */
            for (int a = 0; a < rk; ++a)
                for (int b = 0; b < rk; ++b) {
                    double sc = 0, sr = 0;
                    for (int e = 0; e < 3; ++e) {
                        sc += V[3 * i + e][a] * V[3 * j + e][b];
                        sr += V[3 * e + i][a] * V[3 * e + j][b];
                    }
                    if (a == rk - 1 && b == rk - 1 && i == j) { sc -= 1.0; sr -= 1.0; }
                    P[k][a][b] += 0.5 * sc; P[k][b][a] += 0.5 * sc;
                    P[6 + k][a][b] += 0.5 * sr; P[6 + k][b][a] += 0.5 * sr;
                }
            ++k;
        }
    static const int cyc[3][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}};
    int m = 0;
    for (int g = 0; g < 3; ++g)
        for (int l = 0; l < 3; ++l) {
            int i = cyc[g][0], j = cyc[g][1], kk = cyc[g][2];
            double e[3] = {0, 0, 0};
            e[l] = 1.0;
            double S[3][3] = {{0, -e[2], e[1]}, {e[2], 0, -e[0]}, {-e[1], e[0], 0}};
            /* P = Vcj^T S Vci - [[0],[e_l Vck]] */
            double T[4][4];
            for (int a = 0; a < rk; ++a)
                for (int b = 0; b < rk; ++b) {
                    double s = 0;
                    for (int p = 0; p < 3; ++p)
                        for (int q = 0; q < 3; ++q) s += V[3 * j + p][a] * S[p][q] * V[3 * i + q][b];
                    T[a][b] = s;
                }
            for (int b = 0; b < rk; ++b) T[rk - 1][b] -= V[3 * kk + l][b];
            for (int a = 0; a < rk; ++a)
                for (int b = 0; b < rk; ++b) P[12 + m][a][b] = 0.5 * (T[a][b] + T[b][a]);
            ++m;
        }
    double alpha[4][4];
    int nsol;
    if (rk == 2) {
        double c0 = 0, c1 = 0, c2 = 0; /* mean of [P00, 2 P01, P11] (:304-307) */
        for (int q = 0; q < 21; ++q) { c0 += P[q][0][0]; c1 += 2 * P[q][0][1]; c2 += P[q][1][1]; }
        c0 /= 21; c1 /= 21; c2 /= 21;
        double disc = c1 * c1 - 4 * c0 * c2;
        double root = sqrt(disc > 0 ? disc : 0);
        alpha[0][0] = (-c1 + root) / (2 * c0); alpha[0][1] = 1;
        alpha[1][0] = (-c1 - root) / (2 * c0); alpha[1][1] = 1;
        nsol = 2;
    } else {
        double A[21 * 10];
        for (int q = 0; q < 21; ++q) {
            double *r = A + q * 10;
            r[0] = P[q][0][0]; r[1] = P[q][1][1]; r[2] = P[q][2][2];
            r[3] = 2 * P[q][0][1]; r[4] = 2 * P[q][0][2]; r[5] = 2 * P[q][1][2];
            r[6] = 2 * P[q][0][3]; r[7] = 2 * P[q][1][3]; r[8] = 2 * P[q][2][3]; r[9] = P[q][3][3];
        }
        double a[4], b[4], c[4];
        nsol = orc_re6q3(21, A, a, b, c);
        if (nsol < 0) return -1;
        for (int s = 0; s < nsol; ++s) { alpha[s][0] = a[s]; alpha[s][1] = b[s]; alpha[s][2] = c[s]; alpha[s][3] = 1; }
    }
    for (int s = 0; s < nsol; ++s)
        for (int i = 0; i < 9; ++i) {
            double acc = 0;
            for (int a = 0; a < rk; ++a) acc += alpha[s][a] * V[i][a];
            rc_out[s * 9 + i] = acc;
        }
    return nsol;
}

/* ------------------------------------------------------------------ driver */

/*
 * cvxpnpl.py:492-520 given the solver's x and dobj.  Output poses as (R 3x3 row-major,
 * t 3), up to 4.  status bit 0: NaN sentinel (:493-498); bit 1: rank > 1 branch (:507);
 * bit 2: not certifiable (:517-519).  Returns number of poses.
 */
int orc_recover(const double *x, double dobj, const double *A, int m, const double *B, double eps,
                double *R_out, double *t_out, int *status, int *rank_out, double *eig_out)
{
    double Z[100], W[100], w[10], V[100];
    *status = 0;
    orc_vech10_inv(x, Z);
    for (int i = 0; i < 100; ++i)
        if (Z[i] != Z[i]) {
            for (int k = 0; k < 9; ++k) R_out[k] = NAN;
            for (int k = 0; k < 3; ++k) t_out[k] = NAN;
            *status = 1;
            if (rank_out) *rank_out = 0;
            return 1;
        }
    memcpy(W, Z, sizeof(W));
    orc_eigh(10, W, w, V);
    if (eig_out) memcpy(eig_out, w, sizeof(w));
    int rank = 0;
    for (int i = 0; i < 10; ++i) rank += (w[i] > 1e-3);
    if (rank_out) *rank_out = rank;
    double rc[36];
    int np;
    if (rank == 1) {
        for (int i = 0; i < 9; ++i) rc[i] = V[i * 10 + 9] / V[9 * 10 + 9];
        np = 1;
    } else {
        *status |= 2;
        np = orc_constraint_ortho_det(V, rank, rc);
        if (np < 0) { /* the reference would raise here (rank 0) */
            for (int k = 0; k < 9; ++k) R_out[k] = NAN;
            for (int k = 0; k < 3; ++k) t_out[k] = NAN;
            *status |= 1;
            return 1;
        }
    }
    for (int s = 0; s < np; ++s) {
        double Rp[9], r[9];
        orc_svd3_uvh(rc + 9 * s, Rp);       /* R' = U Vh of reshape(r_c, 3, 3)        :510-511 */
        memcpy(r, Rp, sizeof(r));           /* r = R'.reshape(9)                       :512     */
        for (int i = 0; i < 3; ++i) {
            double acc = 0;
            for (int j = 0; j < 9; ++j) acc += B[i * 9 + j] * r[j];
            t_out[3 * s + i] = -acc;        /* t = -r B^T                              :513     */
        }
        double cost = 0;
        for (int q = 0; q < m; ++q) {
            double acc = 0;
            for (int j = 0; j < 9; ++j) acc += A[q * 9 + j] * r[j];
            cost += acc * acc;
        }
        if (fabs(cost - dobj) > eps) *status |= 4; /*                                   :516-519 */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R_out[9 * s + i * 3 + j] = Rp[j * 3 + i]; /* transpose :520 */
    }
    return np;
}

/* cvxpnpl.py:454-520 */
int orc_solve_relaxation(int m, const double *A, const double *B, double eps, int max_iters,
                         double *R_out, double *t_out, orc_info_t *info)
{
    double Q[100] = {0}, c[NV], x[NV], dobj, pobj, res[3];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double s = 0;
            for (int q = 0; q < m; ++q) s += A[q * 9 + i] * A[q * 9 + j];
            Q[i * 10 + j] = s;
        }
    orc_vech10(Q, 2.0, c);
    /* balance the primal and dual blocks (the role of SCS's `scale` setting): c has the
     * magnitude of tr Q while b = e0 and tr(Z*) = 4; solving with c * 10 / tr(Q) leaves x
     * unchanged (dobj is rescaled back) and needs ~1e3 instead of ~1e4 iterations. */
    double tr = 0;
    for (int i = 0; i < 9; ++i) tr += Q[i * 10 + i];
    double cscale = tr > 0 ? 10.0 / tr : 1.0;
    int iters, st = orc_scs_solve(c, eps, max_iters, cscale, x, NULL, &dobj, &pobj, &iters, res);
    int status, rank;
    double eigs[10];
    int np = orc_recover(x, dobj, A, m, B, eps, R_out, t_out, &status, &rank, eigs);
    if (info) {
        info->n_poses = np; info->status = status; info->rank = rank; info->iters = iters;
        info->scs_status = st; info->dobj = dobj; info->pobj = pobj;
        memcpy(info->x, x, sizeof(x)); memcpy(info->eigs, eigs, sizeof(eigs));
        info->res[0] = res[0]; info->res[1] = res[1]; info->res[2] = res[2];
    }
    return np;
}

/* benchmarks/toolkit/methods/rc.py:67-131 (_solve_relaxation_rc): the same driver on the 16-equality constraint set;
 * the recovery (:104-131) is that of cvxpnpl.py:492-513 without the certificate check. */
int orc_solve_relaxation_rc(int m, const double *A, const double *B, double eps, int max_iters,
                            double *R_out, double *t_out, orc_info_t *info)
{
    double Q[100] = {0}, c[NV], x[NV], dobj, pobj, res[3];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double s = 0;
            for (int q = 0; q < m; ++q) s += A[q * 9 + i] * A[q * 9 + j];
            Q[i * 10 + j] = s;
        }
    orc_vech10(Q, 2.0, c);
    double tr = 0;
    for (int i = 0; i < 9; ++i) tr += Q[i * 10 + i];
    double cscale = tr > 0 ? 10.0 / tr : 1.0;
    int iters, st = orc_scs_solve_var(1, c, eps, max_iters, cscale, x, NULL, &dobj, &pobj, &iters, res);
    int status, rank;
    double eigs[10];
    int np = orc_recover(x, dobj, A, m, B, eps, R_out, t_out, &status, &rank, eigs);
    status &= ~4; /* rc.py has no certificate check */
    if (info) {
        info->n_poses = np; info->status = status; info->rank = rank; info->iters = iters;
        info->scs_status = st; info->dobj = dobj; info->pobj = pobj;
        memcpy(info->x, x, sizeof(x)); memcpy(info->eigs, eigs, sizeof(eigs));
        info->res[0] = res[0]; info->res[1] = res[1]; info->res[2] = res[2];
    }
    return np;
}

/* cvxpnpl.py:586-627 (pnp and pnl are the n_l = 0 / n_p = 0 cases, :523-583) */
int orc_pnpl(int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d, const double *line_3d,
             const double *K, double eps, int max_iters, double *R_out, double *t_out, orc_info_t *info)
{
    int m = 3 * n_p + 2 * n_l;
    double *C = (double *)calloc((size_t)(m > 0 ? m : 1) * 9, sizeof(double));
    double *N = (double *)calloc((size_t)(m > 0 ? m : 1) * 3, sizeof(double));
    double *A = (double *)calloc((size_t)(m > 0 ? m : 1) * 9, sizeof(double));
    double B[27];
    int rc = 0;
    if (n_p > 0) rc |= orc_point_constraints(n_p, pts_2d, pts_3d, K, C, N);
    if (n_l > 0) rc |= orc_line_constraints(n_l, line_2d, line_3d, K, C + (size_t)3 * n_p * 9, N + (size_t)3 * n_p * 3);
    rc |= orc_eliminate(m, C, N, B, A);
    int np;
    if (rc) { /* the reference raises LinAlgError; report as a NaN pose with scs_status -9 */
        for (int k = 0; k < 9; ++k) R_out[k] = NAN;
        for (int k = 0; k < 3; ++k) t_out[k] = NAN;
        if (info) { memset(info, 0, sizeof(*info)); info->n_poses = 1; info->status = 1; info->scs_status = -9; }
        np = 1;
    } else {
        np = orc_solve_relaxation(m, A, B, eps, max_iters, R_out, t_out, info);
        if (info) memcpy(info->B, B, sizeof(B));
    }
    free(C); free(N); free(A);
    return np;
}

/* batch driver for the cpu_baseline leg and the parity tests: same-shape problems,
 * contiguous [batch][n][.] inputs, shared K; R_out [batch][4][9], t_out [batch][4][3]. */
void orc_pnpl_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d,
                    const double *line_3d, const double *K, int K_per_problem, double eps, int max_iters,
                    double *R_out, double *t_out, int *n_poses, int *status, int *iters, double *cost)
{
    scs_static_init(); /* before the parallel region */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int b = 0; b < batch; ++b) {
        orc_info_t info;
        const double *Kb = K + (K_per_problem ? (size_t)9 * b : 0);
        int np = orc_pnpl(n_p, pts_2d ? pts_2d + (size_t)b * n_p * 2 : NULL, pts_3d ? pts_3d + (size_t)b * n_p * 3 : NULL,
                          n_l, line_2d ? line_2d + (size_t)b * n_l * 4 : NULL, line_3d ? line_3d + (size_t)b * n_l * 6 : NULL,
                          Kb, eps, max_iters, R_out + (size_t)b * 36, t_out + (size_t)b * 12, &info);
        if (n_poses) n_poses[b] = np;
        if (status) status[b] = info.status;
        if (iters) iters[b] = info.iters;
        if (cost) { cost[2 * b] = info.pobj; cost[2 * b + 1] = info.dobj; }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
