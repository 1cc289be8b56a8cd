/* oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE, see oracle.c). */
#ifndef CVXPNPL_ORACLE_H
#define CVXPNPL_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_poses;     /* 1, 2 or 4 (cvxpnpl.py:520) */
    int status;      /* bit0 NaN sentinel, bit1 rank>1 branch, bit2 not certifiable */
    int rank;        /* #eig(Z) > 1e-3 (cvxpnpl.py:502) */
    int iters;       /* solver iterations used */
    int scs_status;  /* 0 converged to eps, 1 hit max_iters, -9 singular input */
    double dobj, pobj;
    double x[55];    /* solver x = vech(Z) */
    double eigs[10]; /* eigenvalues of Z, ascending */
    double res[3];   /* primal, dual, gap residuals at exit */
    double B[27];    /* translation map, t = -B r */
} orc_info_t;

void orc_eigh(int n, double *A, double *w, double *V);
void orc_svd3_uvh(const double *M, double *R);
int orc_point_constraints(int n, const double *pts_2d, const double *pts_3d, const double *K, double *C, double *N);
int orc_line_constraints(int n, const double *line_2d, const double *line_3d, const double *K, double *C, double *N);
int orc_eliminate(int m, const double *C, const double *N, double *B, double *A);
void orc_vech10(const double *A, double scale, double *v);
void orc_vech10_inv(const double *v, double *A);
void orc_sdp_constraints(double *Ad, double *b);
void orc_sdp_constraints_rc(double *Ad71x55, double *b71); /* benchmarks/toolkit/methods/rc.py:9-64 */
int orc_scs_solve(const double *c, double eps, int max_iters, double cscale, double *x, double *y,
                  double *dobj, double *pobj, int *iters, double *res);
int orc_scs_solve_var(int variant, const double *c, double eps, int max_iters, double cscale, double *x, double *y,
                      double *dobj, double *pobj, int *iters, double *res); /* variant 1: the rc constraint set */
int orc_solve_relaxation_rc(int m, const double *A, const double *B, double eps, int max_iters,
                            double *R_out, double *t_out, orc_info_t *info); /* rc.py:67-131 */
int orc_poly_roots_real(int deg, const double *p, double *re);
int orc_re6q3(int nrows, const double *A, double *a, double *b, double *c);
int orc_constraint_ortho_det(const double *vecs, int rank, double *rc_out);
int orc_recover(const double *x, double dobj, const double *A, int m, const double *B, double eps,
                double *R_out, double *t_out, int *status, int *rank_out, double *eig_out);
int orc_solve_relaxation(int m, const double *A, const double *B, double eps, int max_iters,
                         double *R_out, double *t_out, orc_info_t *info);
int orc_pnpl(int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d, const double *line_3d,
             const double *K, double eps, int max_iters, double *R_out, double *t_out, orc_info_t *info);
void orc_pnpl_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d,
                    const double *line_3d, const double *K, int K_per_problem, double eps, int max_iters,
                    double *R_out, double *t_out, int *n_poses, int *status, int *iters, double *cost);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
