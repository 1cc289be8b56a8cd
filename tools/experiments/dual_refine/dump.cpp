#include <vector>
#include <mutex>
#include "csrc/solver_core.h"
#include "csrc/problem_io.h"
static thread_local int cur_b = -1;
struct Rec { int b, it, ok; double delta; double S[55]; double R[9]; double Q[45]; double W[55]; double Wp[55]; };
static std::vector<Rec> recs;
static std::mutex mu;
static long g_tried, g_ok;
namespace cvx {
void cert_dump(const double *S, const double *R, double delta, int it, bool ok, const double *Qs45, const double *W, const double *Wp)
{
    Rec r; r.b = cur_b; r.it = it; r.ok = ok; r.delta = delta;
    for (int i = 0; i < 55; ++i) r.S[i] = S[i];
    for (int i = 0; i < 9; ++i) r.R[i] = R[i];
    for (int i = 0; i < 45; ++i) r.Q[i] = Qs45[i];
    for (int i = 0; i < 55; ++i) { r.W[i] = W[i]; r.Wp[i] = Wp[i]; }
    std::lock_guard<std::mutex> g(mu);
    recs.push_back(r);
}
}
extern "C" {
void dr_policy(int mode, double sigma, int nit, double k, int first_too, int dshift_first, int steps) { cvx::g_refine = {mode, sigma, nit, k, first_too, dshift_first, steps}; }

void dr_default_opts(cvx::Opts *o) { *o = cvx::default_opts(); }
int dr_solve_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *l2, const double *l3, const double *K, const cvx::Opts *opts, int *status, int *iters, double *R_out)
{
    recs.clear();
    long tried = 0, okc = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tried, okc)
    for (int b = 0; b < batch; ++b) {
        cur_b = b; cvx::g_ref_tried = 0; cvx::g_ref_ok = 0;
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, n_l, l2, l3, K, 0);
        cvx::Solution sol;
        cvx::solve_problem(pv, *opts, sol, nullptr);
        status[b] = sol.status; iters[b] = sol.iters;
        for (int i = 0; i < 9; ++i) R_out[9 * (size_t)b + i] = sol.R[i];
        tried += cvx::g_ref_tried; okc += cvx::g_ref_ok;
    }
    g_tried = tried; g_ok = okc;
    return (int)recs.size();
}
long dr_tried() { return g_tried; }
long dr_ok() { return g_ok; }
// out: n x 223
void dr_get(double *out)
{
    for (size_t k = 0; k < recs.size(); ++k) {
        double *o = out + k * 223;
        o[0] = recs[k].b; o[1] = recs[k].it; o[2] = recs[k].ok; o[3] = recs[k].delta;
        for (int i = 0; i < 55; ++i) o[4 + i] = recs[k].S[i];
        for (int i = 0; i < 9; ++i) o[59 + i] = recs[k].R[i];
        for (int i = 0; i < 45; ++i) o[68 + i] = recs[k].Q[i];
        for (int i = 0; i < 55; ++i) { o[113 + i] = recs[k].W[i]; o[168 + i] = recs[k].Wp[i]; }
    }
}
}
