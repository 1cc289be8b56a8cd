#!/bin/bash
# diagnostics: marginal cost of one ADMM iteration and of one Jacobi sweep in the quad layout
# (8192 problems = one resident round of wavefronts; no certification until the end; fixed sweep counts)
run() { # li sweeps
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap --layout 3 --batch 8192 --opt lane_iters=$1 --opt max_iters=$(($1+1)) --opt first_check=100000 --opt jacobi_tol=0 --opt jacobi_sweeps=$2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quad iters $1 sweeps/eig $2: %.3f ms  (sweeps/problem %.1f)'%(d['ms_per_step'], d['solver']['mean_jacobi_sweeps']))"
}
run 20 1; run 40 1; run 20 2; run 40 2; run 20 4; run 40 4
