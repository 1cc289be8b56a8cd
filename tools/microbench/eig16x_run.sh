# round 5: orderings of the quad layout's Jacobi step (tools/microbench/eig16x.hip) -> gpurun_out/r05/eig16x.jsonl
cd $GRAFT_REPO_ROOT/tools/microbench
mkdir -p ../../gpurun_out/r05
O=../../gpurun_out/r05/eig16x.jsonl
: > $O
# fixed 8 sweeps (tol 0): cost per sweep at 2 and 1 wavefronts per SIMD; 10 k matrices = the judged launch's 2 500 wavefronts
./eig16x 10000 8 0 2 >> $O
./eig16x 10000 8 0 1 >> $O
./eig16x 4096 8 0 1 >> $O
# convergence: sweeps to the sweep tolerance of the solver (6e-2) and to 1e-6
./eig16x 10000 30 6e-2 2 >> $O
./eig16x 10000 30 1e-6 2 >> $O
cat $O
