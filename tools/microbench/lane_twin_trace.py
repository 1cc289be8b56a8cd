import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); 
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvxpnpl_amd import synth
d = synth.make_pnp(1024, 4, 1.0, seed=9)
lib = sys.argv[1]
L = C.CDLL(lib)
L.repro_run.argtypes = [C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_void_p]
dev = torch.device("cuda:0")
B = 1024
p2, p3, K = (torch.as_tensor(d[k], device=dev).contiguous() for k in ("pts_2d", "pts_3d", "K"))
R = torch.zeros((B, 9), dtype=torch.float64, device=dev); Z = torch.zeros((B, 55), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
which = int(os.environ.get("REPRO_WHICH", "0"))      # 0: no twin logic + single-precision sweeps, 1: twin logic, 2: no twin logic + float64 sweeps
mi = int(os.environ.get("REPRO_MAX_ITERS", "300"))
if len(sys.argv) <= 2:
    L.repro_run(which, B, 4, p2.data_ptr(), p3.data_ptr(), K.data_ptr(), mi, int(os.environ.get('REPRO_F64', '0')), R.data_ptr(), st.data_ptr(), it.data_ptr(), Z.data_ptr(), None)
    torch.cuda.synchronize()
    bad = np.flatnonzero(st.cpu().numpy() == 3)
    print("which", which, "max_iters", mi, "bad", len(bad), bad[:10], "iters at NaN", it.cpu().numpy()[bad[:10]])
if len(sys.argv) > 2:   # one problem alone, e.g. with a -DCVX_TRACE build: the iteration log of solver_core.h
    i = int(sys.argv[2])
    p2i, p3i = p2[i:i + 1].contiguous(), p3[i:i + 1].contiguous()
    L.repro_run(0, 1, 4, p2i.data_ptr(), p3i.data_ptr(), K.data_ptr(), 40, 0, R.data_ptr(), st.data_ptr(), it.data_ptr(), Z.data_ptr(), None)
    torch.cuda.synchronize()
    print("alone: status", int(st[0]), "iters", int(it[0]))
