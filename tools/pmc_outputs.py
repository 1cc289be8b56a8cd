import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth, _lib
from cvxpnpl_amd.api import _ptr
variant, batch, li = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
d = synth.make_pnp(batch, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k], device=dev).contiguous() for k in ("pts_2d", "pts_3d", "K"))
L = _lib.lib()
o = _lib.default_opts(lane_iters=li, layout=3)
R = torch.empty((batch, 3, 3), dtype=torch.float64, device=dev); t = torch.empty((batch, 3), dtype=torch.float64, device=dev)
st = torch.empty((batch,), dtype=torch.int32, device=dev); it = torch.empty((batch,), dtype=torch.int32, device=dev)
cost = torch.empty((batch, 2), dtype=torch.float64, device=dev); work = torch.empty((batch, 2), dtype=torch.int32, device=dev)
full = variant == "all"
for _ in range(4):
    rc = L.cvxpnpl_solve_batch(batch, 10, _ptr(p2), _ptr(p3), 0, None, None, _ptr(K), 0, C.byref(o), _ptr(R), _ptr(t), _ptr(st),
                               _ptr(it) if full else None, _ptr(cost) if full else None, None, _ptr(work) if full else None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
print(variant, batch, li, np.bincount(st.cpu().numpy(), minlength=3).tolist())
