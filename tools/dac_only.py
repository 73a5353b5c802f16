"""dAC matvecs through the prepared operator at a chosen point -- the workload of the rocprofv3 --pmc passes and of
kernel-trace runs.  usage: dac_only.py [D d reps]   (default: the north-star point D=1024 d=2 W=5, 10 applications)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = 5
H = mk.heisenberg_XXX(0.5 if d == 2 else 1.0, be=be)
r = lambda *s: mk.DTensor(torch.rand(*s, dtype=torch.float64, device=be.device).flatten(), s)
GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
h = mk.MPO_ddAC(be, H[0], GL, GR)
for _ in range(reps):
    h(x, out=y)
torch.cuda.synchronize()
