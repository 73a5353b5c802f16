"""10 dAC matvecs at the north-star point (D=1024, d=2, W=5) -- the workload for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D, d, W = 1024, 2, 5
H = mk.heisenberg_XXX(0.5, be=be)
r = lambda *s: mk.DTensor(torch.rand(*s, dtype=torch.float64, device=be.device).flatten(), s)
GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
for _ in range(10):
    be.dAC(H[0], GL, GR, x, out=y)
torch.cuda.synchronize()
