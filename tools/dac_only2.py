"""dAC matvecs at a chosen (D, d, W) for PMC passes: python tools/dac_only2.py D d reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D, d, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W = 5
H = mk.heisenberg_XXX(0.5, be=be) if d == 2 else mk.heisenberg_XXX(1.0, be=be)
r = lambda *s: mk.DTensor(torch.rand(*s, dtype=torch.float64, device=be.device).flatten(), s)
GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
for _ in range(reps):
    be.dAC(H[0], GL, GR, x, out=y)
torch.cuda.synchronize()
