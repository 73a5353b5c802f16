"""N fixed-budget eigsolves (8 matvecs) at the north-star site size: workload for kernel-trace timelines of ONE site's solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import algorithms as alg, krylov
be = mk.Backend(0)
D, d, W = 1024, 2, 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H = mk.heisenberg_XXX(0.5, be=be)
r = lambda *s: mk.DTensor(torch.rand(int(np.prod(s)), dtype=torch.float64, device=be.device), s)
g = torch.rand(W, D, D, dtype=torch.float64, device=be.device); g = g + g.transpose(1, 2)
GL = mk.DTensor(g.flatten().contiguous(), (W, D, D)); GR = mk.DTensor(g.flip(0).flatten().contiguous(), (W, D, D))
x = r(D, d, D)
h = mk.MPO_ddAC(be, H[1], GL, GR)
eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8); ws = krylov.KrylovWorkspace(be)
for _ in range(3):
    alg.fixedpoint(be, h, x, eig, ws, values=os.environ.get("VALUES", "1") == "1")
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n):
    alg.fixedpoint(be, h, x, eig, ws, values=os.environ.get("VALUES", "1") == "1")
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per eigsolve")
