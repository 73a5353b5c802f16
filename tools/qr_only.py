import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D = 1024
A = mk.DTensor(torch.rand(2 * D * D, dtype=torch.float64, device=be.device) - 0.5, (2 * D, D))
for _ in range(6):
    be.qrpos(A)
torch.cuda.synchronize()
