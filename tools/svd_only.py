import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
A = mk.DTensor(torch.rand(n * n, dtype=torch.float64, device=be.device), (n, n))
be.tsvd(A, max_keep=n // 4)
torch.cuda.synchronize()
