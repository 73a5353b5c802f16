"""tsplit 4096^2 under forced GEMM tile shapes (experiment: which tile suits the tall-skinny Jacobi update / Gram GEMMs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
M = (Q1 * torch.logspace(0, -6, n, dtype=torch.float64, device="cuda")) @ Q2.T
A = mk.DTensor(M.T.contiguous().flatten(), (n, n))
for tile in ((0, 0), (64, 64), (128, 64), (64, 128), (128, 128)):
    be.lib.mpsk_ctx_force_tile(be.ctx, *tile)
    be.tsplit(A, max_keep=n // 4); torch.cuda.synchronize()
    t0 = time.time(); be.tsplit(A, max_keep=n // 4); torch.cuda.synchronize()
    print(f"tsplit {n} tile {tile}: {(time.time()-t0)*1e3:.1f} ms sweeps {be.svd_sweeps()}", flush=True)
be.lib.mpsk_ctx_force_tile(be.ctx, 0, 0)
