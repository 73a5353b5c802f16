"""two-site split keeping HALF of the values (d = 2 chains: theta 2D x 2D -> D): python tools/svd_half.py n [kind] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "graded6"; mode = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device="cuda").manual_seed(1)
Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
M = (Q1 * torch.logspace(0, -float(kind[6:]), n, dtype=torch.float64, device="cuda")) @ Q2.T
A = mk.DTensor(M.T.contiguous().flatten(), (n, n))
be.set_svd_mode(mode)
for _ in range(2):
    be.tsplit(A, max_keep=n // 2)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3):
    al, c, ar, S, disc = be.tsplit(A, max_keep=n // 2)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3 * 1e3
sref = torch.linalg.svdvals(M).cpu().numpy()
import numpy as np
print(f"tsplit {n} -> {n // 2} {kind} mode {mode}: {dt:.1f} ms, sweeps {be.svd_sweeps()}, {be.split_stats()}, |S - Sref| {np.abs(S - sref[:n // 2]).max():.1e}", flush=True)
