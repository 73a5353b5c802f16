"""tsvd / tsplit timing: python tools/svd_only.py n [uniform|graded] [pre|plain|split]  -> time, Jacobi sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
mode = sys.argv[3] if len(sys.argv) > 3 else "pre"
be.set_svd_mode(mode != "plain")
if kind == "graded":     # Schmidt-like spectrum over 12 decades behind random orthogonal factors
    g = torch.Generator(device="cuda").manual_seed(1)
    Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    s = torch.logspace(0, -12, n, dtype=torch.float64, device="cuda")
    M = (Q1 * s) @ Q2.T
    A = mk.DTensor(M.T.contiguous().flatten(), (n, n))
else:
    A = mk.DTensor(torch.rand(n * n, dtype=torch.float64, device=be.device), (n, n))
fn = be.tsplit if mode == "split" else be.tsvd
fn(A, max_keep=n // 4)
torch.cuda.synchronize()
t0 = time.time()
fn(A, max_keep=n // 4)
torch.cuda.synchronize()
print(f"tsvd {n}x{n} {kind} {mode}: {(time.time() - t0) * 1e3:.1f} ms, sweeps = {be.svd_sweeps()}, qr = {be.qr_stats()}", flush=True)
