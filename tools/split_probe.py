"""Truncation-aware two-site split (svd mode 3) against the full iteration (mode 2): python tools/split_probe.py [reps]
time, sweeps, subspace iterations, check value, |S3 - S2|, isometry defects, |theta - al c ar| - disc, subspace distance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator(device="cuda").manual_seed(1)
def graded(m, n, dec):
    k = min(m, n)
    Q1, _ = torch.linalg.qr(torch.randn(m, k, dtype=torch.float64, device="cuda", generator=g))
    Q2, _ = torch.linalg.qr(torch.randn(n, k, dtype=torch.float64, device="cuda", generator=g))
    return (Q1 * torch.logspace(0, -dec, k, dtype=torch.float64, device="cuda")) @ Q2.T
def run(M, keep, mode):
    m, n = M.shape
    A = mk.DTensor(M.T.contiguous().flatten(), (m, n))
    be.set_svd_mode(mode)
    out = be.tsplit(A, max_keep=keep)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps):
        out = be.tsplit(A, max_keep=keep)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    al, c, ar, s, disc = out
    tt = lambda x: x.buf[:x.shape[0] * x.shape[1]].view(x.shape[1], x.shape[0]).T
    return dt, tt(al).clone(), tt(c).clone(), tt(ar).clone(), s, disc, be.svd_sweeps(), be.split_stats()
cases = [("graded6 4096^2 -> 1024", graded(4096, 4096, 6), 1024), ("graded3 4096^2 -> 1024", graded(4096, 4096, 3), 1024),
         ("graded6 2048^2 -> 512", graded(2048, 2048, 6), 512), ("graded6 2048x4096 -> 512", graded(2048, 4096, 6), 512),
         ("graded6 4096x2048 -> 512", graded(4096, 2048, 6), 512), ("graded10 1024^2 -> 256", graded(1024, 1024, 10), 256),
         ("uniform 2048^2 -> 512", torch.rand(2048, 2048, dtype=torch.float64, device="cuda", generator=g) - 0.5, 512)]
for name, M, keep in cases:
    r2 = run(M, keep, 2); r3 = run(M, keep, 3)
    k = len(r3[4])
    eye = torch.eye(k, dtype=torch.float64, device="cuda")
    rec = (r3[1] @ r3[2] @ r3[3] - M).norm().item()
    print(f"{name:26s} mode2 {r2[0]*1e3:7.1f} ms ({r2[6]} sweeps)  mode3 {r3[0]*1e3:7.1f} ms ({r3[6]} sweeps, {r3[7]})  "
          f"|S3-S2|/S0 {np.abs(r3[4] - r2[4]).max() / r2[4][0]:.1e}  orth {(r3[1].T @ r3[1] - eye).abs().max().item():.1e} {(r3[3] @ r3[3].T - eye).abs().max().item():.1e}  "
          f"|rec|-disc {rec - r3[5]:.1e} (disc {r3[5]:.3e} vs {r2[5]:.3e})  k - |al3^T al2|_F^2 {k - (r3[1].T @ r2[1]).norm().item() ** 2:.1e}", flush=True)
