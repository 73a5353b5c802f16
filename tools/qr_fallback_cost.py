"""Cost of the CholeskyQR fallback ladder: time of mpsk_qrpos on inputs that take
  (a) the fast path (CholeskyQR3, rounding-level shift),
  (b) the shift retry (published shift of Fukaya et al.),
  (c) the robust variant (perturbed, repeatedly shifted passes),
at the shapes of the benchmark (2048 x 1024) and of the config-4 split (4096 x 1024).  usage: qr_fallback_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)


def timed(A, reps=5):
    be.qrpos(A)
    torch.cuda.synchronize()
    s0, r0 = be.qr_stats(), be.qr_retries()
    t0 = time.perf_counter()
    for _ in range(reps):
        be.qrpos(A)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    s1, r1 = be.qr_stats(), be.qr_retries()
    return dt, {k: (s1[k] - s0[k]) // reps for k in s1}, (r1 - r0) // reps


for (m, n) in ((2048, 1024), (4096, 1024)):
    g = torch.Generator(device="cuda").manual_seed(7)
    U, _ = torch.linalg.qr(torch.randn(m, n, dtype=torch.float64, device="cuda", generator=g))
    V, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    for name, s in (("well conditioned (cond 1e3)", torch.logspace(0, -3, n, dtype=torch.float64, device="cuda")),
                    ("cond 1e12 (shift retry)", torch.logspace(0, -12, n, dtype=torch.float64, device="cuda")),
                    ("rank n/2 (robust variant)", torch.cat([torch.logspace(0, -3, n // 2, dtype=torch.float64, device="cuda"),
                                                            torch.zeros(n - n // 2, dtype=torch.float64, device="cuda")]))):
        M = (U * s) @ V.T
        A = mk.DTensor(M.T.contiguous().flatten(), (m, n))
        dt, st, rt = timed(A)
        print(f"qrpos {m}x{n} {name:28s}: {dt:7.3f} ms  per call: {st}, shift retries {rt}", flush=True)
