"""Experiments behind the two-site split: sweep counts of the block-Jacobi iteration with one and
with two QR preconditioning passes, and the time split of mpsk_tsplit.  usage: svd_probe.py n [graded6|graded12|uniform]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kind = sys.argv[2] if len(sys.argv) > 2 else "graded6"
g = torch.Generator(device="cuda").manual_seed(1)
if kind.startswith("graded"):
    dec = float(kind[6:])
    Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    M = (Q1 * torch.logspace(0, -dec, n, dtype=torch.float64, device="cuda")) @ Q2.T
else:
    M = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g)
A = mk.DTensor(M.T.contiguous().flatten(), (n, n))


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3, out


t, _ = timed(lambda: be.qrpos(A))
print(f"qrpos {n}x{n}: {t:.2f} ms", flush=True)
t, _ = timed(lambda: be.tsplit(A, max_keep=n // 4))
print(f"tsplit {n}x{n} {kind}: {t:.1f} ms, sweeps = {be.svd_sweeps()}", flush=True)
# Jacobi on R^T after ONE QR (what tsplit does) vs after TWO (R^T = Q1 R1, Jacobi on R1^T): plain mode on the given matrix
_, R = be.qrpos(A)
Rt = mk.DTensor(torch.as_strided(R.buf, (n, n), (1, n)).contiguous().flatten(), (n, n))      # column-major R^T... see below
be.set_svd_mode(False)
try:
    # plain mode runs Jacobi on the COLUMNS of its argument: pass R^T (columns of R^T = rows of R)
    RT = mk.DTensor(torch.as_strided(R.buf, (n, n), (n, 1)).contiguous().flatten(), (n, n))
    t1, _ = timed(lambda: be.tsvd(RT, max_keep=n // 4), reps=1)
    s1 = be.svd_sweeps()
    _, R1 = be.qrpos(RT)
    R1T = mk.DTensor(torch.as_strided(R1.buf, (n, n), (n, 1)).contiguous().flatten(), (n, n))
    t2, _ = timed(lambda: be.tsvd(R1T, max_keep=n // 4), reps=1)
    s2 = be.svd_sweeps()
    _, R2 = be.qrpos(R1T)
    R2T = mk.DTensor(torch.as_strided(R2.buf, (n, n), (n, 1)).contiguous().flatten(), (n, n))
    t3, _ = timed(lambda: be.tsvd(R2T, max_keep=n // 4), reps=1)
    s3 = be.svd_sweeps()
    print(f"plain Jacobi (with V) on R^T: {s1} sweeps {t1:.0f} ms | on R1^T (2 QRs): {s2} sweeps {t2:.0f} ms | on R2^T (3 QRs): {s3} sweeps {t3:.0f} ms", flush=True)
finally:
    be.set_svd_mode(True)
