"""Two-site split timing / sweep counts per svd mode (1: QR, 2: QR + QR of R^T, 3: 2 + truncation-aware stage) and spectrum.
usage: svd_probe.py n[,n...] [kind ...]     kinds: graded6 graded12 uniform colgraded"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2048").split(",")]
kinds = sys.argv[2:] or ["graded6", "graded12", "uniform"]


def make(n, kind):
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind.startswith("graded"):
        dec = float(kind[6:])
        Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
        Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
        return (Q1 * torch.logspace(0, -dec, n, dtype=torch.float64, device="cuda")) @ Q2.T
    M = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    if kind == "colgraded":
        M = M * torch.logspace(0, -6, n, dtype=torch.float64, device="cuda")[None, :]
    return M


for n in sizes:
    for kind in kinds:
        M = make(n, kind)
        A = mk.DTensor(M.T.contiguous().flatten(), (n, n))
        sref = torch.linalg.svdvals(M).cpu().numpy()
        for mode in (1, 2, 3):
            be.set_svd_mode(mode)
            be.tsplit(A, max_keep=n // 4)
            torch.cuda.synchronize()
            t0 = time.time()
            al, c, ar, S, disc = be.tsplit(A, max_keep=n // 4)
            torch.cuda.synchronize()
            dt = (time.time() - t0) * 1e3
            k = n // 4
            alT = torch.as_strided(al.buf, (n, k), (1, n)); arT = torch.as_strided(ar.buf, (k, n), (1, k)); cT = torch.as_strided(c.buf, (k, k), (1, k))
            eye = torch.eye(k, dtype=torch.float64, device="cuda")
            orth = max(float((alT.T @ alT - eye).abs().max()), float((arT @ arT.T - eye).abs().max()))
            rec = float(((alT @ cT @ arT) - M).square().sum().sqrt())
            print(f"tsplit {n}x{n} {kind:9s} mode {mode}: {dt:7.1f} ms  sweeps {be.svd_sweeps():2d}  |S - Sref| {np.abs(S - sref[:k]).max():.1e}  "
                  f"orth {orth:.1e}  |rec - theta| - disc {abs(rec - disc):.1e}", flush=True)
be.set_svd_mode(3)
