"""One timed two-site split after a warm-up call, for kernel traces: python tools/svd_once.py [n] [kind] [reps] [svd mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "graded6"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g = torch.Generator(device="cuda").manual_seed(1)
if kind.startswith("graded"):
    Q1, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    Q2, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    M = (Q1 * torch.logspace(0, -float(kind[6:]), n, dtype=torch.float64, device="cuda")) @ Q2.T
else:
    M = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g) - 0.5
A = mk.DTensor(M.T.contiguous().flatten(), (n, n))
be.set_svd_mode(mode)
be.tsplit(A, max_keep=n // 4)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    be.tsplit(A, max_keep=n // 4)
torch.cuda.synchronize()
print(f"tsplit {n} {kind} mode {mode}: {(time.time() - t0) / reps * 1e3:.1f} ms, sweeps {be.svd_sweeps()}, {be.split_stats()}", flush=True)
