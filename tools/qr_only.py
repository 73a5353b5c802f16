"""QRpos timing (shifted CholeskyQR3): python tools/qr_only.py [m n [reps]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
A = mk.DTensor(torch.rand(m * n, dtype=torch.float64, device=be.device) - 0.5, (m, n))
B = mk.DTensor(torch.rand(m * n, dtype=torch.float64, device=be.device) - 0.5, (m, n))
for _ in range(3):
    be.qrpos(A)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.qrpos(A)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / reps * 1e3
for _ in range(2):
    be.qrpos2(A, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.qrpos2(A, B)
torch.cuda.synchronize()
t2 = (time.perf_counter() - t0) / reps * 1e3
print(f"qrpos {m}x{n}: {t1:.3f} ms ; qrpos2 (two factorizations on two streams): {t2:.3f} ms ; stats {be.qr_stats()}", flush=True)
