"""Time one mpsk_gemm shape (library's own tile / split choice, or MPSK_SPLITK_F forced): usage gemm_shape.py M N K tA tB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
M, N, K, tA, tB = (int(v) for v in sys.argv[1:6])
r = lambda *s: mk.DTensor(torch.rand(int(np.prod(s)), dtype=torch.float64, device=be.device), s)
A = r(K, M) if tA else r(M, K)
B = r(N, K) if tB else r(K, N)
C = be.empty(M, N)
for _ in range(5): be.gemm(A, B, transA=bool(tA), transB=bool(tB), out=C)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): be.gemm(A, B, transA=bool(tA), transB=bool(tB), out=C)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
print(f"gemm {M}x{N}x{K} tA={tA} tB={tB} F={os.environ.get('MPSK_SPLITK_F','auto')}: {best*1e3:.1f} us  {2*M*N*K/best*1e-9:.1f} TF/s", flush=True)
