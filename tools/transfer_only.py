"""transfer_left / transfer_right at the north-star site size, n repetitions each: workload for kernel-trace timelines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D, d, W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 2, 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "both"
H = mk.heisenberg_XXX(0.5, be=be)
r = lambda *s: mk.DTensor(torch.rand(int(np.prod(s)), dtype=torch.float64, device=be.device), s)
GL, GR, A = r(W, D, D), r(W, D, D), r(D, d, D)
def timeit(tag, f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / n:.4f} ms", flush=True)
if which in ("both", "left"):
    timeit("transfer_left", lambda: be.transfer_left(H[1], GL, A, A))
if which in ("both", "right"):
    timeit("transfer_right", lambda: be.transfer_right(H[1], GR, A, A))
