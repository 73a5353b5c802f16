"""Timing of the prepared-operator matvec (mpsk_hac_apply) and of its stage kernels for the library named by MPSK_LIB
(default: the in-tree libmpsk.so) -- the A/B harness for kernel experiments.  usage: ab_dac.py D,d [D,d ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk

def flops_dAC(D, d, W):
    return 2 * W * D * D * d * D + 2 * W * W * d * d * D * D + 2 * W * D * d * D * D

be = mk.Backend(0)
if os.environ.get('AB_TILE'):
    be.lib.mpsk_ctx_force_tile(be.ctx, *[int(v) for v in os.environ['AB_TILE'].split(',')])
W = 5
for a in (sys.argv[1:] or ["1024,2"]):
    D, d = (int(v) for v in a.split(","))
    H = mk.heisenberg_XXX(0.5 if d == 2 else 1.0, be=be)
    r = lambda *s: mk.DTensor(torch.rand(*s, dtype=torch.float64, device=be.device).flatten(), s)
    GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
    h = mk.MPO_ddAC(be, H[int(os.environ.get("AB_SITE", "1"))], GL, GR)
    print("   hac", h._hac.info() if getattr(h, "_hac", None) else None)
    for _ in range(5):
        h(x, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            h(x, out=y)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    print(f"{os.environ.get('MPSK_LIB', 'libmpsk.so')}: D={D} d={d}: {best:.4f} ms  {flops_dAC(D, d, W) / best * 1e-9:.2f} TFLOP/s (algorithmic)", flush=True)
    be.prof_enable(True)
    for _ in range(10):
        h(x, out=y)
    for r_ in be.prof_summary():
        print(f"    {r_['kernel']}: {r_['avg_ms']*1e3:.1f} us  {r_['flops']/r_['launches']/r_['avg_ms']*1e-9:.1f} TF/s")
    be.prof_enable(False)
