"""Kernel-level timing of the dAC matvec (north-star point D=1024, d=2, W=5) on synthetic
uniform[0,1) tensors; prints achieved TFLOP/s with the algorithmic flop model of BASELINE.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpskit_jl_amd as mk

def flops_dAC(D, d, W):
    return 2 * W * D * D * d * D + 2 * W * W * d * d * D * D + 2 * W * D * d * D * D

def main():
    be = mk.Backend(0)
    rng = np.random.default_rng(20240213)
    cfgs = [(256, 3, 5), (512, 2, 3), (1024, 2, 5), (1024, 4, 6), (2048, 2, 5)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (D, d, W) in cfgs:
        blocks = {(0, 0): 1.0, (W - 1, W - 1): 1.0}
        for i in range(1, W - 1):
            blocks[(0, i)] = rng.random((1, d, d, 1))
            blocks[(i, W - 1)] = rng.random((1, d, d, 1))
        blocks[(0, W - 1)] = rng.random((1, d, d, 1))
        H = be.mposlice(W, d, [1] * W, [1] * W, blocks)
        GL = mk.DTensor(torch.rand(W * D * D, dtype=torch.float64, device=be.device), (W, D, D))
        GR = mk.DTensor(torch.rand(W * D * D, dtype=torch.float64, device=be.device), (W, D, D))
        x = mk.DTensor(torch.rand(D * d * D, dtype=torch.float64, device=be.device), (D, d, D))
        y = be.empty(D, d, D)
        for tile in [(0, 0), (128, 128), (64, 128), (128, 64), (64, 64)]:
            be.lib.mpsk_ctx_force_tile(be.ctx, *tile)
            for _ in range(3):
                be.dAC(H, GL, GR, x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                be.dAC(H, GL, GR, x, out=y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"D={D} d={d} W={W} tile={tile}: {ms:.3f} ms  {flops_dAC(D, d, W) / ms * 1e-9:.2f} TFLOP/s", flush=True)
            if os.environ.get("MPSK_BENCH_PROF"):
                be.prof_enable(True)
                for _ in range(5):
                    be.dAC(H, GL, GR, x, out=y)
                for r in be.prof_summary():
                    print(f"    {r['kernel']}: {r['avg_ms']*1e3:.1f} us  {r['flops']/r['launches']/r['avg_ms']*1e-9:.1f} TF/s")
                be.prof_enable(False)
        be.lib.mpsk_ctx_force_tile(be.ctx, 0, 0)

main()
