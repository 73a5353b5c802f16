"""complex128 matvec A/B: the native MPSK_C128 operator (interleaved complex, 4x real flops) against the same operator
on the real 2x2 bond embedding (8x real flops), plus a real-time TDVP step of a complex state with each.
usage: bench_cplx.py D [D ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import cplx

be = mk.Backend(0)
d, W = 2, 5


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for D in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    rng = np.random.default_rng(D)
    H = mk.heisenberg_XXX(0.5, be=be)[0]
    cr = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    GL, GR, x = [cr(D, 1, D) for _ in range(W)], [cr(D, 1, D) for _ in range(W)], cr(D, d, D)
    # embedded operands (what a complex FiniteMPS stores): real tensors with doubled bond dimensions
    GLe = be.upload_env([cplx.embed(np.transpose(g, (0, 2, 1)))[:, :, None, :].transpose(0, 2, 1, 3).reshape(2 * D, 1, 2 * D) for g in []] or
                        [np.stack([cplx.embed(g[:, 0, :])], axis=1) for g in GL])
    GRe = be.upload_env([np.stack([cplx.embed(g[:, 0, :])], axis=1) for g in GR])
    xe = be.upload(cplx.embed(x))
    ye = be.empty(2 * D, d, 2 * D)
    hemb = mk.MPO_ddAC(be, H, GLe, GRe)
    t_emb = timeit(lambda: hemb(xe, out=ye))
    hnat = cplx.HalfEmbeddedOp(be, "AC", [H], GLe, GRe)
    xh = hnat.encode(xe)
    yh = be.empty(*xh.shape)
    t_nat = timeit(lambda: hnat.apply_half(xh, yh))
    err = np.abs(cplx.extract(be.download(hnat.decode(yh))) - cplx.extract(be.download(ye))).max() / np.abs(be.download(ye)).max()
    cflops = 4 * (2 * W * D * D * d * D + 2 * W * W * d * d * D * D + 2 * W * D * d * D * D)
    print(f"complex dAC D={D} d={d} W={W}: embedded {t_emb:.3f} ms ({2 * cflops / t_emb / 1e9:.1f} real TF/s), native {t_nat:.3f} ms "
          f"({cflops / t_nat / 1e9:.1f} TF/s of 4x-real flops) -> {t_emb / t_nat:.2f}x, max rel diff {err:.1e}", flush=True)
