"""Wall-clock breakdown of ONE bulk-site update of the DMRG sweep (D=1024, d=2, W=5): every
component timed with a device sync on both sides, so host launch overhead is included."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpskit_jl_amd as mk
from mpskit_jl_amd import krylov, algorithms as alg


def timeit(name, fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:34s} {dt:9.3f} ms", flush=True)
    return dt


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    d, W = 2, 5
    be = mk.Backend(0)
    H = mk.heisenberg_XXX(0.5, be=be)
    r = lambda *s: mk.DTensor(torch.rand(int(np.prod(s)), dtype=torch.float64, device=be.device) - 0.5, s)
    GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
    A = r(D * d, D)
    At = r(D, d * D)
    timeit("dAC", lambda: be.dAC(H[0], GL, GR, x, out=y))
    timeit("transfer_left", lambda: be.transfer_left(H[0], GL, x, x))
    timeit("transfer_right", lambda: be.transfer_right(H[0], GR, x, x))
    timeit("qrpos (2D x D)", lambda: be.qrpos(A))
    timeit("lqpos (D x 2D)", lambda: be.lqpos(At))
    be.set_qr_mode(1)
    timeit("qrpos householder", lambda: be.qrpos(A), n=2)
    be.set_qr_mode(0)
    vs = [r(D, d, D) for _ in range(9)]
    timeit("gs_step k=8", lambda: be.gs_step(vs[:8], vs[8]))
    timeit("norm", lambda: be.norm(x))
    timeit("lincomb k=8", lambda: be.lincomb(vs[:8], np.ones(8), out=y))
    timeit("gemm AL*C (2D x D x D)", lambda: be.gemm(A, r(D, D)))
    ws = krylov.KrylovWorkspace(be)
    eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
    h = mk.MPO_ddAC(be, H[0], GL, GR)
    timeit("eigsolve (8 matvecs)", lambda: alg.fixedpoint(be, h, x, eig, ws), n=3)
    print("qr stats", be.qr_stats())


main()
