"""Secondary measurements for the other BASELINE.json configs (not the bench.py line):
  c2  Heisenberg S=1 FiniteMPS L=100 D=256, 1-site DMRG sweep (fixed Krylov budget 8)
  c3  transverse-field Ising InfiniteMPS D=512, VUMPS iterations (dAC/dC eigsolves + gauge + envs)
  c4  Hubbard FiniteMPS D=1024-class two-site update: dAC2 matvec + tsvd at (D*d) x (d*D) (reduced L)
usage: python tools/bench_configs.py [c2] [c3] [c4] [tsvd]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mpskit_jl_amd as mk
from mpskit_jl_amd import algorithms as alg, krylov


def sync():
    torch.cuda.synchronize()


def c2(be):
    L, D, d = 100, 256, 3
    H = mk.heisenberg_XXX(1.0, be=be)
    psi = mk.FiniteMPS.random(L, d, D, np.random.default_rng(1), be=be)
    envs = mk.FinEnv(psi, H)
    eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
    ws = krylov.KrylovWorkspace(be)
    alg.dmrg_sweep(psi, H, envs, eig, ws)
    sync(); t0 = time.perf_counter()
    for _ in range(2):
        alg.dmrg_sweep(psi, H, envs, eig, ws)
    sync(); dt = (time.perf_counter() - t0) / 2
    print(f"c2 Heisenberg S=1 L=100 D=256 d=3: {dt:.3f} s/sweep = {1 / dt:.3f} sweeps/s", flush=True)


def c3(be):
    D = 512
    H = mk.transverse_field_ising(1.0, 0.5, be=be)
    psi = mk.InfiniteMPS.random(2, D, np.random.default_rng(2), be=be)
    sync(); t0 = time.perf_counter()
    p, e, eps = mk.find_groundstate(psi, H, mk.VUMPS(tol=1e-10, maxiter=6, verbosity=3))
    sync(); dt = time.perf_counter() - t0
    E = float(np.sum(mk.expectation_value(p, H, e)))
    print(f"c3 VUMPS iTFI D=512: 6 iterations in {dt:.2f} s ({dt / 6:.2f} s/iter), e = {E:.12f}, galerkin = {eps:.2e}", flush=True)


def c4(be):
    D, d = 1024, 4
    H = mk.hubbard(1.0, 4.0, be=be)
    W = H[0].Wl
    r = lambda *s: mk.DTensor(torch.rand(int(np.prod(s)), dtype=torch.float64, device=be.device) - 0.5, s)
    GL, GR = r(W, D, D), r(W, D, D)
    x2, y2 = r(D, d, D, d), be.empty(D, d, D, d)
    be.dAC2(H[0], H[0], GL, GR, x2, out=y2)
    sync(); t0 = time.perf_counter()
    for _ in range(3):
        be.dAC2(H[0], H[0], GL, GR, x2, out=y2)
    sync(); dt = (time.perf_counter() - t0) / 3
    fl = 4 * W * d * d * D ** 3 + 4 * W * W * d ** 3 * D * D
    print(f"c4 dAC2 D=1024 d=4 W=6: {dt * 1e3:.2f} ms = {fl / dt / 1e12:.1f} TFLOP/s (algorithmic {fl / 1e9:.0f} GF)", flush=True)


def c4sweep(be, L=16, D=256, nsweeps=2):
    """Two-site DMRG sweeps (Hubbard U/t = 4, d = 4) with tsvd! truncation to D: time split eigsolve / tsvd and the
    Jacobi sweep counts on the theta tensors a real run produces (graded Schmidt spectra)."""
    H = mk.hubbard(1.0, 4.0, be=be)
    psi = mk.FiniteMPS.random(L, 4, D, np.random.default_rng(3), be=be)
    stat = {"svd_s": 0.0, "svd_n": 0, "sweeps": [], "eig_s": 0.0, "paths": [], "iters": []}
    orig_tsvd, orig_fp = be.tsplit, alg.fixedpoint

    def tsvd_timed(*a, **k):
        sync(); t0 = time.perf_counter()
        out = orig_tsvd(*a, **k)
        sync(); stat["svd_s"] += time.perf_counter() - t0; stat["svd_n"] += 1
        stat["sweeps"].append(be.svd_sweeps())
        st = be.split_stats()
        stat["paths"].append(st["path"]); stat["iters"].append(st["iterations"])
        if os.environ.get("C4_VERBOSE"):
            sv = out[3]
            print(f"  split {a[0].shape} keep {len(sv)}: {(time.perf_counter() - t0) * 1e3:7.1f} ms  path {st['path']} iters {st['iterations']:2d} "
                  f"resid {st['residual']:.1e}  jacobi sweeps {be.svd_sweeps():2d}  s_k/s_1 {sv[-1] / sv[0]:.1e}  disc {out[4]:.2e}", flush=True)
        return out

    def fp_timed(*a, **k):
        sync(); t0 = time.perf_counter()
        out = orig_fp(*a, **k)
        sync(); stat["eig_s"] += time.perf_counter() - t0
        return out

    be.tsplit, alg.fixedpoint = tsvd_timed, fp_timed
    try:
        for it in range(nsweeps):
            for k in stat:
                stat[k] = [] if k in ("sweeps", "paths", "iters") else 0
            q0 = be.qr_stats()
            sync(); t0 = time.perf_counter()
            psi, envs, eps = mk.find_groundstate(psi, H, mk.DMRG2(tol=1e-14, maxiter=1, trunc_dim=D,
                                                                  eigalg=mk.Arnoldi(fixed_matvecs=8, krylovdim=8)))
            sync(); dt = time.perf_counter() - t0
            q1 = be.qr_stats()
            sw = stat["sweeps"]
            print(f"c4sweep Hubbard L={L} D={D} sweep {it + 1}: {dt:.2f} s  eigsolve {stat['eig_s']:.2f} s  tsvd {stat['svd_s']:.2f} s "
                  f"({stat['svd_n']} calls, Jacobi sweeps min/mean/max {min(sw)}/{np.mean(sw):.1f}/{max(sw)}), "
                  f"split paths full/subspace/gave-up {stat['paths'].count(0)}/{stat['paths'].count(1)}/{stat['paths'].count(2)}, "
                  f"subspace iterations mean {np.mean([i for i, p in zip(stat['iters'], stat['paths']) if p == 1] or [0]):.1f}, "
                  f"qr +{ {k: q1[k] - q0[k] for k in q1} }, max bond {max(psi.bond_dims())}", flush=True)
    finally:
        be.tsplit, alg.fixedpoint = orig_tsvd, orig_fp


def ctdvp(be, L=32, D=128):
    """real-time TDVP step on a COMPLEX state (bond-embedded, cplx.py): Heisenberg S=1/2, complex D -> real 2D."""
    H = mk.heisenberg_XXX(0.5, be=be)
    rng = np.random.default_rng(5)
    psi = mk.FiniteMPS.random(L, 2, D, rng, be=be, dtype=complex)
    envs = mk.FinEnv(psi, H)
    e0 = float(np.sum(mk.expectation_value(psi, H, envs)))
    psi, envs = mk.timestep(psi, H, 0.0, 0.05, mk.TDVP(tol=1e-10), envs)
    sync(); t0 = time.perf_counter()
    nst = 2
    for k in range(nst):
        psi, envs = mk.timestep(psi, H, 0.05 * (k + 1), 0.05, mk.TDVP(tol=1e-10), envs)
    sync(); dt = (time.perf_counter() - t0) / nst
    e1 = float(np.sum(mk.expectation_value(psi, H, envs)))
    print(f"ctdvp complex Heisenberg L={L} D={D} (embedded {2 * D}): {dt:.3f} s per real-time TDVP step, "
          f"energy drift {abs(e1 - e0):.2e}, norm {psi.norm():.12f}", flush=True)


def tsvd(be):
    for n in (512, 1024, 2048, 4096):
        A = mk.DTensor(torch.rand(n * n, dtype=torch.float64, device=be.device), (n, n))
        sync(); t0 = time.perf_counter()
        U, S, Vh, kept, disc = be.tsvd(A, max_keep=n // 4)
        sync(); dt = time.perf_counter() - t0
        print(f"tsvd {n}x{n} keep {n // 4}: {dt * 1e3:.1f} ms", flush=True)


def main():
    be = mk.Backend(0)
    which = sys.argv[1:] or ["c2", "c3", "c4", "tsvd"]
    for w in which:                       # name[:int args], e.g. c4sweep:16:1024
        name, *a = w.split(":")
        globals()[name](be, *[int(v) for v in a])


main()
