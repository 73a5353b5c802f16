"""per-sweep energies of the interleaved-native and the bond-embedded complex one-site DMRG (same start, same eigensolver)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpskit_jl_amd as mk
from mpskit_jl_amd import native_cplx as nc, algorithms as alg, krylov
L, D = int(sys.argv[1]), int(sys.argv[2])
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
ref = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
As = [ref.download(ref.AL(i)) for i in range(L - 1)] + [ref.download(ref.AC(L - 1))]
for name, eig in (("tol 1e-12", mk.Arnoldi(tol=1e-12, krylovdim=30, maxiter=100)), ("8 matvecs", mk.Arnoldi(fixed_matvecs=8, krylovdim=8))):
    pe = mk.FiniteMPS(As, normalize=True, be=be); ee = mk.FinEnv(pe, H)
    pn = nc.NativeFiniteMPS(As, be); en = nc.NativeFinEnv(pn, H)
    ws = krylov.KrylovWorkspace(be)
    for s in range(3):
        alg.dmrg_sweep(pe, H, ee, eig, ws)
        Ee = float(np.sum(mk.expectation_value(pe, H, ee)))
        En = nc.dmrg_sweep(pn, H, en, eig, ws)
        print(f"{name}  sweep {s + 1}: embedded {Ee:.12f}  native {En:.12f}  diff {abs(Ee - En):.1e}", flush=True)
