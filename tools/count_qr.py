"""Who calls the gauge factorizations during one DMRG sweep (call sites + counts) -- a host-logic audit tool."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpskit_jl_amd as mk
from mpskit_jl_amd import algorithms as alg, krylov
L, D = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 256
be = mk.Backend(0)
sites = collections.Counter()
def wrap(name):
    f = getattr(be, name)
    def w(*a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        sites[(name, " <- ".join(f"{s.name}:{s.lineno}" for s in reversed(st[-4:])))] += 1
        return f(*a, **k)
    setattr(be, name, w)
for n in ("qrpos", "lqpos", "qrpos2"):
    wrap(n)
psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(1), be=be)
H = mk.heisenberg_XXX(0.5, be=be)
envs = mk.FinEnv(psi, H)
eig = mk.Arnoldi(fixed_matvecs=4, krylovdim=4)
ws = krylov.KrylovWorkspace(be)
alg.dmrg_sweep(psi, H, envs, eig, ws)
sites.clear(); s0 = be.qr_stats()
alg.dmrg_sweep(psi, H, envs, eig, ws)
print("stats delta", {k: be.qr_stats()[k] - s0[k] for k in s0})
for k, v in sites.most_common():
    print(v, k)
