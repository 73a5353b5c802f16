import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import cplx
be = mk.Backend(0)
L, D = int(sys.argv[1]), int(sys.argv[2])
H = mk.heisenberg_XXX(0.5, be=be)
psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
sd = lambda t: cplx.structure_defect(be.download(t))
print("qr", be.qr_stats())
print("AL defects", ["%.0e" % sd(psi.AL(i)) for i in range(L)])
print("qr", be.qr_stats())
print("AR defects", ["%.0e" % sd(psi.AR(i)) for i in range(L)])
print("qr", be.qr_stats())
envs = mk.FinEnv(psi, H)
def envdef(t):
    h = be.download(t)   # (W, Db, Dk) logical -> each slab [Db, Dk]
    return max(cplx.structure_defect(h[w]) for w in range(h.shape[0]))
print("GL defects", ["%.0e" % envdef(envs.leftenv(i, psi)) for i in range(L)])
print("GR defects", ["%.0e" % envdef(envs.rightenv(i, psi)) for i in range(L)])
