import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import algorithms as alg, cplx, krylov
from mpskit_jl_amd.derivatives import MPO_ddAC, MPO_ddC
be = mk.Backend(0)
L, D = int(sys.argv[1]), int(sys.argv[2])
H = mk.heisenberg_XXX(0.5, be=be)
psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
envs = mk.FinEnv(psi, H)
pos = L // 2
ac = psi.AC(pos)
GL, GR = envs.leftenv(pos, psi), envs.rightenv(pos, psi)
emb = MPO_ddAC(be, H[pos], GL, GR)
nat = cplx.HalfEmbeddedOp(be, "AC", [H[pos]], GL, GR)
d = lambda a, b: np.abs(be.download(a) - be.download(b)).max() / np.abs(be.download(b)).max()
print("op diff", d(nat(ac), emb(ac)), "structure defect of emb(ac)", cplx.structure_defect(be.download(emb(ac))))
for tol, kd in ((1e-10, 30), (1e-13, 60)):
    a = alg.TDVP(tol=tol, krylovdim=kd)
    y_n = alg.integrate(be, nat, ac, 0.0, 0.025, a, krylov.KrylovWorkspace(be), True)
    y_e = alg.integrate(be, emb, ac, 0.0, 0.025, a, krylov.KrylovWorkspace(be), True)
    print(f"tol {tol}: nat vs emb {d(y_n, y_e):.2e}  |y_n| {be.norm(y_n)/np.sqrt(2):.14f} |y_e| {be.norm(y_e)/np.sqrt(2):.14f} defect(y_e) {cplx.structure_defect(be.download(y_e)):.1e}")
    if tol == 1e-10:
        keep = (y_n, y_e)
    else:
        print(f"   vs reference: nat(1e-10) {d(keep[0], y_n):.2e}  emb(1e-10) {d(keep[1], y_n):.2e}")
