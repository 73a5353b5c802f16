"""How close to orthonormal is X = AC_new * inv(R_prev) (R_prev: the triangular factor of the same site's previous visit in the
same direction)?  Decides whether a warm-started (right-preconditioned) CholeskyQR would save Cholesky chains:
max |X^T X - I| <= 1e-7 -> first-order pass only (no chain), <= 0.5 -> one chain, else two (as now).
usage: python tools/warm_gauge_probe.py [L] [D] [sweeps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import algorithms as alg, krylov
L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(1), be=be)
envs = mk.FinEnv(psi, H)
eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
ws = krylov.KrylovWorkspace(be)
tt = lambda x: x.buf[:x.shape[0] * x.shape[1]].view(x.shape[1], x.shape[0]).T
prev = {"qr": {}, "lq": {}}
cur = {"qr": [], "lq": []}
idx = {"qr": 0, "lq": 0}
o_qr2, o_lq = be.qrpos2, be.lqpos
def qr2(a1, a2):
    out = o_qr2(a1, a2)
    A, R = tt(a2).clone(), tt(out[3]).clone()
    i = idx["qr"]; idx["qr"] += 1
    if i in prev["qr"] and prev["qr"][i].shape == R.shape:
        X = torch.linalg.solve_triangular(prev["qr"][i], A, upper=True, left=False)
        G = X.T @ X
        cur["qr"].append(float((G - torch.eye(G.shape[0], dtype=G.dtype, device=G.device)).abs().max()))
    prev["qr"][i] = R
    return out
def lq(a):
    out = o_lq(a)
    A, Lm = tt(a).clone(), tt(out[0]).clone()
    i = idx["lq"]; idx["lq"] += 1
    if i in prev["lq"] and prev["lq"][i].shape == Lm.shape and A.shape[0] >= 512:
        X = torch.linalg.solve_triangular(prev["lq"][i], A, upper=False, left=True)       # inv(L_prev) A
        G = X @ X.T
        cur["lq"].append(float((G - torch.eye(G.shape[0], dtype=G.dtype, device=G.device)).abs().max()))
    prev["lq"][i] = Lm
    return out
be.qrpos2, be.lqpos = qr2, lq
for s in range(NS):
    idx["qr"] = idx["lq"] = 0; cur["qr"], cur["lq"] = [], []
    eps = alg.dmrg_sweep(psi, H, envs, eig, ws)
    E = float(np.sum(mk.expectation_value(psi, H, envs)))
    def desc(v):
        if not v: return "n/a"
        v = np.array(v)
        return f"n={len(v)} median {np.median(v):.1e} max {v.max():.1e}  <=1e-7: {np.mean(v <= 1e-7):.0%}  <=0.5: {np.mean(v <= 0.5):.0%}"
    print(f"sweep {s + 1}: E = {E:.12f} max galerkin {max(eps):.1e} | QR  {desc(cur['qr'])} | LQ  {desc(cur['lq'])}", flush=True)
