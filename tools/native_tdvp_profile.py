"""cProfile of native_cplx.tdvp_step (host side): python tools/native_tdvp_profile.py [L] [D]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import native_cplx as nc
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
ref = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
As = [ref.download(ref.AL(i)) for i in range(L - 1)] + [ref.download(ref.AC(L - 1))]
psi = nc.NativeFiniteMPS(As, be); envs = nc.NativeFinEnv(psi, H)
alg = mk.TDVP(tol=1e-10)
psi, envs = nc.tdvp_step(psi, H, envs, 0.0, 0.05, alg)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for k in range(2):
    psi, envs = nc.tdvp_step(psi, H, envs, 0.05 * (k + 1), 0.05, alg)
torch.cuda.synchronize(); pr.disable()
print(f"{(time.perf_counter() - t0) / 2:.3f} s per step")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue())
