"""cProfile of complex real-time TDVP steps (host side): python tools/ctdvp_profile.py [L] [D]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
envs = mk.FinEnv(psi, H)
psi, envs = mk.timestep(psi, H, 0.0, 0.05, mk.TDVP(tol=1e-10), envs)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(2):
    psi, envs = mk.timestep(psi, H, 0.05 * (k + 1), 0.05, mk.TDVP(tol=1e-10), envs)
torch.cuda.synchronize()
pr.disable()
print(f"{(time.perf_counter() - t0) / 2:.3f} s per step (profiled)")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue())
