"""complex128 states on interleaved storage (mpskit.jl_amd/native_cplx.py: everything through the MPSK_C128 entry points of
the C ABI) against the bond-embedded host representation (cplx.py): real-time TDVP step and one-site DMRG sweep.
usage: python tools/bench_native_cplx.py [L] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import native_cplx as nc, algorithms as alg, krylov
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
rng = np.random.default_rng(5)
ref = mk.FiniteMPS.random(L, 2, D, rng, be=be, dtype=complex)          # embedded host state (cplx.py)
As = [ref.download(ref.AL(i)) for i in range(L - 1)] + [ref.download(ref.AC(L - 1))]
sync = torch.cuda.synchronize
# ---- embedded
envs = mk.FinEnv(ref, H)
e0 = float(np.sum(mk.expectation_value(ref, H, envs)))
ref, envs = mk.timestep(ref, H, 0.0, 0.05, mk.TDVP(tol=1e-10), envs)
sync(); t0 = time.perf_counter()
for k in range(2):
    ref, envs = mk.timestep(ref, H, 0.05 * (k + 1), 0.05, mk.TDVP(tol=1e-10), envs)
sync(); dt_e = (time.perf_counter() - t0) / 2
e1 = float(np.sum(mk.expectation_value(ref, H, envs)))
mem_e = 8 * sum((ref.ALs[i] or ref.ARs[i] or ref.ACs[i]).size for i in range(L))
# ---- native interleaved
psi = nc.NativeFiniteMPS(As, be)
nenv = nc.NativeFinEnv(psi, H)
n0 = nc.energy(psi, nenv)
talg = mk.TDVP(tol=1e-10)
psi, nenv = nc.tdvp_step(psi, H, nenv, 0.0, 0.05, talg)
sync(); t0 = time.perf_counter()
for k in range(2):
    psi, nenv = nc.tdvp_step(psi, H, nenv, 0.05 * (k + 1), 0.05, talg)
sync(); dt_n = (time.perf_counter() - t0) / 2
n1 = nc.energy(psi, nenv)
print(f"real-time TDVP step, complex Heisenberg L={L} D={D}: embedded host {dt_e:.3f} s (drift {abs(e1 - e0):.1e}, state {mem_e / 2**20:.0f} MiB)  |  "
      f"interleaved native {dt_n:.3f} s (drift {abs(n1 - n0):.1e}, state {psi.bytes() / 2**20:.0f} MiB, norm {psi.norm():.12f})  ->  {dt_e / dt_n:.2f}x", flush=True)
# ---- one-site DMRG sweep (8 matvecs per site)
eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
ws = krylov.KrylovWorkspace(be)
ref2 = mk.FiniteMPS(As, normalize=True, be=be)
env2 = mk.FinEnv(ref2, H)
alg.dmrg_sweep(ref2, H, env2, eig, ws)
sync(); t0 = time.perf_counter()
alg.dmrg_sweep(ref2, H, env2, eig, ws)
sync(); ds_e = time.perf_counter() - t0
Ee = float(np.sum(mk.expectation_value(ref2, H, env2)))
psi = nc.NativeFiniteMPS(As, be)
nenv = nc.NativeFinEnv(psi, H)
nc.dmrg_sweep(psi, H, nenv, eig, ws)
sync(); t0 = time.perf_counter()
En = nc.dmrg_sweep(psi, H, nenv, eig, ws)
sync(); ds_n = time.perf_counter() - t0
print(f"one-site DMRG sweep (8 matvecs / site): embedded host {ds_e:.3f} s (E = {Ee:.10f})  |  interleaved native {ds_n:.3f} s (E = {En:.10f})  ->  {ds_e / ds_n:.2f}x", flush=True)
