"""two-site DMRG sweep of a complex state: interleaved storage (native_cplx.dmrg2_sweep: mpsk_dAC2 complex + mpsk_tsplit under
MPSK_C128) against the bond-embedded host (DMRG2 on FiniteMPS(dtype=complex): cplx.split_two_site).
usage: python tools/bench_native_cplx2.py [L] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
from mpskit_jl_amd import native_cplx as nc, krylov
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
be = mk.Backend(0)
H = mk.heisenberg_XXX(0.5, be=be)
ref = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
As = [ref.download(ref.AL(i)) for i in range(L - 1)] + [ref.download(ref.AC(L - 1))]
sync = torch.cuda.synchronize
eig = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
pe = mk.FiniteMPS(As, normalize=True, be=be)
alg2 = mk.DMRG2(tol=1e-14, maxiter=1, trunc_dim=D, eigalg=eig)
pe, ee, _ = mk.find_groundstate(pe, H, alg2)
sync(); t0 = time.perf_counter()
pe, ee, _ = mk.find_groundstate(pe, H, alg2, ee)
sync(); dt_e = time.perf_counter() - t0
Ee = float(np.sum(mk.expectation_value(pe, H, ee)))
pn = nc.NativeFiniteMPS(As, be); en = nc.NativeFinEnv(pn, H)
ws = krylov.KrylovWorkspace(be)
nc.dmrg2_sweep(pn, H, en, eig, D, ws)
sync(); t0 = time.perf_counter()
En = nc.dmrg2_sweep(pn, H, en, eig, D, ws)
sync(); dt_n = time.perf_counter() - t0
print(f"two-site DMRG sweep, complex Heisenberg L={L} D={D} (8 matvecs / site): embedded host {dt_e:.3f} s (E = {Ee:.10f})  |  "
      f"interleaved native {dt_n:.3f} s (E = {En:.10f})  ->  {dt_e / dt_n:.2f}x", flush=True)
