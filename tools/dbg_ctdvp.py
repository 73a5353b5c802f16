import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpskit_jl_amd as mk
be = mk.Backend(0)
L, D = int(sys.argv[1]), int(sys.argv[2])
H = mk.heisenberg_XXX(0.5, be=be)
for mode in sys.argv[3:]:
    os.environ["MPSK_NATIVE_CPLX"] = mode
    psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
    envs = mk.FinEnv(psi, H)
    e0 = float(np.sum(mk.expectation_value(psi, H, envs)))
    for k in range(2):
        psi, envs = mk.timestep(psi, H, 0.05 * k, 0.05, mk.TDVP(tol=1e-10), envs)
    e1 = float(np.sum(mk.expectation_value(psi, H, envs)))
    print(f"L={L} D={D} native={mode}: drift {abs(e1-e0):.2e} norm-1 {psi.norm()-1:.2e} qr {be.qr_stats()}", flush=True)
