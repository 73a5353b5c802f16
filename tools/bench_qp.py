"""Timing of the quasiparticle effective Hamiltonian (mpskit.jl_amd/quasiparticle.py) on the HIP path: one application of
H_eff = qp environments (4 n slab transfers + 2 triangular transfer systems with their GMRES solves) + 3 mpsk_dAC per
site, at a real-phase momentum (pi: one real part) and a generic one (1.0: real + imaginary parts).
usage: python tools/bench_qp.py [D=256] [model=heis1|tfi]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mpskit_jl_amd as mk  # noqa: E402
from mpskit_jl_amd import quasiparticle as qp  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = sys.argv[2] if len(sys.argv) > 2 else "heis1"
be = mk.default_backend()
H = mk.heisenberg_XXX(1.0, be=be) if model == "heis1" else mk.transverse_field_ising(1.0, 2.0, be=be)
d = 3 if model == "heis1" else 2
psi = mk.InfiniteMPS.random(d, D, np.random.default_rng(0), be=be)
psi, envs, eps = mk.find_groundstate(psi, H, mk.VUMPS(tol=1e-6, maxiter=8))
W = H[0].Wl
print(f"model {model} D={D} d={d} W={W}  vumps galerkin {eps:.2e}")
for p in (np.pi, 1.0):
    phi = qp.LeftGaugedQP.random(psi, p, np.random.default_rng(1))
    ctx = qp._QPContext(H, phi, envs, qp.QuasiparticleAnsatz())
    out = be.empty(phi.vec.shape)
    ctx.heff(phi, phi.vec, out)
    be.synchronize()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        ctx.heff(phi, phi.vec, out)
    be.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    # breakdown by backend entry point (each call synchronised: the sum exceeds the asynchronous total)
    names = ["transfer_left", "transfer_right", "dAC", "gemm", "lincomb", "axpby", "orth_step", "gs_step", "multidot", "dot",
             "norm", "regularize", "copy", "scal", "mposlice", "empty", "zeros"]
    acc, orig = {}, {}
    for nm in names:
        if not hasattr(be, nm):
            continue
        orig[nm] = getattr(be, nm)

        def wrap(*a, _f=orig[nm], _nm=nm, **kw):
            be.synchronize()
            t0 = time.perf_counter()
            r = _f(*a, **kw)
            be.synchronize()
            c = acc.setdefault(_nm, [0, 0.0])
            c[0] += 1
            c[1] += (time.perf_counter() - t0) * 1e3
            return r
        setattr(be, nm, wrap)
    t = time.perf_counter()
    ctx.heff(phi, phi.vec, out)
    be.synchronize()
    tot = (time.perf_counter() - t) * 1e3
    for nm, f in orig.items():
        setattr(be, nm, f)
    print(f"   synchronised total {tot:.1f} ms:", {k: (v[0], round(v[1], 1)) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])})
    # flops of the pieces that do not depend on the GMRES iteration counts: 3 dAC + 4 slab transfers per part
    fl = phi.nparts * 7 * 4.0 * W * d * D ** 3
    print(f"p={p:.3f} parts={phi.nparts}  H_eff apply {ms:8.2f} ms   (>= {fl / ms / 1e9:6.2f} TF/s counting 3 dAC + 4 transfers only)")
