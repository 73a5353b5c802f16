"""Ground-state drivers with the reference's API: find_groundstate(psi, H, DMRG()/DMRG2()/VUMPS())
(src/algorithms/groundstate/dmrg.jl:14-141, vumps.jl:18-92, find_groundstate.jl:19-41), plus
calc_galerkin (toolbox.jl:17-25) and expectation_value (expval.jl:92-124).  Orchestration only:
every flop happens inside libmpsk."""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field

import numpy as np

from .backend import DTensor
from .derivatives import ddAC, ddAC2, ddC
from .environments import FinEnv, MPOHamInfEnv, MultipleEnvironments, environments
from .states import FiniteMPS, InfiniteMPS, leftorth, mul_AC
from . import krylov


@dataclass
class Arnoldi:  # Defaults.eigsolver  (defaults.jl:33)
    tol: float = 1e-12
    maxiter: int = 100
    krylovdim: int = 30
    eager: bool = True
    fixed_matvecs: int | None = None   # benchmark mode: exactly this many matvecs per solve


@dataclass
class DMRG:  # dmrg.jl:14-20
    tol: float = 1e-12
    maxiter: int = 100
    eigalg: Arnoldi = field(default_factory=Arnoldi)
    verbosity: int = 0
    finalize: object = None


@dataclass
class DMRG2:  # dmrg.jl:71-78
    tol: float = 1e-12
    maxiter: int = 100
    eigalg: Arnoldi = field(default_factory=Arnoldi)
    trunc_dim: int = 0            # truncdim(D); 0 = no limit
    trunc_err: float = 1e-6       # truncerr(1e-6) default (dmrg.jl:75); ignored when trunc_dim > 0 unless set
    verbosity: int = 0
    finalize: object = None


@dataclass
class VUMPS:  # vumps.jl:18-27 with the dynamic tolerances of defaults.jl:38-57
    tol: float = 1e-12
    maxiter: int = 100
    verbosity: int = 0
    krylovdim: int = 30
    eig_tol_min: float = 1e-12
    eig_tol_max: float = 1e-5
    eig_tol_factor: float = 1e-5
    gauge_tol_min: float = 1e-14
    gauge_tol_max: float = 1e-5
    gauge_tol_factor: float = 1e-8
    env_tol_min: float = 1e-12
    env_tol_max: float = 1e-5
    env_tol_factor: float = 1e-5
    finalize: object = None


def updatetol(tol_min, tol_max, factor, it, eps):  # dynamictols.jl:50-53
    return min(max(eps * factor / math.sqrt(it), tol_min), tol_max)


def fixedpoint(be, A, x0, alg: Arnoldi, ws=None, first_image=None, values=True):
    """fixedpoint(A, x0, :SR, alg)  (fixedpoint.jl:19-30); non-convergence only warns.
    A native complex operator (cplx.HalfEmbeddedOp) given an EMBEDDED start tensor is solved on the half-embedded
    (interleaved complex) vectors: encode once, iterate, decode once."""
    wrap = hasattr(A, "is_half") and not A.is_half(x0)
    if wrap:
        x0 = A.encode(x0)
        fi, first_image = first_image, (None if first_image is None else be.empty(*x0.shape))
    lam, vec, nmv, res = krylov.eigsolve_sr(be, A, x0, tol=alg.tol, krylovdim=alg.krylovdim,
                                            maxiter=alg.maxiter, fixed_matvecs=alg.fixed_matvecs, ws=ws,
                                            first_image=first_image, values=values)
    if wrap:
        vec = A.decode(vec)
        if fi is not None:
            A.decode(first_image, out=fi)
    return lam, vec


# ---- measurements ------------------------------------------------------------------------------

def _galerkin(be, h, ac, al, g=None, slot=None, offset=0):
    """|| (1 - AL AL^dag) normalize(h(AC)) ||  for explicit tensors.
    g: h(AC) up to a positive factor if the caller already has it (the eigensolver's first matvec).
    slot: device buffer that receives the SQUARED result at `offset` instead of a host read-back (the sweep collects its
    2L - 2 values with one download at the end); the return value is then None."""
    g = h(ac) if g is None else g
    if hasattr(be, "normalize_dev"):
        be.normalize_dev(g)
    else:
        be.scal(1.0 / be.norm(g), g)
    Dl, d, Dr = al.shape
    alm, gm = al.reshape(Dl * d, Dr), g.reshape(Dl * d, g.shape[2])
    t = be.gemm(alm, gm, transA=True)
    be.gemm(alm, t, alpha=-1.0, beta=1.0, out=gm)
    if slot is not None:
        be.nrm2_dev(gm, slot, offset)
        return None
    return be.norm(gm)


def calc_galerkin(psi, pos, envs, h=None, g=None):
    """|| (1 - AL AL^dag) normalize(H_AC AC) ||   (toolbox.jl:17-22).
    h: the site's effective Hamiltonian if the caller already built it (same operator);
    g: H_AC AC (any positive multiple) if the caller already has it."""
    be = psi.be
    if isinstance(psi, FiniteMPS):
        ac, al = psi.AC(pos), psi.AL(pos)
    else:
        ac, al = psi.AC[pos], psi.AL[pos]
    if h is None and g is None:
        h = ddAC(pos, psi, envs.H, envs)
    return _galerkin(be, h, ac, al, g)


def expectation_value(psi, H, envs):
    """Per-site energies (expval.jl:92-109 finite, :111-124 infinite)."""
    be = psi.be
    if isinstance(envs, MultipleEnvironments):          # expval of a LazySum: sum of the terms' (lazysum.jl)
        return sum(f * expectation_value(psi, h, e) for f, h, e in zip(H.fs, H, envs.envs))
    if hasattr(envs, "vector"):                         # ProjectionOperator term (excitations.py)
        from .excitations import _expval_projection
        return _expval_projection(psi, H, envs)
    if isinstance(psi, FiniteMPS):
        L = len(psi)
        ens = np.zeros(L)
        for i in range(L):
            ac = psi.AC(i)
            y = be.dAC(H.energy_slice(i), envs.leftenv(i, psi), envs.rightenv(i, psi), ac)
            ens[i] = be.dot(ac, y)
        n2 = be.norm(psi.AC(L - 1)) ** 2
        return ens / n2
    n, odim = len(psi), H.odim
    ens = np.zeros(n)
    for i in range(n):
        GL = envs.leftenv(i, psi)
        # last column of H[i] applied to the left env, closed with r_LL = C C^dag  (expval.jl:111-124)
        col = {(j, 0): H[i].blocks[(j, odim - 1)] for j in range(odim) if H[i].contains(j, odim - 1)}
        blk = be.mposlice(odim, H.d, H[i].chil, [H[i].chir[odim - 1]] + [1] * (odim - 1), col)
        apl = be.transfer_left(blk, GL, psi.AL[i], psi.AL[i])
        c = psi.CR[i]
        r = be.gemm(c, c, transB=True)
        first = DTensor(apl.buf, (apl.shape[1], apl.shape[2]))   # slab 0
        ens[i] = be.dot(first, DTensor(r.buf, first.shape))      # sum apl[x,y] r[y,x], r symmetric
    return ens


# ---- DMRG ---------------------------------------------------------------------------------------

def _log(alg, name, it, E, eps, t0):
    if alg.verbosity >= 3:
        print(f"[ Info: {name} {it:3d}:\tobj = {E:+.12e}\terr = {eps:.10e}\ttime = {time.time() - t0:.2f} sec", flush=True)


def _no_cplx(psi, what):
    if getattr(psi, "cplx", False):
        raise NotImplementedError(f"{what} on a complex (embedded) state: tsvd of the embedding defines singular vectors "
                                  "only up to a rotation inside each doubled singular value (cplx.py)")


def find_groundstate(psi, H, alg=None, envs=None):
    """find_groundstate(psi, H, alg[, envs]) -> (psi, envs, eps)  (find_groundstate.jl:19-41)."""
    if alg is None:
        alg = DMRG() if isinstance(psi, FiniteMPS) else VUMPS()
    if isinstance(alg, DMRG):
        return _dmrg(psi.copy(), H, alg, envs)
    if isinstance(alg, DMRG2):
        return _dmrg2(psi.copy(), H, alg, envs)
    if isinstance(alg, VUMPS):
        return _vumps(psi, H, alg, envs)
    if isinstance(alg, IDMRG1):
        return _idmrg1(psi, H, alg, envs)
    if isinstance(alg, IDMRG2):
        return _idmrg2(psi, H, alg, envs)
    raise TypeError(f"unknown algorithm {alg!r}")


# Galerkin evaluation of left-moving visits on the ctx's second stream (mpsk_ctx_side_*), under the CholeskyQR chain of the
# LQ step.  MEASURED AND LEFT OFF (MI355X, same-box A/B of bench.py, three runs each): 0.6458 / 0.6413 / 0.6466 sweeps/s with
# it, 0.6476 / 0.6442 / 0.6472 without -- the GEMMs of the projection delay the latency-bound step kernels they run beside
# by as much as they hide (the same outcome as the two-stream Jacobi schedule of mpsk_tsplit).  MPSK_SIDE_STREAM=1 enables it.
_SIDE_STREAM = os.environ.get("MPSK_SIDE_STREAM", "0") == "1"


def dmrg_sweep(psi, H, envs, eigalg: Arnoldi, ws=None):
    """One full DMRG sweep, pos in [1:L-1; L:-1:2] (dmrg.jl:33-38): per site one eigsolve with the
    effective Hamiltonian, one galerkin evaluation, and the lazy gauge / environment updates that
    the next site triggers.  Returns the per-site galerkin errors.
    envs: FinEnv, or dist.ShardedFinEnv (bond-sharded multi-GPU sweep): its `site_op` returns a matvec on vectors in
    the blocked layout together with `encode` / `decode` (rows <-> rank blocks, once per site visit)."""
    be = psi.be
    L = len(psi)
    visits = list(range(0, L - 1)) + list(range(L - 1, 0, -1))
    # galerkin norms stay on the device until the sweep is over (one read-back instead of two stream stalls per site)
    defer = hasattr(be, "nrm2_dev")
    gslot = (ws if ws is not None else krylov.KrylovWorkspace(be)).get((len(visits), 1), 1, tag="galerkin")[0] if defer else None   # (own tag: never a solver buffer)
    eps_s = [0.0] * L
    for iv, pos in enumerate(visits):
        h = envs.site_op(pos, psi) if hasattr(envs, "site_op") else ddAC(pos, psi, H, envs)
        ac_old = psi.AC(pos)
        enc = getattr(h, "encode", None)
        x0 = ac_old if enc is None else enc(ac_old)
        # the eigensolver's first matvec is H_AC (AC_old / |AC_old|): exactly the vector calc_galerkin of the old
        # tensor normalises (toolbox.jl:18), so it is captured instead of applying H_AC to AC_old a second time
        g = be.empty(*x0.shape)
        _, vec = fixedpoint(be, h, x0, eigalg, ws, first_image=g, values=False)   # the sweep only uses the vector
        if enc is not None:
            vec, g = h.decode(vec), h.decode(g)
        # Deferred gauge step (mpsk_ctx_qr_defer / mpsk_qr_commit) on left-moving visits: the LQ factorization the NEXT
        # visit needs is enqueued first, the galerkin evaluation of this visit (which does not depend on it) behind it,
        # and only then does the host wait for the factorization's success flag -- the stream has ~0.2 ms of work while
        # it does (the flag read-back used to leave the GPU idle for ~40 us per visit); the evaluation itself runs on the
        # ctx's second stream, under the latency-bound factorization chain.  Same tensors, same operations.
        qdef = defer and hasattr(be, "qr_defer") and not psi.cplx
        if psi.ALs[pos] is None and pos < L - 1:
            # right-moving visit: leftorth(old AC) (galerkin projector) and leftorth(new AC) (next AL)
            # are both due -> issue them together; same state as the lazy views would produce
            # (not deferred: the galerkin evaluation needs al_old itself, and speculating on it does not pay -- measured
            #  on the benchmark sweep, a third of the old-AC factorizations repeat their last pass (orthogonality of the
            #  first-order pass > 1e-7 on these ill-conditioned tensors), and every repeat costs a second projection)
            al_old = psi.set_AC_with_leftorth(pos, vec)
            e = _galerkin(be, h, ac_old, al_old, g, gslot, iv)
        else:
            # left-moving visit (and the turning point pos = L-1, where the next step is a rightorth): the galerkin
            # projector is leftorth(old AC) (cached AL[pos] if the state holds it), the new AC is stored as is
            # and gauged when the next site asks for AR[pos] (orthoview.jl:27-31, 49-54) -- here: right away, so
            # that the galerkin evaluation runs behind it
            al_old = psi.AL(pos)
            psi.set_AC(pos, vec)
            side = False
            if qdef and pos > 0:
                side = hasattr(be, "side_begin") and _SIDE_STREAM
                if side:
                    be.side_mark()          # the galerkin evaluation depends on nothing enqueued after this point
                be.qr_defer()
                psi.AR(pos)
                if side:
                    be.side_begin()         # ... so it runs on the second stream, under the CholeskyQR chain of the LQ step
            e = _galerkin(be, h, ac_old, al_old, g, gslot, iv)
            if side:
                be.side_end()
            if qdef:
                be.qr_commit()
        if not defer:
            eps_s[pos] = max(eps_s[pos], e)
    if defer:
        vals = np.sqrt(np.maximum(np.asarray(be.download(gslot)).reshape(-1), 0.0))
        for iv, pos in enumerate(visits):
            eps_s[pos] = max(eps_s[pos], float(vals[iv]))
    return eps_s


def _dmrg(psi, H, alg: DMRG, envs=None):  # dmrg.jl:22-55
    be = psi.be
    envs = environments(psi, H) if envs is None else envs
    L = len(psi)
    ws = krylov.KrylovWorkspace(be)
    eps_s = [calc_galerkin(psi, p, envs) for p in range(L)]
    eps = max(eps_s)
    t0 = time.time()
    history = []
    for it in range(1, alg.maxiter + 1):
        eps_s = dmrg_sweep(psi, H, envs, alg.eigalg, ws)
        eps = max(eps_s)
        if alg.finalize is not None:
            psi, envs = alg.finalize(it, psi, H, envs)
        if alg.verbosity >= 3 or eps <= alg.tol or it == alg.maxiter:
            E = float(np.sum(expectation_value(psi, H, envs)))
            history.append((it, E, eps))
            _log(alg, "DMRG", it, E, eps, t0)
        if eps <= alg.tol:
            break
    envs.history = history
    return psi, envs, eps


def _two_site_tensor(be, left: DTensor, right: DTensor):
    """theta[a,s1,b,s2] = sum_m left[a,s1,m] right[m,s2,b]   (dmrg.jl:92 / :108)."""
    Dl, d1, Dm = left.shape
    _, d2, Dr = right.shape
    theta = be.empty(Dl, d1, Dr, d2)
    lm = left.reshape(Dl * d1, Dm)
    for s2 in range(d2):
        # right[:, s2, :] is a (Dm x Dr) matrix with leading dimension Dm*d2 at offset s2*Dm
        be.gemm_raw(False, False, Dl * d1, Dr, Dm, 1.0, lm.ptr, Dl * d1, right.ptr + 8 * s2 * Dm, Dm * d2, 0.0,
                    theta.ptr + 8 * s2 * Dl * d1 * Dr, Dl * d1)
    return theta


def _dmrg2(psi, H, alg: DMRG2, envs=None):  # dmrg.jl:80-137
    be = psi.be
    envs = environments(psi, H) if envs is None else envs
    L = len(psi)
    ws = krylov.KrylovWorkspace(be)
    eps = np.inf
    t0 = time.time()
    history = []
    trunc_err = alg.trunc_err if alg.trunc_dim <= 0 else 0.0

    def update(pos, ac2):
        h = ddAC2(pos, psi, H, envs)
        _, new = fixedpoint(be, h, ac2, alg.eigalg, ws)
        Dl, d1, Dr, d2 = new.shape
        if getattr(psi, "cplx", False):                       # complex state: structured split of the embedding
            from .cplx import split_two_site
            al, c, ar, _, _ = split_two_site(be, new, alg.trunc_dim, trunc_err)
            be.scal(np.sqrt(2.0) / be.norm(c), c)             # normalize!(c) (embedded Frobenius norm = sqrt 2)
            k = c.shape[0]
            t = be.gemm(al.reshape(Dl * d1, k), c)
            arm = be.empty(k, Dr * d2)
            for s2 in range(d2):
                be.copy2d(k, Dr, ar.ptr + 8 * s2 * k, k * d2, arm.ptr + 8 * s2 * k * Dr, k)
            rec = be.gemm(t, arm)
            v = be.dot(ac2, DTensor(rec.buf, ac2.shape)) / 2.0
            return al, c, ar, abs(1 - abs(v))
        alm, c, arm, _, _ = be.tsplit(new.reshape(Dl * d1, Dr * d2), max_keep=alg.trunc_dim, trunc_err=trunc_err)
        k = c.shape[0]
        be.scal(1.0 / be.norm(c), c)                            # normalize!(c)  (|c|_F = |S_kept|)
        al = alm.reshape(Dl, d1, k)
        ar = be.empty(k, d2, Dr)                                # ar[k, s2, b] = arm[k, (b, s2)]
        for s2 in range(d2):
            be.copy2d(k, Dr, arm.ptr + 8 * s2 * k * Dr, k, ar.ptr + 8 * s2 * k, k * d2)
        # fidelity  v = <ac2, al c ar>   (dmrg.jl:98-100)
        rec = be.gemm(be.gemm(alm, c), arm)
        v = be.dot(ac2, DTensor(rec.buf, ac2.shape))
        return al, c, ar, abs(1 - abs(v))

    for it in range(1, alg.maxiter + 1):
        eps_s = [0.0] * L
        for pos in range(0, L - 1):
            ac2 = _two_site_tensor(be, psi.AC(pos), psi.AR(pos + 1))
            al, c, ar, e = update(pos, ac2)
            eps_s[pos] = max(eps_s[pos], e)
            psi.set_AC(pos, (al, c))
            psi.set_AC(pos + 1, (c, ar))
        for pos in range(L - 3, -1, -1):
            ac2 = _two_site_tensor(be, psi.AL(pos), psi.AC(pos + 1))
            al, c, ar, e = update(pos, ac2)
            eps_s[pos] = max(eps_s[pos], e)
            psi.set_AC(pos + 1, (c, ar))
            psi.set_AC(pos, (al, c))
        eps = max(eps_s)
        if alg.finalize is not None:
            psi, envs = alg.finalize(it, psi, H, envs)
        if alg.verbosity >= 3 or eps <= alg.tol or it == alg.maxiter:
            E = float(np.sum(expectation_value(psi, H, envs)))
            history.append((it, E, eps))
            _log(alg, "DMRG2", it, E, eps, t0)
        if eps <= alg.tol:
            break
    envs.history = history
    return psi, envs, eps


# ---- VUMPS ----------------------------------------------------------------------------------------

def regauge(be, AC: DTensor, C: DTensor, cplx=False):
    """regauge!(AC, C; alg = QRpos()) -> AL = Q_AC Q_C^dag   (ortho.jl:127-131)."""
    from .states import _qr
    Dl, d, Dr = AC.shape
    Qac, _ = _qr(be, AC.reshape(Dl * d, Dr), cplx)
    Qc, _ = _qr(be, C, cplx)
    return be.gemm(Qac, Qc, transB=True).reshape(Dl, d, Dr)


def _calc_galerkin_inf(psi, envs):
    return max(calc_galerkin(psi, loc, envs) for loc in range(len(psi)))


def _vumps(psi, H, alg: VUMPS, envs=None):  # vumps.jl:29-92
    be = psi.be
    envs = environments(psi, H) if envs is None else envs
    eps = _calc_galerkin_inf(psi, envs)
    n = len(psi)
    ws = krylov.KrylovWorkspace(be)
    t0 = time.time()
    history = []
    for it in range(1, alg.maxiter + 1):
        eig = Arnoldi(tol=updatetol(alg.eig_tol_min, alg.eig_tol_max, alg.eig_tol_factor, it, eps),
                      krylovdim=alg.krylovdim)
        newAL = []
        for loc in range(n):
            _, AC = fixedpoint(be, ddAC(loc, psi, H, envs), psi.AC[loc], eig, ws)
            _, C = fixedpoint(be, ddC(loc, psi, H, envs), psi.CR[loc], eig, ws)
            newAL.append(regauge(be, AC, C, getattr(psi, "cplx", False)))
        gtol = updatetol(alg.gauge_tol_min, alg.gauge_tol_max, alg.gauge_tol_factor, it, eps)
        psi = InfiniteMPS.from_AL(newAL, psi.CR[n - 1], tol=gtol, be=be, cplx=getattr(psi, "cplx", False))
        etol = updatetol(alg.env_tol_min, alg.env_tol_max, alg.env_tol_factor, it, eps)
        envs.recalculate(psi, etol)
        if alg.finalize is not None:
            psi, envs = alg.finalize(it, psi, H, envs)
        eps = _calc_galerkin_inf(psi, envs)
        if alg.verbosity >= 3 or eps <= alg.tol or it == alg.maxiter:
            E = float(np.sum(expectation_value(psi, H, envs)))
            history.append((it, E, eps))
            _log(alg, "VUMPS", it, E, eps, t0)
        if eps <= alg.tol:
            break
    envs.history = history
    return psi, envs, eps


# ---- time evolution (src/algorithms/timestep/tdvp.jl, integrators.jl, time_evolve.jl) ---------------

@dataclass
class TDVP:  # tdvp.jl:14-19 ; integrator = Lanczos(; tol = Defaults.tol)
    tol: float = 1e-12
    krylovdim: int = 30
    maxiter: int = 100
    tolgauge: float = 1e-14
    gaugemaxiter: int = 100
    finalize: object = None


@dataclass
class TDVP2:  # tdvp.jl:105-111 ; trscheme = truncerr(1e-3)
    tol: float = 1e-12
    krylovdim: int = 30
    maxiter: int = 100
    trunc_dim: int = 0
    trunc_err: float = 1e-3
    finalize: object = None


def integrate(be, f, y0, t, dt, alg, ws=None, cplx=False):
    """integrate(f, y0, t, dt, Lanczos)  (integrators.jl:20-25): y = exp(-im*dt*f) y0.
    Real states: only steps with real -im*dt (imaginary time, dt = -1j*tau) keep the tensors real.
    Embedded complex states (cplx.py): any complex dt;  exp(z f) with z = -im*dt = zr + i zi is the exponential of
    the real generator  x -> zr f(x) + zi (i f(x))  on the embedded tensors (general Arnoldi)."""
    z = -1j * complex(dt)
    enc = getattr(f, "encode", None)
    if enc is not None and cplx:
        # native complex128 operator (cplx.HalfEmbeddedOp): iterate on the half-embedded (= interleaved complex) vector
        yh = _integrate_embedded(be, f.apply_half, enc(y0), z, alg, ws)
        return f.decode(yh)
    if not cplx:
        if abs(z.imag) > 0.0:
            raise NotImplementedError("real-time evolution of a REAL state leaves the reals: build the state from "
                                      "complex tensors (FiniteMPS(..., dtype=complex)) or pass dt = -1j*tau")
        y, _ = krylov.exponentiate(be, f, z.real, y0, tol=alg.tol, krylovdim=alg.krylovdim, maxiter=alg.maxiter, ws=ws)
        return y
    return _integrate_embedded(be, f, y0, z, alg, ws)


def _integrate_embedded(be, f, y0, z, alg, ws):
    """exp(z f) y0 on (half-)embedded complex tensors: multiplication by i is (I (x) J) on the interleaved row pairs."""
    if z.imag == 0.0:
        y, _ = krylov.exponentiate(be, f, z.real, y0, tol=alg.tol, krylovdim=alg.krylovdim, maxiter=alg.maxiter, ws=ws)
        return y
    from .cplx import times_i
    tmp = be.empty(*y0.shape)

    def gen(x, out):
        f(x, tmp)
        times_i(be, tmp, out)                 # out = i f(x)
        be.axpby(z.real, tmp, z.imag, out)    # out = zr f(x) + zi (i f(x))
        return out

    y, _ = krylov.exponentiate_general(be, gen, y0, tol=alg.tol, krylovdim=alg.krylovdim, maxiter=alg.maxiter, ws=ws)
    return y


def _timestep_tdvp(psi, H, t, dt, alg: TDVP, envs):  # tdvp.jl:61-94
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be)
    cx = getattr(psi, "cplx", False)
    for i in range(L - 1):
        psi.set_AC(i, integrate(be, ddAC(i, psi, H, envs), psi.AC(i), t, dt / 2, alg, ws, cx))
        psi.set_CR(i, integrate(be, ddC(i, psi, H, envs), psi.CR(i), t, -dt / 2, alg, ws, cx))
    psi.set_AC(L - 1, integrate(be, ddAC(L - 1, psi, H, envs), psi.AC(L - 1), t, dt / 2, alg, ws, cx))
    for i in range(L - 1, 0, -1):
        psi.set_AC(i, integrate(be, ddAC(i, psi, H, envs), psi.AC(i), t + dt / 2, dt / 2, alg, ws, cx))
        psi.set_CR(i - 1, integrate(be, ddC(i - 1, psi, H, envs), psi.CR(i - 1), t + dt / 2, -dt / 2, alg, ws, cx))
    psi.set_AC(0, integrate(be, ddAC(0, psi, H, envs), psi.AC(0), t + dt / 2, dt / 2, alg, ws, cx))
    return psi, envs


def _split_two_site(be, nac2, alg, cx=False):
    """tsvd!(nac2; trunc) -> (al, c, ar) with ar[k, s2, b]   (tdvp.jl:124-126)."""
    Dl, d1, Dr, d2 = nac2.shape
    if cx:
        from .cplx import split_two_site
        al, c, ar, _, _ = split_two_site(be, nac2, alg.trunc_dim, alg.trunc_err if alg.trunc_dim <= 0 else 0.0)
        return al, c, ar
    trunc_err = alg.trunc_err if alg.trunc_dim <= 0 else 0.0
    alm, c, arm, _, _ = be.tsplit(nac2.reshape(Dl * d1, Dr * d2), max_keep=alg.trunc_dim, trunc_err=trunc_err)
    k = c.shape[0]
    ar = be.empty(k, d2, Dr)
    for s2 in range(d2):
        be.copy2d(k, Dr, arm.ptr + 8 * s2 * k * Dr, k, ar.ptr + 8 * s2 * k, k * d2)
    return alm.reshape(Dl, d1, k), c, ar


def _timestep_tdvp2(psi, H, t, dt, alg: TDVP2, envs):  # tdvp.jl:113-146
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be)
    cx = getattr(psi, "cplx", False)
    for i in range(L - 1):
        ac2 = _two_site_tensor(be, psi.AC(i), psi.AR(i + 1))
        al, c, ar = _split_two_site(be, integrate(be, ddAC2(i, psi, H, envs), ac2, t, dt / 2, alg, ws, cx), alg, cx)
        psi.set_AC(i, (al, c))
        psi.set_AC(i + 1, (c, ar))
        if i != L - 2:
            psi.set_AC(i + 1, integrate(be, ddAC(i + 1, psi, H, envs), psi.AC(i + 1), t, -dt / 2, alg, ws, cx))
    for i in range(L - 1, 0, -1):
        ac2 = _two_site_tensor(be, psi.AL(i - 1), psi.AC(i))
        al, c, ar = _split_two_site(be, integrate(be, ddAC2(i - 1, psi, H, envs), ac2, t + dt / 2, dt / 2, alg, ws, cx), alg, cx)
        psi.set_AC(i - 1, (al, c))
        psi.set_AC(i, (c, ar))
        if i != 1:
            psi.set_AC(i - 1, integrate(be, ddAC(i - 1, psi, H, envs), psi.AC(i - 1), t + dt / 2, -dt / 2, alg, ws, cx))
    return psi, envs


def _timestep_inf(psi, H, t, dt, alg: TDVP, envs):  # tdvp.jl:21-59 (leftorthflag = true)
    be, n = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be)
    newAL = []
    for loc in range(n):
        ac = integrate(be, ddAC(loc, psi, H, envs), psi.AC[loc], t, dt, alg, ws)
        c = integrate(be, ddC(loc, psi, H, envs), psi.CR[loc], t, dt, alg, ws)
        newAL.append(regauge(be, ac, c, getattr(psi, "cplx", False)))
    psi2 = InfiniteMPS.from_AL(newAL, psi.CR[n - 1], tol=alg.tolgauge, maxiter=alg.gaugemaxiter, be=be,
                               cplx=getattr(psi, "cplx", False))
    envs.recalculate(psi2)
    return psi2, envs


def timestep(psi, H, t, dt, alg=None, envs=None):
    """timestep(psi, H, t, dt, alg[, envs]) -> (psi', envs): the copying version (tdvp.jl:148-151)."""
    alg = TDVP() if alg is None else alg
    if isinstance(psi, InfiniteMPS):
        envs = environments(psi, H) if envs is None else envs
        return _timestep_inf(psi, H, t, dt, alg, envs)
    psi = psi.copy()
    envs = environments(psi, H) if envs is None else envs
    if isinstance(alg, TDVP2):
        return _timestep_tdvp2(psi, H, t, dt, alg, envs)
    return _timestep_tdvp(psi, H, t, dt, alg, envs)


def time_evolve(psi, H, t_span, alg=None, envs=None):
    """time_evolve(psi, H, t_span, alg[, envs])  (time_evolve.jl:20-40): consecutive timesteps."""
    alg = TDVP() if alg is None else alg
    for t0, t1 in zip(t_span[:-1], t_span[1:]):
        psi, envs = timestep(psi, H, t0, t1 - t0, alg, envs)
        if alg.finalize is not None:
            psi, envs = alg.finalize(t0, psi, H, envs)
    return psi, envs


# ---- IDMRG1 (src/algorithms/groundstate/idmrg.jl:21-77, src/environments/idmrgenv.jl) ---------------------

@dataclass
class IDMRG1:  # idmrg.jl:13-19
    tol: float = 1e-12
    tol_gauge: float = 1e-14
    maxiter: int = 100
    krylovdim: int = 30
    verbosity: int = 0
    eig_tol_min: float = 1e-12
    eig_tol_max: float = 1e-5
    eig_tol_factor: float = 1e-5


class IDMRGEnv:
    """idmrgenv.jl:6-49: private copies of the converged infinite environments, updated by hand with the
    MPO transfer kernels (no regularisation, no dependency checks)."""

    def __init__(self, psi, env: MPOHamInfEnv):
        if env.dependency is not psi:
            env.recalculate(psi)
        be, n = psi.be, len(psi)
        self.be, self.n, self.H = be, n, env.H
        self.opp = [env.H[s] for s in range(n)]
        self.lw = [be.copy(env._assemble("l", s, n)) for s in range(n)]
        self.rw = [be.copy(env._assemble("r", s, n)) for s in range(n)]

    def leftenv(self, pos, psi=None):
        return self.lw[pos % self.n]

    def rightenv(self, pos, psi=None):
        return self.rw[pos % self.n]

    def update_leftenv(self, psi, pos):   # idmrgenv.jl:69-72 : lw[pos] = lw[pos-1] * TM(AL[pos-1], H[pos-1])
        p = (pos - 1) % self.n
        self.lw[pos % self.n] = self.be.transfer_left(self.opp[p], self.lw[p], psi.AL[p], psi.AL[p])

    def update_rightenv(self, psi, pos):  # idmrgenv.jl:64-67 : rw[pos] = TM(AR[pos+1], H[pos+1]) * rw[pos+1]
        p = (pos + 1) % self.n
        self.rw[pos % self.n] = self.be.transfer_right(self.opp[p], self.rw[p], psi.AR[p], psi.AR[p])


def _idmrg1(ost, H, alg: IDMRG1, oenvs=None):  # idmrg.jl:21-77
    from .states import rightorth
    be = ost.be
    oenvs = MPOHamInfEnv(ost, H) if oenvs is None else oenvs
    eps = _calc_galerkin_inf(ost, oenvs)
    n = len(ost)
    psi = InfiniteMPS(list(ost.AL), list(ost.AR), list(ost.CR), list(ost.AC), be)
    envs = IDMRGEnv(ost, oenvs)
    ws = krylov.KrylovWorkspace(be)
    t0 = time.time()
    for it in range(1, alg.maxiter + 1):
        eig = Arnoldi(tol=updatetol(alg.eig_tol_min, alg.eig_tol_max, alg.eig_tol_factor, it, eps), krylovdim=alg.krylovdim)
        c_cur = psi.CR[n - 1]
        for pos in range(n):
            _, psi.AC[pos] = fixedpoint(be, ddAC(pos, psi, H, envs), psi.AC[pos], eig, ws)
            psi.AL[pos], psi.CR[pos] = leftorth(be, psi.AC[pos])
            envs.update_leftenv(psi, pos + 1)
        for pos in range(n - 1, -1, -1):
            _, psi.AC[pos] = fixedpoint(be, ddAC(pos, psi, H, envs), psi.AC[pos], eig, ws)
            psi.CR[(pos - 1) % n], psi.AR[pos] = rightorth(be, psi.AC[pos])
            envs.update_rightenv(psi, pos - 1)
        new = psi.CR[n - 1]
        if new.shape == c_cur.shape:
            diff = be.copy(new)
            be.axpby(-1.0, c_cur, 1.0, diff)
            eps = be.norm(diff)
        else:
            eps = 1.0
        if alg.verbosity >= 3:
            print(f"[ Info: IDMRG {it:3d}:\terr = {eps:.10e}\ttime = {time.time() - t0:.2f} sec", flush=True)
        if eps < alg.tol:
            break
    nst = InfiniteMPS.from_tensors(psi.AR, tol=alg.tol_gauge, be=be)
    return nst, MPOHamInfEnv(nst, H), eps


# ---- IDMRG2 (idmrg.jl:79-204) -----------------------------------------------------------------------------

@dataclass
class IDMRG2:  # idmrg.jl:89-96 ; trscheme = truncerr(1e-6)
    tol: float = 1e-12
    tol_gauge: float = 1e-14
    maxiter: int = 100
    krylovdim: int = 30
    verbosity: int = 0
    trunc_dim: int = 0
    trunc_err: float = 1e-6
    eig_tol_min: float = 1e-12
    eig_tol_max: float = 1e-5
    eig_tol_factor: float = 1e-5


def _bond_inv(be, C: DTensor):
    """inv(C) of a bond matrix (idmrg.jl:118,150 `inv(psi.CR[..])`) on the device: C = U S Vh by mpsk_tsvd,
    inv(C) = Vh^T diag(1/S) U^T by two GEMMs (only the D reciprocals pass through the host)."""
    U, S, Vh, k, _ = be.tsvd(C)
    sv = np.asarray(be.download(S)).reshape(-1)
    W = be.gemm(Vh, be.upload(np.diag(1.0 / sv)), transA=True)
    return be.gemm(W, U, transB=True)


def _idmrg2(ost, H, alg: IDMRG2, oenvs=None):  # idmrg.jl:97-204
    from .derivatives import MPO_ddAC2
    from .states import mul_CA
    be = ost.be
    n = len(ost)
    if n < 2:
        raise ValueError("unit cell should be >= 2")
    _no_cplx(ost, "IDMRG2")
    oenvs = MPOHamInfEnv(ost, H) if oenvs is None else oenvs
    eps = _calc_galerkin_inf(ost, oenvs)
    psi = InfiniteMPS(list(ost.AL), list(ost.AR), list(ost.CR), list(ost.AC), be)
    envs = IDMRGEnv(ost, oenvs)
    ws = krylov.KrylovWorkspace(be)
    t0 = time.time()

    def solve(ac2, pl, pr, eig):
        h = MPO_ddAC2(be, envs.opp[pl], envs.opp[pr], envs.lw[pl], envs.rw[pr])
        _, new = fixedpoint(be, h, ac2, eig, ws)
        al, c, ar = _split_two_site(be, new, alg)
        be.scal(1.0 / be.norm(c), c)                                    # normalize!(c)
        return al, c, ar

    for it in range(1, alg.maxiter + 1):
        eig = Arnoldi(tol=updatetol(alg.eig_tol_min, alg.eig_tol_max, alg.eig_tol_factor, it, eps), krylovdim=alg.krylovdim)
        for pos in range(n - 1):
            al, c, ar = solve(_two_site_tensor(be, psi.AC[pos], psi.AR[pos + 1]), pos, pos + 1, eig)
            psi.AL[pos], psi.CR[pos], psi.AR[pos + 1] = al, c, ar
            psi.AC[pos + 1] = mul_CA(be, c, ar)
            envs.update_leftenv(psi, pos + 1)
            envs.update_rightenv(psi, pos)
        # edge (sites n-1, 0):  AC[end] inv(CR[end]) . AL[1] CR[1]
        left = mul_AC(be, psi.AC[n - 1], _bond_inv(be, psi.CR[n - 1]))
        right = mul_AC(be, psi.AL[0], psi.CR[0])
        al, c, ar = solve(_two_site_tensor(be, left, right), n - 1, 0, eig)
        psi.AC[n - 1] = mul_AC(be, al, c)
        psi.AL[n - 1], psi.CR[n - 1], psi.AR[0] = al, c, ar
        psi.AC[0] = mul_CA(be, c, ar)
        psi.AL[0] = mul_AC(be, psi.AC[0], _bond_inv(be, psi.CR[0]))
        c_cur = c
        envs.update_leftenv(psi, 0)
        envs.update_rightenv(psi, n - 1)
        for pos in range(n - 2, -1, -1):
            al, c, ar = solve(_two_site_tensor(be, psi.AL[pos], psi.AC[pos + 1]), pos, pos + 1, eig)
            psi.AL[pos], psi.CR[pos], psi.AR[pos + 1] = al, c, ar
            psi.AC[pos] = mul_AC(be, al, c)
            psi.AC[pos + 1] = mul_CA(be, c, ar)
            envs.update_leftenv(psi, pos + 1)
            envs.update_rightenv(psi, pos)
        # edge again:  CR[end-1] AR[end] . inv(CR[end]) AC[1]
        left = mul_CA(be, psi.CR[n - 2], psi.AR[n - 1])
        right = mul_CA(be, _bond_inv(be, psi.CR[n - 1]), psi.AC[0])
        al, c, ar = solve(_two_site_tensor(be, left, right), n - 1, 0, eig)
        psi.AR[n - 1] = mul_CA(be, _bond_inv(be, psi.CR[n - 2]), mul_AC(be, al, c))
        psi.AL[n - 1], psi.CR[n - 1], psi.AR[0] = al, c, ar
        psi.AC[0] = mul_CA(be, c, ar)
        envs.update_leftenv(psi, 0)
        envs.update_rightenv(psi, n - 1)
        a, b = be.download(c), be.download(c_cur)
        k = min(a.shape[0], b.shape[0])
        eps = float(np.linalg.norm(a[:k, :k] - b[:k, :k]))
        if alg.verbosity >= 3:
            print(f"[ Info: IDMRG2 {it:3d}:\terr = {eps:.10e}\ttime = {time.time() - t0:.2f} sec", flush=True)
        if eps < alg.tol:
            break
    nst = InfiniteMPS.from_tensors(psi.AR, tol=alg.tol_gauge, be=be)
    return nst, MPOHamInfEnv(nst, H), eps
