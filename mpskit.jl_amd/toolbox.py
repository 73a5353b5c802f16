"""variance(state, H, envs)  (src/algorithms/toolbox.jl:128-172): energy variance of a FiniteMPS / InfiniteMPS / finite
quasiparticle state.  H * H is the odim^2-level product MPOHamiltonian (operators.MPOHamiltonian.__mul__), its expectation
value runs through the same environment / matvec kernels with W = odim^2 MPO levels."""
from __future__ import annotations

import numpy as np

from .algorithms import expectation_value
from .environments import environments
from .states import FiniteMPS


def variance(state, H, envs=None):
    from .quasiparticle import LeftGaugedQP, QuasiparticleAnsatz, _QPContext
    if isinstance(state, LeftGaugedQP):
        if not state.finite or not state.trivial:
            raise NotImplementedError("variance of an infinite / domain-wall quasiparticle state")
        # toolbox.jl:153-155 converts the state to a FiniteMPS of twice the bond dimension; the same number from the
        # tangent-space machinery: with H' = H - E0 / L,  <phi|H'^2|phi> = <X|H_eff[H' * H'] X> + <gs|H'^2|gs>,
        # <phi|H'|phi> = <X|H_eff[H'] X>  for <X|X> = 1
        be, gs = state.be, state.left_gs
        envs = environments(gs, H) if envs is None else envs
        L = len(gs)
        Hr = H - float(np.sum(expectation_value(gs, H, envs))) / L
        x = be.copy(state.vec)
        be.scal(1.0 / be.norm(x), x)
        vals = []
        for Hx in (Hr, Hr * Hr):
            ctx = _QPContext(Hx, state, environments(gs, Hx), QuasiparticleAnsatz())
            y = ctx.heff(state, x, be.zeros(x.shape))
            vals.append(be.dot(x, y) + ctx.E[0])
        return float(vals[1] - vals[0] ** 2)
    envs = environments(state, H) if envs is None else envs
    if isinstance(state, FiniteMPS):                                              # :140-144
        H2 = H * H
        return float(np.sum(expectation_value(state, H2, environments(state, H2))) - np.sum(expectation_value(state, H, envs)) ** 2)
    Hr = H - expectation_value(state, H, envs)                                    # :135-138
    H2 = Hr * Hr
    return float(np.sum(expectation_value(state, H2, environments(state, H2))))


def _xlogx_trace(be, c):
    """-tr(rho log rho) with rho = c c^dag from the singular values of c (mpsk_tsvd)."""
    _, S, _, kept, _ = be.tsvd(c)
    s2 = np.asarray(be.download(S)).reshape(-1)[:kept] ** 2
    s2 = s2[s2 > 0]
    return float(-np.sum(s2 * np.log(s2)))


def entanglement_spectrum(state, site=0):
    """entanglement_spectrum(psi; site)  (toolbox.jl:60-75): singular values of the bond matrix CR[site]."""
    c = state.CR(site) if isinstance(state, FiniteMPS) else state.CR[site % len(state)]
    _, S, _, kept, _ = state.be.tsvd(c)
    return np.asarray(state.be.download(S)).reshape(-1)[:kept]


def entropy(state, loc=None):
    """entropy(state[, loc])  (toolbox.jl:1-5): -tr(rho log rho) of the bond right of site loc; all bonds of the unit cell for
    an InfiniteMPS without loc."""
    be = state.be
    if loc is None:
        if isinstance(state, FiniteMPS):
            raise TypeError("entropy(state::FiniteMPS) needs a site")
        return [_xlogx_trace(be, c) for c in state.CR]
    return _xlogx_trace(be, state.CR(loc) if isinstance(state, FiniteMPS) else state.CR[loc % len(state)])


def transfer_spectrum(above, below=None, tol=1e-10, num_vals=20, krylovdim=None, rng=None):
    """transfer_spectrum(above; below, num_vals)  (toolbox.jl:44-58): leading eigenvalues of v -> v * T(above.AL, below.AL)
    through the unit cell (mpsk_transfer_left without an MPO); complex in general."""
    from . import krylov
    below = above if below is None else below
    be = above.be
    n = len(above)
    Da, Db = above.AL[0].shape[0], below.AL[0].shape[0]
    rng = np.random.default_rng(0) if rng is None else rng
    x0 = be.upload(rng.standard_normal((1, Db, Da))).reshape(1, Db, Da)
    num_vals = min(num_vals, Da * Db)

    def op(x, out):
        y = x
        for s in range(n):
            y = be.transfer_left(None, y, above.AL[s], below.AL[s])
        be.axpby(1.0, y, 0.0, out)
        return out
    kd = max(3 * num_vals + 10, 40) if krylovdim is None else krylovdim
    vals, conv = krylov.arnoldi_eigvals(be, op, x0, num=num_vals, tol=tol, krylovdim=kd)
    if conv < num_vals:
        import warnings
        warnings.warn(f"correlation length failed to converge: {conv} of {num_vals} values")     # :54-55
    return vals


def marek_gap(spectrum, tol_angle=0.1):
    """marek_gap(spectrum)  (toolbox.jl:77-113): inverse correlation length eps, the gap delta between the two leading
    values at the dominant angle, and that angle."""
    spectrum = np.asarray(spectrum, dtype=complex)
    spectrum = spectrum[np.abs(spectrum) < 1 - 1e-12]
    order = np.argsort(-np.abs(spectrum))
    spectrum = spectrum[order]
    angles = np.angle(spectrum)
    angles = np.where(angles < -1e-12, angles + 2 * np.pi, angles)
    theta = angles[0]
    at = spectrum[np.abs(angles - theta) < tol_angle]
    lambdas = -np.log(np.abs(at))
    delta = lambdas[1] - lambdas[0] if len(lambdas) > 2 else np.inf
    return float(lambdas[0]), float(delta), float(theta)


def correlation_length(above, **kw):
    """correlation_length(psi::InfiniteMPS)  (toolbox.jl:115-125)."""
    eps, _, _ = marek_gap(transfer_spectrum(above, **kw))
    return 1.0 / eps


def exact_diagonalization(H, len=None, num=1, tol=1e-12, krylovdim=30, maxiter=100, rng=None):
    """exact_diagonalization(H; len, num, which = :SR)  (src/algorithms/ED.jl:4-53): the largest possible FiniteMPS of that
    length (identity isometries left and right of the middle site), so that the effective Hamiltonian of the middle site IS
    the full Hamiltonian; its lowest eigenpairs by the Krylov solver over mpsk_dAC.  Returns (energies, states)."""
    from . import krylov
    from .derivatives import ddAC
    L = H.period if len is None else int(len)
    be, d = H.be, H.d
    rng = np.random.default_rng(0) if rng is None else rng
    # ED.jl:13 takes middle_site = round(len / 2); any site works (no truncation anywhere).  Here the balanced one, so that
    # every tensor of the state is a tall (Dl d >= Dr) / wide (Dl <= d Dr) matrix the QRpos / LQpos kernels accept.
    mid = (L - 1) // 2
    As, left = [], 1
    for _ in range(mid):                                                          # :23-27
        As.append(np.eye(left * d).reshape(left, d, left * d, order="F"))
        left *= d
    rights, right = [], 1
    for _ in range(L - 1, mid, -1):                                               # :28-33
        rights.append(np.eye(right * d).reshape(right * d, d, right, order="F"))
        right *= d
    As.append(rng.standard_normal((left, d, right)))                              # :34-37
    As.extend(reversed(rights))
    state = FiniteMPS(As, normalize=True, be=be)
    envs = environments(state, H)
    H_ac = ddAC(mid, state, H, envs)                                              # "this linear operator is now the actual full hamiltonian"
    found, vals, states = [], [], []
    ws = krylov.KrylovWorkspace(be)
    for _ in range(num):
        def op(x, out):
            H_ac(x, out=out)
            for f, lf in zip(found, vals):                                        # states already found are shifted up out of the way
                be.axpby((10.0 + 10.0 * abs(lf)) * be.dot(f, x), f, 1.0, out)
            return out
        lam, v, _, res = krylov.eigsolve_sr(be, op, state.AC(mid), tol=tol, krylovdim=krylovdim, maxiter=maxiter, ws=ws)
        v = be.copy(v)
        found.append(v)
        vals.append(float(lam))
        cs = state.copy()
        cs.set_AC(mid, be.copy(v))
        states.append(cs)
    return vals, states
