"""Host-side Krylov solvers on DEVICE vectors (the KrylovKit role in the reference:
src/algorithms/fixedpoint.jl:9-30, mpohaminfenv.jl:95).  The loop stays on the host, every
matvec is one call into libmpsk, and the Gram-Schmidt step is the fused multi-dot / multi-axpy
pair fused with the normalisation in `mpsk_vorth_step` (coefficients stay on the device, ONE host sync
per Krylov iteration).
"""
from __future__ import annotations

import numpy as np

from .backend import Backend, DTensor


class KrylovWorkspace:
    """Reusable pool of device vectors so that an eigsolve allocates nothing after warm-up."""

    def __init__(self, be: Backend):
        self.be = be
        self.pool = {}

    def get(self, shape, n, tag=None):
        """n device buffers of `shape`.  Buffers are pooled by (tag, shape): two users with different tags never share
        a buffer even when their shapes coincide (the sweep's galerkin slot (2L-2, 1) and the fixed-budget solver's scalar
        slot (m (2m+1) + 40, 1) have the same shape at L = 39 for m = 4, at L = 89 for m = 8)."""
        shp = tuple(shape)
        lst = self.pool.setdefault((tag, shp), [])
        while len(lst) < n:
            lst.append(self.be.empty(*shp))
        return lst[:n]


def eigsolve_sr(be: Backend, matvec, x0: DTensor, tol=1e-12, krylovdim=30, maxiter=100,
                fixed_matvecs=None, ws: KrylovWorkspace | None = None, first_image: DTensor | None = None,
                values: bool = True):
    """Smallest-real eigenpair of a Hermitian operator: restarted Lanczos/Arnoldi with twice-iterated
    classical Gram-Schmidt and an 'eager' convergence test each step (defaults.jl:33:
    Arnoldi(; tol, maxiter, eager=true)).  matvec(x: DTensor, out: DTensor) -> out.
    first_image: optional buffer that receives A (x0 / |x0|), the very first matvec of the solve, before it is
    orthogonalised -- calc_galerkin of the OLD tensor needs exactly this vector (toolbox.jl:18), so the DMRG sweep
    does not apply the effective Hamiltonian to the same tensor twice.
    values=False (fixed-budget solves only): the caller needs the vector alone -- the Ritz step then runs on the device
    as well (mpsk_vritz_dev) and the solve returns (None, vec, n_matvecs, None) without ever stalling the stream.
    Returns (lambda, vec, n_matvecs, residual)."""
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = x0.shape
    vecs = ws.get(shape, krylovdim + 2)
    V, ritz = vecs[:krylovdim + 1], vecs[krylovdim + 1]
    nmv, lam, res = 0, 0.0, np.inf
    start = x0
    if fixed_matvecs is not None and fixed_matvecs <= krylovdim and hasattr(be, "orth_step_dev"):
        # fixed budget within one Krylov cycle: nothing is decided per step, so the whole recurrence is enqueued
        # without host synchronisation and the projected matrix is read back once (the per-step sync left the GPU
        # idle ~10 % of a site update at D = 1024)
        # a Krylov space cannot be larger than the vector space: beyond x0.size steps the (rounding-level) residual
        # would be renormalised into a noise direction and pollute the projected matrix (chain-edge sites of a
        # benchmark sweep have dimension 4 < 8)
        m = max(1, min(fixed_matvecs, x0.size))
        stride = 2 * m + 1
        if not values and m <= 32 and hasattr(matvec, "eigsolve_fixed"):
            # the whole solve as ONE library call (mpsk_hac_eigsolve_fixed): same kernels in the same order as below,
            # without the ~30 entry-point calls per site the host otherwise makes (they starve the stream at small D)
            out = be.empty(*shape)
            scal = ws.get((m * stride + 40, 1), 1, tag="eigscal")[0]
            if matvec.eigsolve_fixed(start, m, V[:m + 1] + [ritz], scal, out, first_image) is not None:
                return None, out, fixed_matvecs, None
        slot = ws.get((m * stride,), 1, tag="eigslot")[0]
        be.normalize_dev(start, out=V[0])              # no host round trip: the solve has ONE synchronisation, below
        for k in range(m):
            w = V[k + 1]
            matvec(V[k], w)
            if k == 0 and first_image is not None:
                be.axpby(1.0, w, 0.0, first_image)
            be.orth_step_dev(V[:k + 1], w, slot, k * stride)
        if not values and m <= 32 and hasattr(be, "ritz_dev"):
            rb = ws.get((be.RITZ_BUF, 1), 1, tag="ritz")[0]
            be.ritz_dev(m, stride, slot, rb)
            be.lincomb_dev(V[:m], rb, out=ritz)
            out = be.empty(*shape)
            be.normalize_dev(ritz, out=out)
            return None, out, fixed_matvecs, None
        co = be.download(slot)
        Hm = np.zeros((m + 1, m))
        for k in range(m):
            kk = k + 1
            blk = co[k * stride:k * stride + 2 * kk + 1]
            Hm[:kk, k] = blk[:kk] + blk[kk:2 * kk]
            Hm[kk, k] = np.sqrt(max(blk[2 * kk], 0.0))
        # invariant subspace reached before the budget ran out (beta_k at rounding level): the later columns were
        # built from a renormalised rounding residual -> project on the exact part only
        scale = max(np.abs(Hm[:m, :m]).max(), 1e-300)
        for k in range(m):
            if Hm[k + 1, k] <= 1e-13 * scale:
                m = k + 1
                break
        Hk = Hm[:m, :m]
        ev, S = np.linalg.eigh((Hk + Hk.T) / 2)
        lam, sv = ev[0], S[:, 0]
        res = abs(Hm[m, m - 1] * sv[-1])
        be.lincomb(V[:m], sv, out=ritz)
        out = be.empty(*shape)
        be.normalize_dev(ritz, out=out)
        return lam, out, fixed_matvecs, res
    # Thick restart (Krylov-Schur for a Hermitian operator, what KrylovKit's eigsolve does with an Arnoldi / Lanczos
    # factorization; the `keep` rule below is recalled from KrylovKit's source, which is not vendored in the reference:
    # "parity unpinned", only the converged eigenpair is compared anywhere).  After krylovdim steps the basis is SHRUNK to
    # the `keep` lowest Ritz vectors plus the residual direction instead of restarting from the Ritz vector alone:
    #   A Y = Y diag(theta) + v_{m+1} b^T ,   b = beta_m S[m-1, :keep]
    # is again an Arnoldi relation, so the expansion simply continues.  On the first to-tolerance sweep of the benchmark
    # chain the single-vector restart needed 340 matvecs per site (profiles/r03_*), most of them re-learning directions
    # the restart had thrown away.
    m = int(min(krylovdim, max(x0.size, 1)))
    nrm = be.norm(start)
    be.axpby(1.0 / nrm, start, 0.0, V[0])
    basis = list(V[:m + 1])                     # basis[i] = i-th Krylov vector; the remaining pool entries are scratch
    spare = None
    Hm = np.zeros((m + 1, m))
    k, conv = 0, False
    sv = None
    # One-step-ahead expansion (device backends with orth_step_async): step k + 1 (matvec, Gram-Schmidt, normalisation -- none
    # of which needs the host) is enqueued BEFORE the host reads the scalars of step k and takes the eager convergence
    # decision, so the stream does not drain once per Krylov step.  If step k turns out to be the last one, the speculative
    # step is discarded: it only wrote basis[k + 2], which the Ritz vector of step k does not involve (at most one wasted
    # matvec per solve, counted in nmv).
    spec = hasattr(be, "orth_step_async") and fixed_matvecs is None
    agree = getattr(matvec, "agree", None)

    def account(kk, h, beta):
        """column kk of the projected matrix is known: eager convergence test of the (kk + 1)-dimensional space"""
        nonlocal lam, sv, res, S, ev
        Hm[:kk + 1, kk] = h
        Hm[kk + 1, kk] = beta
        Hk = Hm[:kk + 1, :kk + 1]
        ev, S = np.linalg.eigh((Hk + Hk.T) / 2)
        lam, sv = ev[0], S[:, 0]
        res = abs(Hm[kk + 1, :kk + 1] @ sv)               # |(last row of the (k+1) x k matrix) . s| = Ritz residual norm
        done_fixed = fixed_matvecs is not None and nmv >= fixed_matvecs
        # breakdown: the Krylov space is invariant (always the case once k reaches the vector-space dimension)
        breakdown = beta <= 1e-13 * max(np.abs(Hk).max(), 1e-300) or kk + 1 >= x0.size
        ok = (fixed_matvecs is None and res < tol) or breakdown or done_fixed
        # a bond-sharded operator (dist.ShardedSiteOp) makes the ranks AGREE on every data-dependent decision: they hold
        # bit-identical vectors by construction, the tiny all-reduce is insurance against one of them ever seeing another
        # rounding (a rank that stops while its peers enter the next all-gather would hang the job)
        return agree(ok) if (agree is not None and fixed_matvecs is None) else ok

    S = ev = None
    for _restart in range(maxiter):
        pending = None                                    # (column index, handle) whose scalars are still in flight
        while k < m:
            w = basis[k + 1]
            matvec(basis[k], w)
            if nmv == 0 and first_image is not None:
                be.axpby(1.0, w, 0.0, first_image)
            nmv += 1
            if spec:
                hnd = be.orth_step_async(basis[:k + 1], w)
                if pending is not None and account(pending[0], *pending[1].result()):
                    k, conv = pending[0] + 1, True        # converged one step earlier: drop the speculative step
                    pending = None
                    break
                pending = (k, hnd)
                k += 1
                continue
            h, beta = be.orth_step(basis[:k + 1], w)      # CGS2 + normalise, one host sync
            k += 1
            if account(k - 1, h, beta):
                conv = True
                break
        if pending is not None:
            kk = pending[0]
            if account(kk, *pending[1].result()):
                k, conv = kk + 1, True
        if conv or _restart == maxiter - 1:
            break
        # shrink: keep the lowest Ritz vectors (KrylovKit: keep = div(3 krylovdim + 2 converged, 5), converged = 0 here)
        keep = max(1, min(m - 1, (3 * m) // 5))
        if spare is None:
            spare = ws.get(shape, keep, tag="thick-restart")
        if hasattr(be, "multilincomb") and m <= 32:
            be.multilincomb(basis[:m], S[:, :keep], spare[:keep])       # one pass: m + keep vector reads / writes
        else:
            for j in range(keep):
                be.lincomb(basis[:m], S[:, j], out=spare[j])
        resid = basis[m]                                  # v_{m+1}
        coupling = Hm[m, m - 1] * S[m - 1, :keep]
        old = basis[:m]
        basis = list(spare[:keep]) + [resid] + old[:m - keep]      # m + 1 vectors again
        spare = old[m - keep:m]                                       # the other `keep` old vectors: next shrink's targets
        Hm = np.zeros((m + 1, m))
        Hm[:keep, :keep] = np.diag(ev[:keep])
        Hm[keep, :keep] = coupling
        k = keep
    be.lincomb(basis[:k], sv, out=ritz)
    out = be.empty(*shape)
    nrm = be.norm(ritz)
    if nrm == 0.0:
        raise ZeroDivisionError("eigsolve_sr: zero Ritz vector (zero start vector?)")
    be.axpby(1.0 / nrm, ritz, 0.0, out)
    return lam, out, nmv, res


def eigsolve_lm_real(be: Backend, matvec, x0: DTensor, tol=1e-12, krylovdim=30, maxiter=100,
                     ws: KrylovWorkspace | None = None):
    """Dominant (largest-magnitude) eigenpair of a real non-symmetric operator whose dominant
    eigenvector is real (transfer matrices: ortho.jl:184,241).  Restarted Arnoldi."""
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = x0.shape
    vecs = ws.get(shape, krylovdim + 2)
    V, ritz = vecs[:krylovdim + 1], vecs[krylovdim + 1]
    lam = 0.0
    start = x0
    for _restart in range(maxiter):
        nrm = be.norm(start)
        be.axpby(1.0 / nrm, start, 0.0, V[0])
        Hm = np.zeros((krylovdim + 1, krylovdim))
        k, conv = 0, False
        s = None
        while k < krylovdim:
            w = V[k + 1]
            matvec(V[k], w)
            h, beta = be.orth_step(V[:k + 1], w)
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = beta
            k += 1
            ev, S = np.linalg.eig(Hm[:k, :k])
            idx = int(np.argmax(np.abs(ev)))
            lam, sv = ev[idx], S[:, idx]
            res = abs(beta * sv[-1])
            ph = sv[np.argmax(np.abs(sv))]
            s = np.real(sv * np.conj(ph) / abs(ph))
            if res < tol or beta < 1e-300:
                conv = True
                break
        be.lincomb(V[:k], s, out=ritz)
        start = ritz
        if conv:
            break
    out = be.empty(*shape)
    nrm = be.norm(ritz)
    be.axpby(1.0 / nrm, ritz, 0.0, out)
    return float(np.real(lam)), out


def exponentiate(be: Backend, matvec, z: float, x0: DTensor, tol=1e-12, krylovdim=30, maxiter=100,
                 ws: KrylovWorkspace | None = None):
    """y = exp(z A) x0 for a real symmetric operator and REAL z (KrylovKit.exponentiate as called by
    integrators.jl:20-25 with z = -im*dt: imaginary-time steps dt = -i tau give z = -tau).
    Lanczos on device vectors with the fused CGS2 step; the small exp(z T_k) e_1 is formed on the
    host from the eigendecomposition of T_k; a-posteriori estimate beta_k |e_k^T exp(z T_k) e_1|;
    if krylovdim is exhausted the step is cut to the converged fraction and restarted.
    Returns (y, n_matvecs)."""
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = x0.shape
    vecs = ws.get(shape, krylovdim + 2)
    V, cur = vecs[:krylovdim + 1], vecs[krylovdim + 1]
    be.axpby(1.0, x0, 0.0, cur)
    remaining, nmv = 1.0, 0
    for _ in range(maxiter):
        nrm = be.norm(cur)
        if nrm == 0.0 or remaining <= 0.0:
            break
        be.axpby(1.0 / nrm, cur, 0.0, V[0])
        Hm = np.zeros((krylovdim + 1, krylovdim))
        k, s, u = 0, remaining, None
        while k < krylovdim:
            w = V[k + 1]
            matvec(V[k], w)
            nmv += 1
            h, beta = be.orth_step(V[:k + 1], w)
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = beta
            k += 1
            Tk = (Hm[:k, :k] + Hm[:k, :k].T) / 2
            ev, S = np.linalg.eigh(Tk)
            u = S @ (np.exp(remaining * z * ev) * S[0])
            if beta * abs(u[-1]) <= tol * max(remaining, 1e-300) or beta < 1e-300:
                s = remaining
                break
            if k == krylovdim:
                s = remaining
                while True:
                    u = S @ (np.exp(s * z * ev) * S[0])
                    if beta * abs(u[-1]) <= tol * s or s < 1e-12:
                        break
                    s *= 0.5
                break
        be.lincomb(V[:k], nrm * u, out=cur)
        remaining -= s
    out = be.empty(*shape)
    be.axpby(1.0, cur, 0.0, out)
    return out, nmv


def exponentiate_general(be: Backend, gen, x0: DTensor, tol=1e-12, krylovdim=30, maxiter=100,
                         ws: KrylovWorkspace | None = None):
    """y = exp(G) x0 for a general real linear generator gen(x, out) -> out (Arnoldi; the small exp of the
    Hessenberg matrix on the host).  Used for complex time steps on embedded tensors, where
    G = Im(dt) H + Re(dt) (-i H) is neither symmetric nor antisymmetric.  Same a-posteriori estimate and
    step-cutting as `exponentiate`.  Returns (y, n_matvecs)."""
    from scipy.linalg import expm
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = x0.shape
    vecs = ws.get(shape, krylovdim + 2)
    V, cur = vecs[:krylovdim + 1], vecs[krylovdim + 1]
    be.axpby(1.0, x0, 0.0, cur)
    remaining, nmv = 1.0, 0
    for _ in range(maxiter):
        nrm = be.norm(cur)
        if nrm == 0.0 or remaining <= 0.0:
            break
        be.axpby(1.0 / nrm, cur, 0.0, V[0])
        Hm = np.zeros((krylovdim + 1, krylovdim))
        k, s, u = 0, remaining, None
        while k < krylovdim:
            w = V[k + 1]
            gen(V[k], w)
            nmv += 1
            h, beta = be.orth_step(V[:k + 1], w)
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = beta
            k += 1
            u = expm(remaining * Hm[:k, :k])[:, 0]
            if beta * abs(u[-1]) <= tol * max(remaining, 1e-300) or beta < 1e-300:
                s = remaining
                break
            if k == krylovdim:
                s = remaining
                while True:
                    u = expm(s * Hm[:k, :k])[:, 0]
                    if beta * abs(u[-1]) <= tol * s or s < 1e-12:
                        break
                    s *= 0.5
                break
        be.lincomb(V[:k], nrm * u, out=cur)
        remaining -= s
    out = be.empty(*shape)
    be.axpby(1.0, cur, 0.0, out)
    return out, nmv


def gmres(be: Backend, matvec, b: DTensor, x0: DTensor, tol=1e-12, krylovdim=30, maxiter=100,
          ws: KrylovWorkspace | None = None):
    """Restarted GMRES for matvec(x) = b on device vectors (KrylovKit.linsolve stand-in,
    mpohaminfenv.jl:95,113,146,164).  Returns x (new DTensor)."""
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = b.shape
    vecs = ws.get(shape, krylovdim + 3)
    V, r, tmp = vecs[:krylovdim + 1], vecs[krylovdim + 1], vecs[krylovdim + 2]
    x = be.copy(x0)
    bnorm = be.norm(b)
    if bnorm == 0:
        return be.zeros(*shape)
    res = np.inf
    for _ in range(maxiter):
        matvec(x, tmp)
        be.axpby(1.0, b, 0.0, r)
        be.axpby(-1.0, tmp, 1.0, r)
        beta = be.norm(r)
        if beta <= tol:
            break
        be.axpby(1.0 / beta, r, 0.0, V[0])
        Hm = np.zeros((krylovdim + 1, krylovdim))
        k, y = 0, None
        spec = hasattr(be, "orth_step_async")      # one-step-ahead expansion, as in eigsolve_sr

        def account(kk, h, hn):
            nonlocal y, res
            Hm[:kk + 1, kk] = h
            Hm[kk + 1, kk] = hn
            e1 = np.zeros(kk + 2)
            e1[0] = beta
            y, *_ = np.linalg.lstsq(Hm[:kk + 2, :kk + 1], e1, rcond=None)
            res = np.linalg.norm(Hm[:kk + 2, :kk + 1] @ y - e1)
            return res <= tol or hn < 1e-300

        pending, done = None, False
        while k < krylovdim:
            w = V[k + 1]
            matvec(V[k], w)
            if spec:
                hnd = be.orth_step_async(V[:k + 1], w)
                if pending is not None and account(pending[0], *pending[1].result()):
                    k, done, pending = pending[0] + 1, True, None
                    break
                pending = (k, hnd)
                k += 1
                continue
            h, hn = be.orth_step(V[:k + 1], w)
            k += 1
            if account(k - 1, h, hn):
                break
        if pending is not None and not done:
            if account(pending[0], *pending[1].result()):
                k = pending[0] + 1
        be.lincomb(V[:k], y, out=tmp)
        be.axpby(1.0, tmp, 1.0, x)
        if res <= tol:
            break
    return x


def arnoldi_eigvals(be: Backend, matvec, x0: DTensor, num=1, tol=1e-10, krylovdim=60, ws: KrylovWorkspace | None = None):
    """The `num` largest-magnitude eigenvalues (complex in general) of a real non-symmetric operator matvec(x, out): one
    Arnoldi factorisation of dimension <= krylovdim, Ritz values of the Hessenberg matrix on the host; a Ritz value counts
    as converged when |beta * s_last| < tol (KrylovKit eigsolve(..., :LM) stand-in for `transfer_spectrum`, toolbox.jl:44-58).
    Returns (eigenvalues sorted by decreasing magnitude, number converged)."""
    ws = KrylovWorkspace(be) if ws is None else ws
    shape = x0.shape
    krylovdim = int(min(krylovdim, x0.size))
    V = ws.get(shape, krylovdim + 1)
    be.axpby(1.0 / be.norm(x0), x0, 0.0, V[0])
    Hm = np.zeros((krylovdim + 1, krylovdim))
    k = 0
    vals, conv = np.zeros(0, dtype=complex), 0
    while k < krylovdim:
        matvec(V[k], V[k + 1])
        h, beta = be.orth_step(V[:k + 1], V[k + 1])
        Hm[:k + 1, k] = h
        Hm[k + 1, k] = beta
        k += 1
        if k >= num and (k % 5 == 0 or k == krylovdim or beta < 1e-300):
            ev, S = np.linalg.eig(Hm[:k, :k])
            order = np.argsort(-np.abs(ev))
            vals = ev[order]
            res = np.abs(beta * S[-1, order])
            conv = int(np.sum(res[:num] < tol))
            if conv >= min(num, k) or beta < 1e-300:
                break
    return vals[:num], conv
