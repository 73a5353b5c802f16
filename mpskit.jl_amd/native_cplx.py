"""complex128 finite MPS on INTERLEAVED storage: states, environments and the one-site drivers entirely on the MPSK_C128 entry
points of the C ABI (include/mpsk.h) -- mpsk_hac_* / mpsk_dC / mpsk_transfer_* (native complex kernels) and, since round 3,
mpsk_qrpos / mpsk_lqpos / mpsk_gemm under mpsk_ctx_set_dtype(MPSK_C128).  A site tensor A[Dl, d, Dr] of complex numbers is ONE
device buffer of 2 Dl d Dr doubles in Julia's Array{ComplexF64} layout (shape (2 Dl, d, Dr) as a real tensor): 2x the memory of
a real state, where the bond-embedded states of cplx.py take 4x and run their transfers at 8x the real flops.

This module is the proof that a complex host needs nothing but the C ABI (it is what a Julia `ROCTensor{ComplexF64}` backend
would do): the state is kept in explicit mixed-canonical form (sites left of the centre left-orthonormal,
right of it right-orthonormal), which is all the one-site algorithms need --
    find_groundstate (DMRG, dmrg.jl:22-55)   ->  NativeFiniteMPS + dmrg_sweep / dmrg
    find_groundstate (DMRG2, dmrg.jl:80-137) ->  dmrg2_sweep   (mpsk_dAC2 complex, mpsk_tsplit under MPSK_C128)
    timestep (TDVP, tdvp.jl:61-94)           ->  tdvp_step
    timestep (TDVP2, tdvp.jl:113-146)        ->  tdvp2_step
The lazy-gauge FiniteMPS of states.py (all drivers, two-site algorithms, infinite systems, excitations) stays on the embedded
representation; the Krylov solvers are shared (real inner products on the 2n doubles of an interleaved vector are all a
Hermitian Lanczos / Arnoldi iteration needs, and multiplication by i is mpsk_vtimes_i)."""
from __future__ import annotations

import numpy as np

from .backend import Backend, DTensor
from . import krylov
from .cplx import HalfEmbeddedOp


def _mat(t: DTensor, rows2, cols):
    return t.reshape(rows2, cols)


class NativeFiniteMPS:
    """finite MPS with interleaved complex128 site tensors in mixed-canonical form around `center` (which holds AC)."""

    def __init__(self, tensors, be: Backend, normalize=True):
        self.be = be
        self.A = [be.upload_c(np.asarray(t, dtype=np.complex128)) for t in tensors]     # (2 Dl, d, Dr) each
        self.N = len(self.A)
        self.center = self.N - 1
        for i in range(self.N - 1, 0, -1):                     # right-canonicalise: A_i = L Q, A_{i-1} <- A_{i-1} L
            self._shift_left(i)
        if normalize:
            be.scal(1.0 / be.norm(self.A[0]), self.A[0])

    def __len__(self):
        return self.N

    def dims(self, i):
        Dl2, d, Dr = self.A[i].shape
        return Dl2 // 2, d, Dr

    def _shift_right(self, i):
        """centre i -> i + 1:  AC_i = AL C (QRpos),  AC_{i+1} = C AR_{i+1}   (orthoview.jl:49-60, :99)"""
        be = self.be
        Dl, d, Dr = self.dims(i)
        Q, R = be.qrpos_c(_mat(self.A[i], 2 * Dl * d, Dr))
        k = R.shape[1]
        self.A[i] = Q.reshape(2 * Dl, d, k)
        Dn, dn, Drn = self.dims(i + 1)
        nxt = be.gemm_c(R, _mat(self.A[i + 1], 2 * Dn, dn * Drn))
        self.A[i + 1] = nxt.reshape(2 * k, dn, Drn)
        self.center = i + 1
        return R

    def _shift_left(self, i):
        """centre i -> i - 1:  AC_i = C AR (LQpos),  AC_{i-1} = AL_{i-1} C   (orthoview.jl:61-72, :103)"""
        be = self.be
        Dl, d, Dr = self.dims(i)
        Lm, Q = be.lqpos_c(_mat(self.A[i], 2 * Dl, d * Dr))
        k = Lm.shape[1]
        self.A[i] = Q.reshape(2 * k, d, Dr)
        Dp, dp, Drp = self.dims(i - 1)
        prv = be.gemm_c(_mat(self.A[i - 1], 2 * Dp * dp, Drp), Lm)
        self.A[i - 1] = prv.reshape(2 * Dp, dp, k)
        self.center = i - 1
        return Lm

    def move_center(self, pos):
        while self.center < pos:
            self._shift_right(self.center)
        while self.center > pos:
            self._shift_left(self.center)

    def norm(self):
        return self.be.norm(self.A[self.center])

    def to_host(self):
        """complex host tensors (mixed-canonical around the current centre)."""
        return [self.be.download_c(t) for t in self.A]

    def bytes(self):
        return 8 * sum(t.size for t in self.A)

    def copy(self):
        out = object.__new__(NativeFiniteMPS)
        out.be, out.N, out.center = self.be, self.N, self.center
        out.A = [self.be.copy(t) for t in self.A]
        return out


class ComplexMPOHamiltonian:
    """MPOHamiltonian with ComplexF64 entries (mpohamiltonian.jl:8-31): a periodic list of MPSK_C128 slices.  `data[site]` =
    {(i, j): scalar | d x d | [chi_i, d, d, chi_j]} with complex values, levels 0 .. odim - 1, H[0, 0] = H[odim-1, odim-1] = 1.
    (The real MPOHamiltonian of operators.py is accepted by the drivers as well: its slices get complex twins.)"""

    def __init__(self, data, be: Backend, d=None):
        if isinstance(data, dict):
            data = [data]
        self.be, self.period = be, len(data)
        self.odim = 1 + max(max(i, j) for blk in data for (i, j) in blk)
        for blk in data:
            for v in blk.values():
                if d is None and not np.isscalar(v):
                    a = np.asarray(v)
                    d = a.shape[0] if a.ndim == 2 else a.shape[1]
        self.d = int(d)
        chis = [[1] * self.odim for _ in range(self.period + 1)]
        for s_, blk in enumerate(data):
            for (i, j), v in blk.items():
                if not np.isscalar(v) and np.asarray(v).ndim == 4:
                    chis[s_][i], chis[s_ + 1][j] = np.asarray(v).shape[0], np.asarray(v).shape[3]
        for i in range(self.odim):
            chis[0][i] = chis[self.period][i] = max(chis[0][i], chis[self.period][i])
        self.chis, self.slices = chis, []
        for s_, blk in enumerate(data):
            blocks = {}
            for (i, j), v in blk.items():
                if np.isscalar(v):
                    if v != 0:
                        blocks[(i, j)] = complex(v)
                    continue
                a = np.asarray(v, dtype=np.complex128)
                blocks[(i, j)] = a[None, :, :, None] if a.ndim == 2 else a
            self.slices.append(be.mposlice(self.odim, self.d, chis[s_], chis[s_ + 1], blocks, cplx=True))

    def __getitem__(self, i):
        return self.slices[i % self.period]

    def __len__(self):
        return self.period


def _boundary(be, chis, D, active):
    """FinEnv.jl:49-67: identity on the active level, zeros elsewhere -- complex interleaved slabs (W, 2 D, D)."""
    blocks = []
    for i, chi in enumerate(chis):
        b = np.zeros((D, chi, D), dtype=np.complex128)
        if i == active:
            for k in range(chi):
                b[:, k, :] = np.eye(D)
        blocks.append(b)
    return be.upload_env_c(blocks)


class NativeFinEnv:
    """left / right environments of a NativeFiniteMPS (FinEnv.jl:9-145), interleaved complex slabs (W, 2 Dbra, Dket); valid
    for sites strictly left / right of the centre and refreshed as the centre moves (the drivers below move it one site at
    a time and call `extend_left` / `extend_right`)."""

    def __init__(self, psi: NativeFiniteMPS, H):
        be = self.be = psi.be
        L = self.L = len(psi)
        if isinstance(H, ComplexMPOHamiltonian):
            self.opp = [H[i] for i in range(L)]
            chil, chir = H.chis[0], H.chis[(L - 1) % H.period + 1]
        else:                                                   # real MPOHamiltonian: complex twins of its slices
            self.opp = [HalfEmbeddedOp._cslice(be, H[i]) for i in range(L)]
            chil, chir = H[0].chil, H[L - 1].chir
        odim = H.odim
        self.GL = [_boundary(be, chil, psi.dims(0)[0], 0)] + [None] * L
        self.GR = [None] * L + [_boundary(be, chir, psi.dims(L - 1)[2], odim - 1)]
        self.n_transfers = 0
        c = psi.center
        for j in range(L - 1, c, -1):
            self.extend_right(psi, j)
        for j in range(0, c):
            self.extend_left(psi, j)

    def extend_left(self, psi, j):
        """GL[j + 1] from GL[j] and the left-orthonormal A[j]  (FinEnv.jl:131-145, transfer.jl:105-110)"""
        self.GL[j + 1] = self.be.transfer_left(self.opp[j], self.GL[j], psi.A[j], psi.A[j])
        self.n_transfers += 1

    def extend_right(self, psi, j):
        """GR[j] from GR[j + 1] and the right-orthonormal A[j]  (FinEnv.jl:114-129)"""
        self.GR[j] = self.be.transfer_right(self.opp[j], self.GR[j + 1], psi.A[j], psi.A[j])
        self.n_transfers += 1

    def bytes(self):
        return 8 * sum(t.size for t in self.GL + self.GR if t is not None)


class _HAC:
    """H_AC of the centre site on interleaved vectors (derivatives.jl:77-93): prepared operator, one call per application."""

    def __init__(self, be, envs, pos):
        self.h = be.hac_create(envs.opp[pos], envs.GL[pos], envs.GR[pos + 1])

    def __call__(self, x, out=None):
        return self.h.apply(x, out=out)


class _HC:
    """H_C of the bond right of site pos (derivatives.jl:3-31)."""

    def __init__(self, be, envs, pos):
        self.be, self.GL, self.GR = be, envs.GL[pos + 1], envs.GR[pos + 1]

    def __call__(self, x, out=None):
        return self.be.dC(self.GL, self.GR, x, out=out, cplx=True)


def energy(psi: NativeFiniteMPS, envs: NativeFinEnv):
    """<psi|H|psi> / <psi|psi> at the current centre (Re <AC, H_AC AC>; the imaginary part vanishes for Hermitian H)."""
    be = psi.be
    ac = psi.A[psi.center]
    return be.dot(ac, _HAC(be, envs, psi.center)(ac)) / be.dot(ac, ac)


def dmrg_sweep(psi: NativeFiniteMPS, H, envs: NativeFinEnv, eigalg, ws=None):
    """One DMRG sweep, pos in [1:L-1; L:-1:2] (dmrg.jl:33-38): eigsolve with H_AC at the centre, QRpos / LQpos to the next
    site, one environment transfer.  Starts and ends with the centre at site 0.  Returns the energy after the sweep."""
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be) if ws is None else ws
    psi.move_center(0)

    def solve(pos):
        h = _HAC(be, envs, pos)
        _, vec, _, _ = krylov.eigsolve_sr(be, h, psi.A[pos], tol=eigalg.tol, krylovdim=eigalg.krylovdim,
                                          maxiter=eigalg.maxiter, fixed_matvecs=eigalg.fixed_matvecs, ws=ws)
        psi.A[pos] = vec

    for pos in range(0, L - 1):
        solve(pos)
        psi._shift_right(pos)
        envs.extend_left(psi, pos)
    for pos in range(L - 1, 0, -1):
        solve(pos)
        psi._shift_left(pos)
        envs.extend_right(psi, pos)
    return energy(psi, envs)


def dmrg(psi: NativeFiniteMPS, H, eigalg, maxiter=10, tol=1e-10, envs=None, verbose=False):
    """find_groundstate(psi, H, DMRG(...)) on interleaved complex storage: sweeps until the energy change is below tol."""
    envs = NativeFinEnv(psi, H) if envs is None else envs
    ws = krylov.KrylovWorkspace(psi.be)
    E_old, log = np.inf, []
    for it in range(1, maxiter + 1):
        E = dmrg_sweep(psi, H, envs, eigalg, ws)
        log.append((it, E))
        if verbose:
            print(f"DMRG (native complex) {it:3d}: obj = {E:+.12e}")
        if abs(E - E_old) <= tol * max(1.0, abs(E)):
            break
        E_old = E
    return psi, envs, log


def tdvp_step(psi: NativeFiniteMPS, H, envs: NativeFinEnv, t, dt, alg, ws=None):
    """timestep!(psi, H, t, dt, TDVP())  (tdvp.jl:61-94): forward integration of AC by dt / 2, backward integration of the bond
    matrix, left to right and back; any complex dt (real time: exp(-i dt H))."""
    from .algorithms import _integrate_embedded
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be) if ws is None else ws
    psi.move_center(0)
    fwd, bwd = -1j * complex(dt) / 2, 1j * complex(dt) / 2
    for i in range(L - 1):
        psi.A[i] = _integrate_embedded(be, _HAC(be, envs, i), psi.A[i], fwd, alg, ws)
        Dl, d, Dr = psi.dims(i)
        Q, R = be.qrpos_c(_mat(psi.A[i], 2 * Dl * d, Dr))
        psi.A[i] = Q.reshape(2 * Dl, d, R.shape[1])
        envs.extend_left(psi, i)
        C = _integrate_embedded(be, _HC(be, envs, i), R, bwd, alg, ws)
        Dn, dn, Drn = psi.dims(i + 1)
        psi.A[i + 1] = be.gemm_c(C, _mat(psi.A[i + 1], 2 * Dn, dn * Drn)).reshape(2 * C.shape[0] // 2, dn, Drn)
        psi.center = i + 1
    psi.A[L - 1] = _integrate_embedded(be, _HAC(be, envs, L - 1), psi.A[L - 1], fwd, alg, ws)
    for i in range(L - 1, 0, -1):
        psi.A[i] = _integrate_embedded(be, _HAC(be, envs, i), psi.A[i], fwd, alg, ws)
        Dl, d, Dr = psi.dims(i)
        Lm, Q = be.lqpos_c(_mat(psi.A[i], 2 * Dl, d * Dr))
        psi.A[i] = Q.reshape(2 * Lm.shape[1], d, Dr)
        envs.extend_right(psi, i)
        # the bond matrix between i - 1 and i: H_C with GL[i] and GR[i]
        hc = _HC.__new__(_HC)
        hc.be, hc.GL, hc.GR = be, envs.GL[i], envs.GR[i]
        C = _integrate_embedded(be, hc, Lm, bwd, alg, ws)
        Dp, dp, Drp = psi.dims(i - 1)
        psi.A[i - 1] = be.gemm_c(_mat(psi.A[i - 1], 2 * Dp * dp, Drp), C).reshape(2 * Dp, dp, C.shape[1])
        psi.center = i - 1
    psi.A[0] = _integrate_embedded(be, _HAC(be, envs, 0), psi.A[0], fwd, alg, ws)
    return psi, envs


def _two_site(be, left: DTensor, right: DTensor):
    """theta[a, s1, b, s2] = sum_m left[a, s1, m] right[m, s2, b] on interleaved tensors (dmrg.jl:92 / :108): d2 complex GEMMs
    on strided views of `right` (mpsk_gemm under MPSK_C128; leading dimensions and offsets in complex elements)."""
    Dl, d1, Dm = left.shape[0] // 2, left.shape[1], left.shape[2]
    _, d2, Dr = right.shape
    theta = be.empty(2 * Dl, d1, Dr, d2)
    be._set_dtype(True)
    try:
        for s2 in range(d2):
            be.gemm_raw(False, False, Dl * d1, Dr, Dm, 1.0, left.ptr, Dl * d1, right.ptr + 16 * s2 * Dm, Dm * d2, 0.0,
                        theta.ptr + 16 * s2 * Dl * d1 * Dr, Dl * d1)
    finally:
        be._set_dtype(False)
    return theta


def dmrg2_sweep(psi: NativeFiniteMPS, H, envs: NativeFinEnv, eigalg, trunc_dim, ws=None):
    """One two-site DMRG sweep (dmrg.jl:86-120) on interleaved storage: theta = AC AR, eigsolve with H_AC2 (mpsk_dAC2 complex),
    al, c, ar = tsvd!(theta; trunc = truncdim(D)) through mpsk_tsplit under MPSK_C128, normalize!(c).  Centre at site 0 before
    and after.  Returns the energy after the sweep."""
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be) if ws is None else ws
    psi.move_center(0)

    def update(pos, theta):
        h1, h2, GL, GR = envs.opp[pos], envs.opp[pos + 1], envs.GL[pos], envs.GR[pos + 2]
        op = lambda x, out=None: be.dAC2(h1, h2, GL, GR, x, out=out)
        _, new, _, _ = krylov.eigsolve_sr(be, op, theta, tol=eigalg.tol, krylovdim=eigalg.krylovdim, maxiter=eigalg.maxiter,
                                          fixed_matvecs=eigalg.fixed_matvecs, ws=ws)
        Dl2, d1, Dr, d2 = new.shape
        al, c, arm, _, _ = be.tsplit_c(new.reshape(Dl2 * d1, Dr * d2), max_keep=trunc_dim)
        k = c.shape[1]
        be.scal(1.0 / be.norm(c), c)                                     # normalize!(c)
        ar = be.empty(2 * k, d2, Dr)                                     # ar[k, s2, b] = arm[k, (b, s2)]
        for s2 in range(d2):
            be.copy2d(2 * k, Dr, arm.ptr + 8 * s2 * 2 * k * Dr, 2 * k, ar.ptr + 8 * s2 * 2 * k, 2 * k * d2)
        return al.reshape(Dl2, d1, k), c, ar

    for pos in range(0, L - 1):
        al, c, ar = update(pos, _two_site(be, psi.A[pos], psi.A[pos + 1]))
        k, d2, Dr = c.shape[1], ar.shape[1], ar.shape[2]
        psi.A[pos] = al
        psi.A[pos + 1] = be.gemm_c(c, ar.reshape(2 * k, d2 * Dr)).reshape(2 * k, d2, Dr)      # AC_{pos+1} = c ar
        psi.center = pos + 1
        envs.extend_left(psi, pos)
    for pos in range(L - 2, -1, -1):
        al, c, ar = update(pos, _two_site(be, psi.A[pos], psi.A[pos + 1]))
        Dl2, d1, k = al.shape
        psi.A[pos + 1] = ar
        psi.A[pos] = be.gemm_c(al.reshape(Dl2 * d1, k), c).reshape(Dl2, d1, k)                # AC_pos = al c
        psi.center = pos
        envs.extend_right(psi, pos + 1)
    return energy(psi, envs)


def tdvp2_step(psi: NativeFiniteMPS, H, envs: NativeFinEnv, t, dt, alg, trunc_dim, ws=None):
    """timestep!(psi, H, t, dt, TDVP2(truncdim(D)))  (tdvp.jl:113-146) on interleaved storage: two-site tensors integrated
    forward by dt / 2 and split (mpsk_tsplit under MPSK_C128; NO normalisation of c: the evolution is unitary), the new
    centre integrated backward, left to right and back."""
    from .algorithms import _integrate_embedded
    be, L = psi.be, len(psi)
    ws = krylov.KrylovWorkspace(be) if ws is None else ws
    psi.move_center(0)
    fwd, bwd = -1j * complex(dt) / 2, 1j * complex(dt) / 2

    def evolve_and_split(pos, theta):
        h1, h2, GL, GR = envs.opp[pos], envs.opp[pos + 1], envs.GL[pos], envs.GR[pos + 2]
        op = lambda x, out=None: be.dAC2(h1, h2, GL, GR, x, out=out)
        new = _integrate_embedded(be, op, theta, fwd, alg, ws)
        Dl2, d1, Dr, d2 = new.shape
        al, c, arm, _, _ = be.tsplit_c(new.reshape(Dl2 * d1, Dr * d2), max_keep=trunc_dim)
        k = c.shape[1]
        ar = be.empty(2 * k, d2, Dr)
        for s2 in range(d2):
            be.copy2d(2 * k, Dr, arm.ptr + 8 * s2 * 2 * k * Dr, 2 * k, ar.ptr + 8 * s2 * 2 * k, 2 * k * d2)
        return al.reshape(Dl2, d1, k), c, ar

    for i in range(L - 1):
        al, c, ar = evolve_and_split(i, _two_site(be, psi.A[i], psi.A[i + 1]))
        k, d2, Dr = c.shape[1], ar.shape[1], ar.shape[2]
        psi.A[i] = al
        psi.A[i + 1] = be.gemm_c(c, ar.reshape(2 * k, d2 * Dr)).reshape(2 * k, d2, Dr)
        psi.center = i + 1
        envs.extend_left(psi, i)
        if i != L - 2:
            psi.A[i + 1] = _integrate_embedded(be, _HAC(be, envs, i + 1), psi.A[i + 1], bwd, alg, ws)
    for i in range(L - 1, 0, -1):
        al, c, ar = evolve_and_split(i - 1, _two_site(be, psi.A[i - 1], psi.A[i]))
        Dl2, d1, k = al.shape
        psi.A[i] = ar
        psi.A[i - 1] = be.gemm_c(al.reshape(Dl2 * d1, k), c).reshape(Dl2, d1, k)
        psi.center = i - 1
        envs.extend_right(psi, i)
        if i != 1:
            psi.A[i - 1] = _integrate_embedded(be, _HAC(be, envs, i - 1), psi.A[i - 1], bwd, alg, ws)
    return psi, envs


# ---- the reference's entry points on interleaved states (same algorithm objects as algorithms.py) ---------------------------

def find_groundstate(psi: NativeFiniteMPS, H, alg, envs: NativeFinEnv = None):
    """find_groundstate(psi, H, DMRG(...) | DMRG2(...))  (dmrg.jl:22-55, :80-137): sweeps until the energy moves by less than
    alg.tol (relative) or alg.maxiter is reached.  Returns (psi, envs, |dE| of the last sweep)."""
    from .algorithms import DMRG, DMRG2
    envs = NativeFinEnv(psi, H) if envs is None else envs
    ws = krylov.KrylovWorkspace(psi.be)
    E_old, delta = np.inf, np.inf
    for it in range(1, alg.maxiter + 1):
        if isinstance(alg, DMRG2):
            if alg.trunc_dim <= 0:
                raise NotImplementedError("interleaved DMRG2 truncates by trunc_dim (mpsk_tsplit under MPSK_C128: truncdim scheme)")
            E = dmrg2_sweep(psi, H, envs, alg.eigalg, alg.trunc_dim, ws)
        elif isinstance(alg, DMRG):
            E = dmrg_sweep(psi, H, envs, alg.eigalg, ws)
        else:
            raise TypeError(f"find_groundstate on interleaved states takes DMRG or DMRG2, not {type(alg).__name__}")
        delta = abs(E - E_old)
        if alg.verbosity >= 3:
            print(f"[ Info: {type(alg).__name__} (interleaved complex) {it:3d}:\tobj = {E:+.12e}\tdE = {delta:.3e}")
        if delta <= alg.tol * max(1.0, abs(E)):
            break
        E_old = E
    return psi, envs, delta


def timestep(psi: NativeFiniteMPS, H, t, dt, alg, envs: NativeFinEnv = None):
    """timestep(psi, H, t, dt, TDVP() | TDVP2(...))  (tdvp.jl:61-94, :113-146).  Returns (psi, envs)."""
    from .algorithms import TDVP, TDVP2
    envs = NativeFinEnv(psi, H) if envs is None else envs
    if isinstance(alg, TDVP2):
        if alg.trunc_dim <= 0:
            raise NotImplementedError("interleaved TDVP2 truncates by trunc_dim (truncdim scheme)")
        return tdvp2_step(psi, H, envs, t, dt, alg, alg.trunc_dim)
    if isinstance(alg, TDVP):
        return tdvp_step(psi, H, envs, t, dt, alg)
    raise TypeError(f"timestep on interleaved states takes TDVP or TDVP2, not {type(alg).__name__}")
