"""FiniteMPS / InfiniteMPS with device-resident tensors.

FiniteMPS restates the reference's lazy-gauge state machine (src/states/finitemps.jl:53-169,
orthoview.jl:1-143): ALs / ARs / ACs / CLs entries are device tensors or None, gauge movement
happens on read through QRpos / LQpos (libmpsk: mpsk_qrpos / mpsk_lqpos) and writes to AC
invalidate exactly the entries the reference invalidates, so the FinEnv identity checks
(FinEnv.jl:115,132) trigger the same recomputations.
"""
from __future__ import annotations

import numpy as np

from .backend import Backend, DTensor, default_backend
from . import krylov


# ---- gauge primitives on device tensors -------------------------------------------------------

def _structured(be, cplx):
    """embedded complex tensors on a device backend: the gauge steps must return embeddings (cplx.qrpos_structured)."""
    return cplx and hasattr(be, "times_i") and hasattr(be, "triu_")


def _qr(be, M, cplx=False):
    if _structured(be, cplx):
        from .cplx import qrpos_structured
        return qrpos_structured(be, M)
    return be.qrpos(M)


def _lq(be, M, cplx=False):
    if _structured(be, cplx):
        from .cplx import lqpos_structured
        return lqpos_structured(be, M)
    return be.lqpos(M)


def leftorth(be: Backend, A: DTensor, cplx=False):
    """A[a,s,b] -> AL[a,s,k], C[k,b], diag(C) > 0   (leftorth(A; alg = QRpos()), orthoview.jl:56)."""
    Dl, d, Dr = A.shape
    if Dl * d < Dr:
        raise ValueError(f"leftorth: tensor {A.shape} is not full column rank shaped (Dl*d < Dr)")
    if _structured(be, cplx):
        from .cplx import qrpos_structured
        Q, R = qrpos_structured(be, A.reshape(Dl * d, Dr))
    else:
        Q, R = be.qrpos(A.reshape(Dl * d, Dr))
    return Q.reshape(Dl, d, Dr), R


def rightorth(be: Backend, A: DTensor, cplx=False):
    """A[a,s,b] -> C[a,k], AR[k,s,b], diag(C) > 0   (rightorth(_transpose_tail(A); alg = LQpos()),
    orthoview.jl:52-54)."""
    Dl, d, Dr = A.shape
    if Dl > d * Dr:
        raise ValueError(f"rightorth: tensor {A.shape} is not full row rank shaped (Dl > d*Dr)")
    if _structured(be, cplx):
        # the (b, s) tail of A[a, (s, b)] carries the embedding on b (slowest): as a matrix Dl x (d Dr) the 2x2 blocks
        # sit on (row pair, column pair) with columns (s, 2b + beta) -> beta is NOT the fastest column index; go through
        # the [a; (b, s)]-ordered copy, whose column pairs are adjacent
        from .cplx import lqpos_structured
        At = be.empty(Dl, Dr, d)                              # At[a, b, s] = A[a, s, b]
        for s_ in range(d):
            be.copy2d(Dl, Dr, A.ptr + 8 * s_ * Dl, Dl * d, At.ptr + 8 * s_ * Dl * Dr, Dl)
        # columns of At as a matrix: (b, s) with b fastest -> pairs (2b, 2b + 1) adjacent
        L, Qt = lqpos_structured(be, At.reshape(Dl, Dr * d))
        Q = be.empty(Dl, d, Dr)
        for s_ in range(d):
            be.copy2d(Dl, Dr, Qt.ptr + 8 * s_ * Dl * Dr, Dl, Q.ptr + 8 * s_ * Dl, Dl * d)
        return L, Q
    L, Q = be.lqpos(A.reshape(Dl, d * Dr))
    return L, Q.reshape(Dl, d, Dr)


def mul_AC(be: Backend, AL: DTensor, C: DTensor):
    """AC = AL * C   (orthoview.jl:103)."""
    Dl, d, Dr = AL.shape
    out = be.gemm(AL.reshape(Dl * d, Dr), C)
    return out.reshape(Dl, d, C.shape[1])


def mul_CA(be: Backend, C: DTensor, AR: DTensor):
    """AC = C * AR   (orthoview.jl:99)."""
    Dl, d, Dr = AR.shape
    out = be.gemm(C, AR.reshape(Dl, d * Dr))
    return out.reshape(C.shape[0], d, Dr)


class FiniteMPS:
    def __init__(self, As, normalize=False, be: Backend | None = None, cplx: bool | None = None):
        """finitemps.jl:143-169: left-to-right QRpos sweep; only CLs[end] is set.
        As: list of host arrays (Dl, d, Dr) or DTensors.  cplx=True: the DTensors given are EMBEDDED complex tensors
        (cplx.py) -- needed when a complex state is rebuilt from device tensors (excitations.py)."""
        self.be = default_backend() if be is None else be
        be = self.be
        # complex128 input: carried as the real 2x2 embedding on the bond indices (cplx.py); the embedded
        # Frobenius norm is sqrt(2) times the complex one
        self.cplx = any((not isinstance(a, DTensor)) and np.iscomplexobj(a) for a in As) or bool(cplx)
        if self.cplx:
            from .cplx import embed
            As = [a if isinstance(a, DTensor) else embed(np.asarray(a)) for a in As]
        nrm_target = np.sqrt(2.0) if self.cplx else 1.0
        As = [a if isinstance(a, DTensor) else be.upload(np.asarray(a)) for a in As]
        N = len(As)
        C = None
        for i in range(N):
            if C is not None:
                As[i] = mul_CA(be, C, As[i])
            As[i], C = leftorth(be, As[i], self.cplx)
            if normalize:
                be.scal(nrm_target / be.norm(C), C)
        self.N = N
        self.ALs = list(As)
        self.ARs = [None] * N
        self.ACs = [None] * N
        self.CLs = [None] * (N + 1)
        self.CLs[N] = C

    @classmethod
    def random(cls, L, d, D, rng, normalize=True, be=None, dtype=None):
        """FiniteMPS(rand, elt, L, P, maxV): bond dims min(d^i, D, d^(L-i)) (finitemps.jl:171-207),
        entries uniform[0,1)."""
        dims = [1]
        for _ in range(1, L):
            dims.append(min(dims[-1] * d, D))
        dims.append(1)
        for k in range(L - 1, 0, -1):
            dims[k] = min(dims[k], dims[k + 1] * d)
        if dtype is not None and np.issubdtype(dtype, np.complexfloating):
            return cls([rng.random((dims[i], d, dims[i + 1])) + 1j * rng.random((dims[i], d, dims[i + 1]))
                        for i in range(L)], normalize=normalize, be=be)
        return cls([rng.random((dims[i], d, dims[i + 1])) for i in range(L)], normalize=normalize, be=be)

    def copy(self):
        o = object.__new__(FiniteMPS)
        o.be, o.N, o.cplx = self.be, self.N, self.cplx
        o.ALs, o.ARs, o.ACs, o.CLs = list(self.ALs), list(self.ARs), list(self.ACs), list(self.CLs)
        return o

    def __len__(self):
        return self.N

    # ---- views (0-based; CR(i), i in -1..N-1, is the bond to the right of site i) ----
    def AL(self, i):  # orthoview.jl:6-9
        if self.ALs[i] is None:
            self.CR(i)
        return self.ALs[i]

    def AR(self, i):  # :27-31
        if self.ARs[i] is None:
            self.CR(i - 1)
        return self.ARs[i]

    def CR(self, i):  # :49-60
        if self.CLs[i + 1] is None:
            if i == -1 or self.ALs[i] is not None:
                C, ar = rightorth(self.be, self.AC(i + 1), self.cplx)
                self.CLs[i + 1], self.ARs[i + 1] = C, ar
            else:
                al, C = leftorth(self.be, self.AC(i), self.cplx)
                self.ALs[i], self.CLs[i + 1] = al, C
        return self.CLs[i + 1]

    def AC(self, i):  # :95-106
        if self.ACs[i] is None and self.ARs[i] is not None:
            self.ACs[i] = mul_CA(self.be, self.CR(i - 1), self.ARs[i])
        elif self.ACs[i] is None and self.ALs[i] is not None:
            self.ACs[i] = mul_AC(self.be, self.ALs[i], self.CR(i))
        return self.ACs[i]

    def set_AC(self, i, vec):  # :108-143
        if self.ACs[i] is None:
            if i < self.N - 1:
                self.AR(i + 1)
            if i > 0:
                self.AL(i - 1)
        self.ACs = [None] * self.N
        self.CLs = [None] * (self.N + 1)
        for k in range(i, self.N):
            self.ALs[k] = None
        for k in range(0, i + 1):
            self.ARs[k] = None
        if isinstance(vec, tuple):
            a, b = vec
            if len(a.shape) == 2:      # (c, ar)
                self.CLs[i], self.ARs[i] = a, b
            else:                      # (al, c)
                self.CLs[i + 1], self.ALs[i] = b, a
        else:
            self.ACs[i] = vec

    def set_CR(self, i, vec):  # CRView.setindex!  orthoview.jl:62-78
        if self.CLs[i + 1] is None:
            if self.ALs[i] is not None:
                C, ar = rightorth(self.be, self.AC(i + 1), self.cplx)
                self.CLs[i + 1], self.ARs[i + 1] = C, ar
            else:
                al, C = leftorth(self.be, self.AC(i), self.cplx)
                self.ALs[i], self.CLs[i + 1] = al, C
        self.ACs = [None] * self.N
        self.CLs = [None] * (self.N + 1)
        for k in range(i + 1, self.N):
            self.ALs[k] = None
        for k in range(0, i + 1):
            self.ARs[k] = None
        self.CLs[i + 1] = vec

    def set_AC_with_leftorth(self, i, vec):
        """Right-moving site update: the reference computes leftorth(old AC[i]) for calc_galerkin
        (toolbox.jl:20, via AL[i] -> CRView, orthoview.jl:56) and leftorth(new AC[i]) when AL[i] is next
        needed (FinEnv.jl:136).  Both inputs are known once the eigensolver returns, so the two QRpos
        factorizations are issued TOGETHER (mpsk_qrpos2) and the state is left exactly as the lazy views
        would leave it.  Returns the OLD AL[i] (the galerkin projector)."""
        be = self.be
        if self.ALs[i] is not None or self.ACs[i] is None or not hasattr(be, "qrpos2") or _structured(be, self.cplx):
            al_old = self.AL(i)
            self.set_AC(i, vec)
            return al_old
        old = self.ACs[i]
        Dl, d, Dr = old.shape
        if vec.shape != old.shape or Dl * d < Dr:
            al_old = self.AL(i)
            self.set_AC(i, vec)
            return al_old
        Q1, R1, Q2, R2 = be.qrpos2(old.reshape(Dl * d, Dr), vec.reshape(Dl * d, Dr))
        al_old = Q1.reshape(Dl, d, Dr)
        self.ALs[i], self.CLs[i + 1] = al_old, R1          # what CR(i) would have cached
        self.set_AC(i, vec)                                 # invalidates, stores ACs[i] = vec
        self.ALs[i], self.CLs[i + 1] = Q2.reshape(Dl, d, Dr), R2   # what the next AL(i) / CR(i) computes
        return al_old

    def norm(self):  # finitemps.jl:467
        n = self.be.norm(self.AC(0))
        return n / np.sqrt(2.0) if self.cplx else n

    def download(self, t):
        """host copy of one of this state's tensors (complex if the state is complex: un-embedded)."""
        h = self.be.download(t)
        if self.cplx:
            from .cplx import extract
            return extract(h)
        return h

    def bond_dims(self):
        out = []
        for i in range(self.N):
            t = self.ALs[i] or self.ARs[i] or self.ACs[i]
            out.append(t.shape[2] // 2 if self.cplx else t.shape[2])
        return out

    def to_host(self):
        """left-canonical host tensors [AL_1 .. AL_{N-1}, AC_N] (for checks)."""
        return [self.be.download(self.AL(i)) for i in range(self.N - 1)] + [self.be.download(self.AC(self.N - 1))]


# ---- uniform gauge (src/states/ortho.jl) -------------------------------------------------------

def _transfer_left_bond(be, v, A, Ab, out=None):
    D2, D1 = v.shape
    r = be.transfer_left(None, v.reshape(1, D2, D1), A, Ab, out=None if out is None else out.reshape(1, *out.shape))
    return r.reshape(r.shape[1], r.shape[2])


def _transfer_right_bond(be, v, A, Ab, out=None):
    r = be.transfer_right(None, v.reshape(1, *v.shape), A, Ab, out=None if out is None else out.reshape(1, *out.shape))
    return r.reshape(r.shape[1], r.shape[2])


def uniform_leftorth(be, A, C0, tol=1e-14, maxiter=100, eig_miniter=10, cplx=False):
    """ortho.jl uniform_leftorth!: iterate {optional Arnoldi on flip(TransferMatrix(A, AL));
    per site C.A -> QRpos} until ||C0 - C1|| < tol.  Returns (AL list, CR list)."""
    n = len(A)
    CR = [None] * n
    c = be.copy(C0)
    be.scal(1.0 / be.norm(c), c)
    CR[n - 1] = c
    AL = [None] * n
    eps, it = np.inf, 0
    while True:
        if it >= eig_miniter:
            etol = max(eps ** 2, 1e-15)

            def tm(v, out):
                cur = v
                for i in range(n):
                    cur = _transfer_left_bond(be, cur, A[i], AL[i])
                be.axpby(1.0, cur, 0.0, out)
                return out
            _, vec = krylov.eigsolve_lm_real(be, tm, CR[n - 1], tol=etol)
            _, CR[n - 1] = _qr(be, vec, cplx)
        C0_ = CR[n - 1]
        for i in range(n):
            AL[i], CR[i] = leftorth(be, mul_CA(be, CR[(i - 1) % n], A[i]), cplx)
        be.scal(1.0 / be.norm(CR[n - 1]), CR[n - 1])
        diff = be.copy(C0_)
        be.axpby(-1.0, CR[n - 1], 1.0, diff)
        eps = be.norm(diff)
        it += 1
        if eps < tol or it > maxiter:
            return AL, CR


def uniform_rightorth(be, A, C0, tol=1e-14, maxiter=100, eig_miniter=10, cplx=False):
    n = len(A)
    CR = [None] * n
    c = be.copy(C0)
    be.scal(1.0 / be.norm(c), c)
    CR[n - 1] = c
    AR = [None] * n
    eps, it = np.inf, 0
    while True:
        if it >= eig_miniter:
            etol = max(eps ** 2, 1e-15)

            def tm(v, out):
                cur = v
                for i in range(n - 1, -1, -1):
                    cur = _transfer_right_bond(be, cur, A[i], AR[i])
                be.axpby(1.0, cur, 0.0, out)
                return out
            _, vec = krylov.eigsolve_lm_real(be, tm, CR[n - 1], tol=etol)
            CR[n - 1], _ = _lq(be, vec, cplx)
        C0_ = CR[n - 1]
        for i in range(n - 1, -1, -1):
            CR[(i - 1) % n], AR[i] = rightorth(be, mul_AC(be, A[i], CR[i]), cplx)
        be.scal(1.0 / be.norm(CR[n - 1]), CR[n - 1])
        diff = be.copy(C0_)
        be.axpby(-1.0, CR[n - 1], 1.0, diff)
        eps = be.norm(diff)
        it += 1
        if eps < tol or it > maxiter:
            return AR, CR


class InfiniteMPS:
    """src/states/infinitemps.jl:46-104 : fields AL, AR, CR (bond right of site i), AC (device)."""

    def __init__(self, AL, AR, CR, AC, be):
        self.AL, self.AR, self.CR, self.AC, self.be = AL, AR, CR, AC, be

    def __len__(self):
        return len(self.AL)

    @classmethod
    def from_tensors(cls, A, tol=1e-14, maxiter=100, be=None):
        """infinitemps.jl:139-170 (gaugefix! order = :LR)."""
        be = default_backend() if be is None else be
        cx = any((not isinstance(a, DTensor)) and np.iscomplexobj(a) for a in A)
        if cx:                                 # complex128 via the bond embedding (cplx.py)
            from .cplx import embed
            A = [embed(np.asarray(a)) for a in A]
        A = [a if isinstance(a, DTensor) else be.upload(np.asarray(a)) for a in A]
        D = A[0].shape[0]
        AL, CR = uniform_leftorth(be, A, be.upload(np.eye(D)), tol, maxiter, cplx=cx)
        AR, CR = uniform_rightorth(be, AL, CR[-1], tol, maxiter, cplx=cx)
        AC = [mul_AC(be, AL[i], CR[i]) for i in range(len(A))]
        out = cls(AL, AR, CR, AC, be)
        out.cplx = cx
        return out

    @classmethod
    def from_AL(cls, AL, C0, tol=1e-14, maxiter=100, be=None, cplx=False):
        """infinitemps.jl:172-186 (gaugefix! order = :R)."""
        be = default_backend() if be is None else be
        AR, CR = uniform_rightorth(be, AL, C0, tol, maxiter, cplx=cplx)
        AC = [mul_AC(be, AL[i], CR[i]) for i in range(len(AL))]
        out = cls(list(AL), AR, CR, AC, be)
        out.cplx = cplx
        return out

    @classmethod
    def random(cls, d, D, rng, n=1, be=None):
        return cls.from_tensors([rng.random((D, d, D)) for _ in range(n)], be=be)
