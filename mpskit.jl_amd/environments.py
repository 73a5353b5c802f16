"""Environment caches on the device: FinEnv (src/environments/FinEnv.jl:9-145) and MPOHamInfEnv
(src/environments/mpohaminfenv.jl:4-215).  Each environment is ONE device tensor of W slabs
(include/mpsk.h layout); an update is one mpsk_transfer_left / mpsk_transfer_right call."""
from __future__ import annotations

import numpy as np

from .backend import Backend, DTensor
from . import krylov


def _start_env(be: Backend, chis, D, active_level):
    """FinEnv.jl:49-67: identity (x) ones(chi_i) on the active level, zeros elsewhere."""
    W = sum(chis)
    host = np.zeros((W, D, D))
    off = 0
    for i, chi in enumerate(chis):
        if i == active_level:
            for k in range(chi):
                host[off + k] = np.eye(D)
        off += chi
    t = be.upload(np.transpose(host, (1, 2, 0)))       # logical (D, D, W) -> column-major slabs
    return DTensor(t.buf, (W, D, D))


class FinEnv:
    """Cache of L+1 left / right environments with identity(`is`)-based invalidation
    (FinEnv.jl:114-145): an AL/AR that was invalidated and recomputed is a new object, so the
    environments that depend on it are rebuilt exactly when the reference rebuilds them."""

    def __init__(self, psi, H):
        be = self.be = psi.be
        L = len(psi)
        self.H = H
        self.opp = [H[i] for i in range(L)]
        odim = H.odim
        D0 = psi.AL(0).shape[0]
        t = psi.ARs[L - 1] if psi.ARs[L - 1] is not None else psi.AL(L - 1)
        DL = t.shape[2]
        self.leftenvs = [_start_env(be, self.opp[0].chil, D0, 0)] + [None] * L
        self.rightenvs = [None] * L + [_start_env(be, self.opp[L - 1].chir, DL, odim - 1)]
        self.ldeps = [None] * L
        self.rdeps = [None] * L
        self.n_transfers = 0

    def rightenv(self, ind, psi):  # FinEnv.jl:114-129
        L = len(psi)
        a = None
        for i in range(L - 1, ind, -1):
            if psi.AR(i) is not self.rdeps[i]:
                a = i
                break
        if a is not None:
            for j in range(a, ind, -1):
                ar = psi.AR(j)
                self.rightenvs[j] = self.be.transfer_right(self.opp[j], self.rightenvs[j + 1], ar, ar)
                self.rdeps[j] = ar
                self.n_transfers += 1
        return self.rightenvs[ind + 1]

    def leftenv(self, ind, psi):  # FinEnv.jl:131-145
        a = None
        for i in range(0, ind):
            if psi.AL(i) is not self.ldeps[i]:
                a = i
                break
        if a is not None:
            for j in range(a, ind):
                al = psi.AL(j)
                self.leftenvs[j + 1] = self.be.transfer_left(self.opp[j], self.leftenvs[j], al, al)
                self.ldeps[j] = al
                self.n_transfers += 1
        return self.leftenvs[ind]

    def poison(self, ind):  # FinEnv.jl:108-111
        self.ldeps[ind] = None
        self.rdeps[ind] = None


class MultipleEnvironments:
    """MultipleEnvironments (src/environments/multipleenv.jl:1-62): one environment object per LazySum term."""

    def __init__(self, H, envs):
        self.H, self.envs = H, list(envs)

    def recalculate(self, psi, tol=None):
        for e in self.envs:
            e.recalculate(psi, tol)
        return self


def environments(psi, H, **kw):
    """environments(psi, H)  (FinEnv.jl:41-70 / mpohaminfenv.jl:40-44 / multipleenv.jl:30-34)."""
    from .states import FiniteMPS
    from .operators import LazySum
    if isinstance(H, LazySum):
        return MultipleEnvironments(H, [environments(psi, h, **kw) for h in H])
    if isinstance(psi, FiniteMPS):
        return FinEnv(psi, H)
    return MPOHamInfEnv(psi, H, **kw)


class MPOHamInfEnv:
    """Infinite-MPS Hamiltonian environments: level-by-level triangular solve, GMRES on the
    regularised transfer matrix for identity levels (mpohaminfenv.jl:76-215).  Level i of site s is
    stored as its own (chi, D, D) device tensor; a site's full environment for the matvec kernels
    is assembled (device-to-device) on demand."""

    def __init__(self, psi, H, tol=1e-12, maxiter=100, rng=None):
        self.be = psi.be
        self.H, self.tol, self.maxiter = H, tol, maxiter
        n, odim = len(psi), H.odim
        be = self.be
        rng = np.random.default_rng(0) if rng is None else rng
        self.lw = [[be.upload(np.transpose(rng.random((H[s].chil[i], psi.AL[s].shape[0], psi.AL[s].shape[0])), (1, 2, 0)))
                    for s in range(n)] for i in range(odim)]
        self.rw = [[be.upload(np.transpose(rng.random((H[s].chir[i], psi.AR[s].shape[2], psi.AR[s].shape[2])), (1, 2, 0)))
                    for s in range(n)] for i in range(odim)]
        for i in range(odim):
            for s in range(n):
                self.lw[i][s] = DTensor(self.lw[i][s].buf, (H[s].chil[i],) + self.lw[i][s].shape[:2])
                self.rw[i][s] = DTensor(self.rw[i][s].buf, (H[s].chir[i],) + self.rw[i][s].shape[:2])
        self.ws = krylov.KrylovWorkspace(be)
        self._cache = {}
        self.recalculate(psi, tol)

    # ---- public ------------------------------------------------------------------------------
    def recalculate(self, psi, tol=None):
        tol = self.tol if tol is None else tol
        self._calclw(psi, tol)
        self._calcrw(psi, tol)
        self.dependency = psi
        self._cache = {}
        return self

    def _assemble(self, which, pos, n):
        key = (which, pos % n)
        if key not in self._cache:
            src = self.lw if which == "l" else self.rw
            parts = [src[i][pos % n] for i in range(self.H.odim)]
            W = sum(p.shape[0] for p in parts)
            out = self.be.empty(W, parts[0].shape[1], parts[0].shape[2])
            off = 0
            for p in parts:
                out.buf[off:off + p.size].copy_(p.buf[:p.size])
                off += p.size
            self._cache[key] = out
        return self._cache[key]

    def leftenv(self, pos, psi):
        if self.dependency is not psi:
            self.recalculate(psi)
        return self._assemble("l", pos, len(psi))

    def rightenv(self, pos, psi):
        if self.dependency is not psi:
            self.recalculate(psi)
        return self._assemble("r", pos, len(psi))

    # ---- helpers -------------------------------------------------------------------------------
    def _tl(self, v, O, A):
        """one (j -> idx) block of the left transfer: O scalar -> pass-through * O, dense -> MPO block."""
        be = self.be
        if np.isscalar(O):
            out = be.transfer_left(None, v, A, A)
            if O != 1:
                be.scal(O, out)
            return out
        blk = be.mposlice(1, A.shape[1], [O.shape[0]], [O.shape[3]], {(0, 0): O})
        return be.transfer_left(blk, v, A, A)

    def _tr(self, v, O, A):
        be = self.be
        if np.isscalar(O):
            out = be.transfer_right(None, v, A, A)
            if O != 1:
                be.scal(O, out)
            return out
        blk = be.mposlice(1, A.shape[1], [O.shape[0]], [O.shape[3]], {(0, 0): O})
        return be.transfer_right(blk, v, A, A)

    def _left_cycle(self, idx, psi):  # mpohaminfenv.jl:177-195
        n, H, be = len(psi), self.H, self.be
        for s in range(n):
            tgt = self.lw[idx][(s + 1) % n]
            acc = be.zeros(*tgt.shape)
            for j in range(idx, -1, -1):
                if not H[s].contains(j, idx):
                    continue
                t = self._tl(self.lw[j][s], H[s].blocks[(j, idx)], psi.AL[s])
                be.axpby(1.0, t, 1.0, acc)
            self.lw[idx][(s + 1) % n] = acc

    def _right_cycle(self, idx, psi):  # :197-215
        n, H, be = len(psi), self.H, self.be
        for s in range(n - 1, -1, -1):
            tgt = self.rw[idx][(s - 1) % n]
            acc = be.zeros(*tgt.shape)
            for j in range(idx, H.odim):
                if not H[s].contains(idx, j):
                    continue
                t = self._tr(self.rw[j][s], H[s].blocks[(idx, j)], psi.AR[s])
                be.axpby(1.0, t, 1.0, acc)
            self.rw[idx][(s - 1) % n] = acc

    def _CCd(self, psi, s):  # r_LL: C C^T of the bond right of site s
        c = psi.CR[s % len(psi)]
        return self.be.gemm(c, c, transB=True)

    def _CdC(self, psi, s):  # l_RR: C^T C of the bond left of site s
        c = psi.CR[(s - 1) % len(psi)]
        return self.be.gemm(c, c, transA=True)

    def _eye(self, D):
        return self.be.upload(np.eye(D))

    def _calclw(self, psi, tol):  # mpohaminfenv.jl:76-123
        n, H, odim, be = len(psi), self.H, self.H.odim, self.be
        D0 = psi.AL[0].shape[0]
        chi0 = H[0].chil[0]
        self.lw[0][0] = DTensor(be.upload(np.transpose(np.stack([np.eye(D0)] * chi0), (1, 2, 0))).buf, (chi0, D0, D0))
        if n > 1:
            self._left_cycle(0, psi)
        for i in range(1, odim):
            prev = self.lw[i][0]
            self.lw[i][0] = be.zeros(*prev.shape)
            self._left_cycle(i, psi)
            if H.isid(i):
                # regularised transfer matrix  x -> x - [x T - <r, x T> l]   (transfermatrix.jl:29-33,70-76)
                rfix = self._CCd(psi, n - 1)      # contracted with v:  sum_xy r[x,y] v[y,x]
                lfix = self._eye(D0)

                def op(x, out):
                    y = x
                    for s in range(n):
                        y = be.transfer_left(None, y, psi.AL[s], psi.AL[s])
                    be.regularize(y, rfix, lfix)
                    be.axpby(1.0, x, 0.0, out)
                    be.axpby(-1.0, y, 1.0, out)
                    return out
                self.lw[i][0] = krylov.gmres(be, op, self.lw[i][0], prev, tol=tol, maxiter=self.maxiter, ws=self.ws)
                if n > 1:
                    self._left_cycle(i, psi)
                for s in range(n):  # :103-107 subtract the fixed-point projection at every site
                    be.regularize(self.lw[i][s], self._CCd(psi, s - 1), self._eye(self.lw[i][s].shape[1]))
            else:
                if all(H[s].contains(i, i) for s in range(n)):
                    def op(x, out):
                        y = x
                        for s in range(n):
                            y = self._tl(y, H[s].blocks[(i, i)], psi.AL[s])
                        be.axpby(1.0, x, 0.0, out)
                        be.axpby(-1.0, y, 1.0, out)
                        return out
                    self.lw[i][0] = krylov.gmres(be, op, self.lw[i][0], prev, tol=tol, maxiter=self.maxiter, ws=self.ws)
                if n > 1:
                    self._left_cycle(i, psi)

    def _calcrw(self, psi, tol):  # :125-175
        n, H, odim, be = len(psi), self.H, self.H.odim, self.be
        DL = psi.AR[n - 1].shape[2]
        chiL = H[n - 1].chir[odim - 1]
        self.rw[odim - 1][n - 1] = DTensor(
            be.upload(np.transpose(np.stack([np.eye(DL)] * chiL), (1, 2, 0))).buf, (chiL, DL, DL))
        if n > 1:
            self._right_cycle(odim - 1, psi)
        for i in range(odim - 2, -1, -1):
            prev = self.rw[i][n - 1]
            self.rw[i][n - 1] = be.zeros(*prev.shape)
            self._right_cycle(i, psi)
            if H.isid(i):
                # <l_RR, x> with l = C^T C of the bond left of site 0:  sum_xy l[x,y] x[x,y]
                # mpsk_regularize contracts lvec[x,y] v[y,x] -> pass l^T (symmetric here)
                lfix = self._CdC(psi, 0)
                rfix = self._eye(DL)

                def op(x, out):
                    y = x
                    for s in range(n - 1, -1, -1):
                        y = be.transfer_right(None, y, psi.AR[s], psi.AR[s])
                    be.regularize(y, lfix, rfix)
                    be.axpby(1.0, x, 0.0, out)
                    be.axpby(-1.0, y, 1.0, out)
                    return out
                self.rw[i][n - 1] = krylov.gmres(be, op, self.rw[i][n - 1], prev, tol=tol, maxiter=self.maxiter, ws=self.ws)
                if n > 1:
                    self._right_cycle(i, psi)
                for s in range(n):
                    be.regularize(self.rw[i][s], self._CdC(psi, s + 1), self._eye(self.rw[i][s].shape[1]))
            else:
                if all(H[s].contains(i, i) for s in range(n)):
                    def op(x, out):
                        y = x
                        for s in range(n - 1, -1, -1):
                            y = self._tr(y, H[s].blocks[(i, i)], psi.AR[s])
                        be.axpby(1.0, x, 0.0, out)
                        be.axpby(-1.0, y, 1.0, out)
                        return out
                    self.rw[i][n - 1] = krylov.gmres(be, op, self.rw[i][n - 1], prev, tol=tol, maxiter=self.maxiter, ws=self.ws)
                if n > 1:
                    self._right_cycle(i, psi)
