"""excitations(H, QuasiparticleAnsatz(), ...) : the quasiparticle (tangent-space) ansatz on top of a finite or a
uniform ground state  (src/algorithms/excitation/quasiparticleexcitation.jl:39-143,254-362, src/environments/qpenv.jl:55-170,
src/algorithms/excitation/exci_transfer_system.jl, src/states/quasiparticle_state.jl:8-104).

Trivial charge sector: the utility leg of the excitation tensor has dimension one, B[a, s, b] = VL[a, s, n] X[n, b], so every
"excited-tensor" contraction of the reference (transfer.jl:48-62,113-126, the three terms of
_effective_excitation_local_apply) IS one of the hot-path kernels with another operand in the ket / environment slot:

    B in the centre      mpsk_dAC(H, GL,  GR,  B)
    B to the left        mpsk_dAC(H, lB,  GR,  AR)        lB = sum of the (AR ket, AL bra) transfers of B further left
    B to the right       mpsk_dAC(H, GL,  rB,  AL)
    lB / rB updates      mpsk_transfer_left/right(H, env, ket = B or AR/AL, bra = AL/AR)

Complex arithmetic.  The ground state and H are real, so every map above is REAL-linear in B; the only complex numbers
are the momentum phases.  A quasiparticle object is therefore carried as one (|sin p| = 0: momentum 0 or pi, and every
finite chain) or two (real, imaginary) real device tensors, the kernels run once per part and a phase e^{i a} mixes the two
parts with one mpsk_vlincomb each.  That is 2x the real flops for a general momentum -- the optimum for a complex vector
under a real operator (no bond embedding, no complex GEMM).  The Hermitian eigenproblem on C^N becomes the symmetric one on
R^{2N} = [Re | Im] with every eigenvalue doubled (v and i v); the Lanczos solver sees an ordinary real vector.

Built: LeftGaugedQP over an MPOHamiltonian -- FiniteQP, InfiniteQP at any momentum, topologically trivial
(left_gs is right_gs, regularised environments) and domain-wall excitations between two different ground states (no
regularisation, energies renormalised by the mean of the two); `variance` of a finite QP state (toolbox.py).  Not built:
charged sectors, the statistical-mechanics (MPOMultiline) variant, RightGaugedQP conversion."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import krylov
from .backend import DTensor
from .environments import environments as _environments
from .states import FiniteMPS


@dataclass
class QuasiparticleAnsatz:  # quasiparticleexcitation.jl:23-34 : Arnoldi(krylovdim = 30, tol = 1e-10, eager = true)
    tol: float = 1e-10
    krylovdim: int = 30
    maxiter: int = 100
    solver_tol: float = 1e-12      # Defaults.linearsolver (GMRES) of the quasiparticle environments
    solver_maxiter: int = 100


def _view(t: DTensor, off, shape):
    n = int(np.prod(shape))
    return DTensor(t.buf[off:off + n], shape)


class LeftGaugedQP:
    """quasiparticle_state.jl:8-17 : B[i] = VL[i] X[i],  AL[i]^dag VL[i] = 0.  `vec` is the flat device vector
    [Re X_0 .. Re X_{n-1} | Im X_0 .. Im X_{n-1}] (the second half only for a complex momentum phase)."""

    def __init__(self, gs, VLs, xshapes, momentum, vec: DTensor, nparts, right_gs=None):
        self.left_gs = gs
        self.right_gs = gs if right_gs is None else right_gs       # !(left_gs === right_gs) => domain-wall excitation  :9-11
        self.be = gs.be
        self.VLs, self.xshapes, self.momentum, self.vec, self.nparts = VLs, xshapes, momentum, vec, nparts
        self.finite = isinstance(gs, FiniteMPS)
        self.offs = np.concatenate([[0], np.cumsum([a * b for a, b in xshapes])]).astype(int)
        self.N = int(self.offs[-1])

    @classmethod
    def random(cls, gs, momentum=0.0, rng=None, right_gs=None):
        """LeftGaugedQP(rand, left_gs, right_gs; momentum)  :33-46 ; VL from a projected random block + QRpos (any
        orthonormal basis of the complement of AL spans the same tangent space)."""
        from .changebonds import _complement_cols
        be = gs.be
        rng = np.random.default_rng(0) if rng is None else rng
        finite = isinstance(gs, FiniteMPS)
        n = len(gs)
        ALs = [gs.AL(i) for i in range(n)] if finite else gs.AL
        rgs = gs if right_gs is None else right_gs
        ARs = [rgs.AR(i) for i in range(n)] if finite else rgs.AR
        VLs, shapes = [], []
        for i in range(n):
            Dl, d, Dr = ALs[i].shape
            VLs.append(_complement_cols(be, ALs[i].reshape(Dl * d, Dr), rng))
            shapes.append((Dl * d - Dr, ARs[i].shape[2]))
        nparts = 1 if (finite or abs(np.sin(momentum)) < 1e-12) else 2
        N = sum(a * b for a, b in shapes)
        vec = be.upload(rng.random(N * nparts))
        return cls(gs, VLs, shapes, 0.0 if finite else float(momentum), vec, nparts, right_gs)

    @property
    def trivial(self):
        return self.left_gs is self.right_gs

    def __len__(self):
        return len(self.VLs)

    def with_vec(self, vec):
        return LeftGaugedQP(self.left_gs, self.VLs, self.xshapes, self.momentum, vec, self.nparts, self.right_gs)

    def X(self, part, i, vec=None):
        vec = self.vec if vec is None else vec
        return _view(vec, part * self.N + int(self.offs[i]), self.xshapes[i])

    def B(self, part, i, vec=None):  # Base.getindex  :95
        be = self.be
        gs = self.left_gs
        al = gs.AL(i) if self.finite else gs.AL[i]
        Dl, d, _ = al.shape
        if self.xshapes[i][0] == 0:
            return be.zeros(Dl, d, self.xshapes[i][1])
        return be.gemm(self.VLs[i], self.X(part, i, vec)).reshape(Dl, d, self.xshapes[i][1])

    def B_host(self, i):
        """the excitation tensor of site i as a (complex) NumPy array -- test / inspection helper."""
        re = self.be.download(self.B(0, i))
        return re if self.nparts == 1 else re + 1j * self.be.download(self.B(1, i))


# ---- complex objects as lists of 1 or 2 real device tensors -------------------------------------------------------
def _lin(f, z):
    return [f(p) for p in z]


def _acc(be, a, b):
    """a += b (None = zero)."""
    if b is None:
        return a
    if a is None:
        return b
    for pa, pb in zip(a, b):
        be.axpby(1.0, pb, 1.0, pa)
    return a


def _phase(be, z, ang):
    """z * e^{i ang}, in place for a real phase."""
    c, s = np.cos(ang), np.sin(ang)
    if len(z) == 1:
        if round(c) != 1:
            be.scal(float(round(c)), z[0])
        return z
    re, im = z
    return [be.lincomb([re, im], [c, -s]), be.lincomb([re, im], [s, c])]


class _QPContext:
    """Everything that does not depend on X: ground-state tensors, environments, regularisation bonds, energies."""

    def __init__(self, H, phi: LeftGaugedQP, lenvs, alg: QuasiparticleAnsatz, renvs=None):
        self.H, self.alg, self.be = H, alg, phi.be
        be = self.be
        gs, rgs = phi.left_gs, phi.right_gs
        self.trivial = phi.trivial
        renvs = lenvs if (renvs is None and self.trivial) else (_environments(rgs, H) if renvs is None else renvs)
        n = self.n = len(phi)
        self.finite = phi.finite
        self.p = phi.momentum
        self.AL = [gs.AL(i) for i in range(n)] if self.finite else list(gs.AL)
        self.AR = [rgs.AR(i) for i in range(n)] if self.finite else list(rgs.AR)
        self.GL = [lenvs.leftenv(i, gs) for i in range(n)]
        self.GR = [renvs.rightenv(i, rgs) for i in range(n)]

        def energies(st, envs):        # effective_excitation_renormalization_energy  :330-362
            AC = [st.AC(i) for i in range(n)] if self.finite else list(st.AC)
            return [be.dot(AC[i], be.dAC(H[i], envs.leftenv(i, st), envs.rightenv(i, st), AC[i])) for i in range(n)]
        self.E = energies(gs, lenvs)
        if not self.trivial:
            self.E = [(a + b) / 2 for a, b in zip(self.E, energies(rgs, renvs))]
        self.odim = H.odim
        self.ids = [i for i in range(1, self.odim - 1) if H.isid(i)]
        self.ws = krylov.KrylovWorkspace(be)
        if not self.finite and self.trivial:
            self.C = list(gs.CR)                                                    # bond right of site s
            self.Ct = [be.gemm(c, be.upload(np.eye(c.shape[0])), transA=True) for c in self.C]   # C^T on the device

    # ---- level views of an assembled (W, D, D) environment ----
    def _levels(self, t: DTensor, chis):
        _, Db, Dk = t.shape
        out, off = [], 0
        for c in chis:
            out.append(_view(t, off, (c, Db, Dk)))
            off += c * Db * Dk
        return out

    def _reg(self, v: DTensor, bond):
        """v[:, w, :] -= <C, v[:, w, :]> C   (qpenv.jl:69-76 / transfermatrix.jl:87-90 with a one-dimensional MPO leg)."""
        return self.be.regularize(v, self.Ct[bond % self.n], self.C[bond % self.n])

    def _reg_ids(self, z, chis, bond):
        if not self.trivial:                # qpenv.jl:68,85 `if exci.trivial`
            return z
        for part in z:
            lv = self._levels(part, chis)
            for i in self.ids:
                self._reg(lv[i], bond)
        return z

    def _tblock(self, left, v: DTensor, O, A, Ab):
        be = self.be
        f = be.transfer_left if left else be.transfer_right
        if np.isscalar(O):
            out = f(None, v, A, Ab)
            if O != 1:
                be.scal(O, out)
            return out
        blk = be.mposlice(1, A.shape[1], [O.shape[0]], [O.shape[3]], {(0, 0): O})
        return f(blk, v, A, Ab)

    # ---- exci_transfer_system.jl ----
    def _transfer_system(self, left, start_levels):
        """x = b + e^{-+ i p n} (x T_cell), level by level (triangular H); identity levels by GMRES on the
        regularised mixed transfer matrix.  start_levels[i]: list of parts (level views)."""
        be, H, n, odim = self.be, self.H, self.n, self.odim
        ket, bra = (self.AR, self.AL) if left else (self.AL, self.AR)
        ang = (-self.p if left else self.p) * n
        c, s = np.cos(ang), np.sin(ang)
        nparts = len(start_levels[0])
        if nparts == 1:
            c, s = float(round(c)), 0.0
        found = [None] * odim
        order = range(odim) if left else range(odim - 1, -1, -1)
        sites = range(n) if left else range(n - 1, -1, -1)
        for i in order:
            # found[<i] (left) / found[>i] (right) pushed once through the unit cell, level i of the result
            v = list(found)
            last = n - 1 if left else 0
            for st in sites:
                out = [None] * odim
                rng_k = [i] if st == last else (range(0, i + 1) if left else range(i, odim))
                for k in rng_k:
                    rng_j = range(0, k + 1) if left else range(k, odim)
                    for j in rng_j:
                        blk = (j, k) if left else (k, j)
                        if v[j] is None or not H[st].contains(*blk):
                            continue
                        O = H[st].blocks[blk]
                        out[k] = _acc(be, out[k], _lin(lambda x: self._tblock(left, x, O, ket[st], bra[st]), v[j]))
                v = out
            start = v[i]
            if start is not None:
                start = _phase(be, start, ang)
                if H.isid(i) and self.trivial:
                    for part in start:
                        self._reg(part, n - 1)
            b = _acc(be, [be.copy(x) for x in start_levels[i]], start)
            if all(H[st].contains(i, i) for st in range(n)):
                isid = H.isid(i)
                shape = b[0].shape
                sz = b[0].size
                flat = be.empty(sz * nparts)
                for k, part in enumerate(b):
                    flat.buf[k * sz:(k + 1) * sz].copy_(part.buf[:sz])

                def op(x, out):
                    xs = [_view(x, k * sz, shape) for k in range(nparts)]
                    ys = []
                    for xp in xs:
                        y = xp
                        for st in sites:
                            O = 1.0 if isid else H[st].blocks[(i, i)]
                            y = self._tblock(left, y, O, ket[st], bra[st])
                        if isid and self.trivial:
                            self._reg(y, n - 1)
                        ys.append(y)
                    if nparts == 1:
                        be.lincomb([xs[0], ys[0]], [1.0, -c], out=_view(out, 0, shape))
                    else:
                        be.lincomb([xs[0], ys[0], ys[1]], [1.0, -c, s], out=_view(out, 0, shape))
                        be.lincomb([xs[1], ys[0], ys[1]], [1.0, -s, -c], out=_view(out, sz, shape))
                    return out
                sol = krylov.gmres(be, op, flat, flat, tol=self.alg.solver_tol, maxiter=self.alg.solver_maxiter, ws=self.ws)
                b = [_view(sol, k * sz, shape) for k in range(nparts)]
            found[i] = b
        return found

    def _assemble(self, levels, nparts):
        be = self.be
        out = []
        for k in range(nparts):
            parts = [lv[k] for lv in levels]
            W = sum(p.shape[0] for p in parts)
            t = be.empty(W, parts[0].shape[1], parts[0].shape[2])
            off = 0
            for p in parts:
                t.buf[off:off + p.size].copy_(p.buf[:p.size])
                off += p.size
            out.append(t)
        return out

    # ---- qpenv.jl ----
    def qp_envs(self, phi: LeftGaugedQP, vec: DTensor):
        be, H, n = self.be, self.H, self.n
        AL, AR = self.AL, self.AR
        nparts = phi.nparts
        Bs = [[phi.B(k, s, vec) for k in range(nparts)] for s in range(n)]
        lBs, rBs = [None] * n, [None] * n
        if self.finite:                                                              # :146-170
            for pos in range(n - 1):
                nxt = _lin(lambda b: be.transfer_left(H[pos], self.GL[pos], b, AL[pos]), Bs[pos])
                if lBs[pos] is not None:
                    nxt = _acc(be, nxt, _lin(lambda v: be.transfer_left(H[pos], v, AR[pos], AL[pos]), lBs[pos]))
                lBs[pos + 1] = nxt
            for pos in range(n - 1, 0, -1):
                nxt = _lin(lambda b: be.transfer_right(H[pos], self.GR[pos], b, AR[pos]), Bs[pos])
                if rBs[pos] is not None:
                    nxt = _acc(be, nxt, _lin(lambda v: be.transfer_right(H[pos], v, AL[pos], AR[pos]), rBs[pos]))
                rBs[pos - 1] = nxt
            return Bs, lBs, rBs
        p = self.p
        for pos in range(n):                                                         # :66-79
            nxt = _lin(lambda b: be.transfer_left(H[pos], self.GL[pos], b, AL[pos]), Bs[pos])
            if lBs[pos] is not None:
                nxt = _acc(be, nxt, _lin(lambda v: be.transfer_left(H[pos], v, AR[pos], AL[pos]), lBs[pos]))
            lBs[(pos + 1) % n] = self._reg_ids(_phase(be, nxt, -p), H[pos].chir, pos)
        for pos in range(n - 1, -1, -1):                                             # :81-97
            nxt = _lin(lambda b: be.transfer_right(H[pos], self.GR[pos], b, AR[pos]), Bs[pos])
            if rBs[pos] is not None and pos != n - 1:
                nxt = _acc(be, nxt, _lin(lambda v: be.transfer_right(H[pos], v, AL[pos], AR[pos]), rBs[pos]))
            rBs[(pos - 1) % n] = self._reg_ids(_phase(be, nxt, p), H[pos].chil, pos - 1)
        # :99-105  geometric sums over all unit cells further away
        lv = [self._levels(part, H[0].chil) for part in lBs[0]]
        found = self._transfer_system(True, [[lv[k][i] for k in range(nparts)] for i in range(self.odim)])
        lBs[0] = self._assemble(found, nparts)
        rv = [self._levels(part, H[n - 1].chir) for part in rBs[n - 1]]
        found = self._transfer_system(False, [[rv[k][i] for k in range(nparts)] for i in range(self.odim)])
        rBs[n - 1] = self._assemble(found, nparts)
        cur = lBs[0]
        for i in range(n - 1):                                                       # :107-123
            cur = _lin(lambda v: be.transfer_left(H[i], v, AR[i], AL[i]), cur)
            cur = self._reg_ids(_phase(be, cur, -p), H[i].chir, i)
            lBs[i + 1] = _acc(be, lBs[i + 1], cur)
        cur = rBs[n - 1]
        for i in range(n - 1, 0, -1):                                                # :124-141
            cur = _lin(lambda v: be.transfer_right(H[i], v, AL[i], AR[i]), cur)
            cur = self._reg_ids(_phase(be, cur, p), H[i].chil, i - 1)
            rBs[i - 1] = _acc(be, rBs[i - 1], cur)
        return Bs, lBs, rBs

    # ---- quasiparticleexcitation.jl:254-328 ----
    def heff(self, phi: LeftGaugedQP, x: DTensor, out: DTensor):
        be, H, n = self.be, self.H, self.n
        Bs, lBs, rBs = self.qp_envs(phi, x)
        for loc in range(n):
            if phi.xshapes[loc][0] == 0:
                continue
            Dl, d, _ = self.AL[loc].shape
            for k in range(phi.nparts):
                B = Bs[loc][k]
                Bn = be.dAC(H[loc], self.GL[loc], self.GR[loc], B)                     # B in the centre
                be.axpby(-self.E[loc], B, 1.0, Bn)
                if lBs[loc] is not None:                                              # B to the left
                    be.axpby(1.0, be.dAC(H[loc], lBs[loc][k], self.GR[loc], self.AR[loc]), 1.0, Bn)
                if rBs[loc] is not None:                                              # B to the right
                    be.axpby(1.0, be.dAC(H[loc], self.GL[loc], rBs[loc][k], self.AL[loc]), 1.0, Bn)
                be.gemm(phi.VLs[loc], Bn.reshape(Dl * d, Bn.shape[2]), transA=True, out=phi.X(k, loc, out))   # setindex! :101
        return out


def _times_i_vec(be, phi, v):
    """i v on [Re | Im]."""
    out = be.empty(v.shape)
    N = phi.N
    be.lincomb([_view(v, N, (N,))], [-1.0], out=_view(out, 0, (N,)))
    be.lincomb([_view(v, 0, (N,))], [1.0], out=_view(out, N, (N,)))
    return out


def excitations_qp(H, alg: QuasiparticleAnsatz, phi0: LeftGaugedQP, lenvs=None, renvs=None, num=1):
    """excitations(H, alg, phi0::QP, lenvs, renvs; num)  :39-64,127-143 -> (energies, [LeftGaugedQP])."""
    be = phi0.be
    lenvs = _environments(phi0.left_gs, H) if lenvs is None else lenvs
    ctx = _QPContext(H, phi0, lenvs, alg, renvs)
    eig_ws = krylov.KrylovWorkspace(be)      # NOT ctx.ws: the GMRES solves run inside this solver's matvec
    found, Es = [], []
    for _ in range(num):
        def op(x, out):
            ctx.heff(phi0, x, out)
            for f, lf in zip(found, Es):       # states already found are shifted up out of the way
                be.axpby((10.0 + 10.0 * abs(lf)) * be.dot(f, x), f, 1.0, out)
            return out
        lam, v, _, res = krylov.eigsolve_sr(be, op, phi0.vec, tol=alg.tol, krylovdim=alg.krylovdim, maxiter=alg.maxiter,
                                            ws=eig_ws)
        if res > alg.tol:
            import warnings
            warnings.warn(f"excitation failed to converge: normres = {res:.3e}")     # :47-48
        v = be.copy(v)
        Es.append(float(lam))
        found.append(v)
        if phi0.nparts == 2:
            found.append(_times_i_vec(be, phi0, v))
            Es.append(float(lam))
    keep = range(0, len(found), phi0.nparts)
    return [Es[k] for k in keep], [phi0.with_vec(found[k]) for k in keep]


def excitations_momenta(H, alg: QuasiparticleAnsatz, momenta, psi, lenvs=None, rpsi=None, renvs=None, num=1, rng=None):
    """excitations(H, alg, momentum | momenta, lmps::InfiniteMPS, lenvs, rmps, renvs; num)  :84-125.  A scalar momentum
    returns (energies[num], states[num]); a list returns (E[len(momenta), num], states[len(momenta)][num]).  rmps different
    from lmps: domain-wall (topologically non-trivial) excitations between two ground states."""
    lenvs = _environments(psi, H) if lenvs is None else lenvs
    rpsi = psi if rpsi is None else rpsi
    renvs = (lenvs if rpsi is psi else _environments(rpsi, H)) if renvs is None else renvs
    rng = np.random.default_rng(0) if rng is None else rng
    if np.isscalar(momenta):
        return excitations_qp(H, alg, LeftGaugedQP.random(psi, momenta, rng, rpsi), lenvs, renvs, num)
    Ep, Bp = [], []
    for p in momenta:
        e, b = excitations_qp(H, alg, LeftGaugedQP.random(psi, float(p), rng, rpsi), lenvs, renvs, num)
        Ep.append(e)
        Bp.append(b)
    return np.array(Ep), Bp
