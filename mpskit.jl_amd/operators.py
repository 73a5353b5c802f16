"""MPOHamiltonian / SparseMPOSlice on the device side (src/operators/mpohamiltonian.jl:8-31,
sparsempo/sparseslice.jl:13-106 of the reference).  The container logic is host-only (no flops);
each distinct slice owns one `mpsk_mposlice` handle.  Model builders mirror the toy Hamiltonians
of test/setup.jl:38-65 and docs/src/man/operators.md:52-78."""
from __future__ import annotations

import math

import numpy as np

from .backend import Backend, DeviceMPOSlice, default_backend


class MPOHamiltonian:
    """Periodic list of block-sparse slices with H[.][1,1] = 1 and H[.][odim,odim] = 1
    (mpohamiltonian.jl:8-31).  `data[site]` = dict {(i, j): scalar | d x d | [chi_i, d, d, chi_j]}."""

    def __init__(self, data, d=None, chis=None, be: Backend | None = None):
        self.be = default_backend() if be is None else be
        if isinstance(data, dict):
            data = [data]
        self.period = len(data)
        self.odim = 1 + max(max(i, j) for blk in data for (i, j) in blk)
        odim = self.odim
        # infer d and the level dimensions chi from the blocks
        if d is None:
            for blk in data:
                for v in blk.values():
                    if not np.isscalar(v):
                        a = np.asarray(v)
                        d = a.shape[0] if a.ndim == 2 else a.shape[1]
                        break
                if d is not None:
                    break
        if d is None:
            raise ValueError("physical dimension cannot be inferred from scalar-only blocks; pass d=")
        self.d = int(d)
        if chis is None:
            chis = [[1] * odim for _ in range(self.period + 1)]
            for s, blk in enumerate(data):
                for (i, j), v in blk.items():
                    if not np.isscalar(v) and np.asarray(v).ndim == 4:
                        a = np.asarray(v)
                        chis[s][i] = a.shape[0]
                        chis[s + 1][j] = a.shape[3]
            for i in range(odim):  # periodic closure
                m = max(chis[0][i], chis[self.period][i])
                chis[0][i] = chis[self.period][i] = m
        self.chis = chis
        self.slices = []
        for s, blk in enumerate(data):
            blocks = {}
            for (i, j), v in blk.items():
                if np.isscalar(v):
                    if v != 0:
                        blocks[(i, j)] = v
                    continue
                a = np.asarray(v)
                if a.ndim == 2:
                    a = a[None, :, :, None]
                if np.abs(a).max() < 1e-14:                      # sparsempo.jl:122-132
                    continue
                if a.shape[0] == a.shape[3]:
                    c = a[0, 0, 0, 0]
                    ident = np.einsum("wv,ts->wtsv", np.eye(a.shape[0]), np.eye(self.d))
                    if c != 0 and np.allclose(a, c * ident, rtol=0, atol=1e-14 * max(1.0, abs(c))):
                        blocks[(i, j)] = 1.0 if abs(c - 1) < 1e-14 else c
                        continue
                blocks[(i, j)] = a
            self.slices.append(self.be.mposlice(odim, self.d, chis[s], chis[s + 1], blocks))
        self._energy_slices = {}

    def __getitem__(self, i):
        return self.slices[i % self.period]

    def __len__(self):
        return self.period

    def isid(self, i):  # mpohamiltonian.jl:53-58
        return all(s.isscal(i, i) and abs(s.blocks[(i, i)] - 1) < 1e-14 for s in self.slices)

    # ---- arithmetic (mpohamiltonian.jl:77-160) ----
    def _src(self):
        return [(sl.blocks, sl.chil, sl.chir) for sl in self.slices]

    def __mul__(self, other):
        """b * a for two MPOHamiltonians (a applied first; sparsempo.jl:232-264) or b * number (mpohamiltonian.jl:147-154:
        every block of the last column but the corner is scaled)."""
        if isinstance(other, MPOHamiltonian):
            if other.period != self.period:
                raise ValueError(f"periodicity should match {self.period} != {other.period}")
            data, dims = _mpo_product_data(other._src(), self._src(), self.d)
            return MPOHamiltonian(data, d=self.d, chis=dims, be=self.be)
        data = []
        for sl in self.slices:
            blk = dict(sl.blocks)
            for (i, j), v in sl.blocks.items():
                if j == self.odim - 1 and i < self.odim - 1:
                    blk[(i, j)] = v * other
            data.append(blk)
        return MPOHamiltonian(data, d=self.d, chis=[list(c) for c in self.chis], be=self.be)

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    def __add__(self, e):
        """H + e (mpohamiltonian.jl:78-94): e[c] * identity added to the on-site block (1, odim) of site c; H + H' is LazySum's job."""
        e = np.broadcast_to(np.asarray(e, dtype=float), (self.period,))
        data = []
        for c, sl in enumerate(self.slices):
            blk = dict(sl.blocks)
            cur = blk.get((0, self.odim - 1), 0.0)
            cur = cur * np.eye(self.d) if np.isscalar(cur) else np.asarray(cur)[0, :, :, 0]
            blk[(0, self.odim - 1)] = (cur + e[c] * np.eye(self.d))[None, :, :, None]
            data.append(blk)
        return MPOHamiltonian(data, d=self.d, chis=[list(c) for c in self.chis], be=self.be)

    def __sub__(self, e):
        return self + (-np.asarray(e, dtype=float))

    def repeat(self, n):
        """Base.repeat(H, n)  (mpohamiltonian.jl:157)."""
        data = [dict(sl.blocks) for sl in self.slices] * n
        chis = [list(c) for c in self.chis[:-1]] * n + [list(self.chis[-1])]
        return MPOHamiltonian(data, d=self.d, chis=chis, be=self.be)

    def energy_slice(self, site):
        """Slice holding only the blocks that enter the per-site energy of expval.jl:92-109
        ((j == 1 and k != 1) or (k == odim and j != odim)), halved unless (j, k) == (1, odim);
        <AC| dAC_E |AC> / ||AC||^2 is then the reference's ens[site]."""
        key = site % self.period
        if key not in self._energy_slices:
            s = self.slices[key]
            odim = self.odim
            blocks = {}
            for (j, k), v in s.blocks.items():
                if not ((j == 0 and k != 0) or (k == odim - 1 and j != odim - 1)):
                    continue
                f = 1.0 if (j == 0 and k == odim - 1) else 0.5
                blocks[(j, k)] = f * v if np.isscalar(v) else f * np.asarray(v)
            self._energy_slices[key] = self.be.mposlice(odim, self.d, s.chil, s.chir, blocks)
        return self._energy_slices[key]


def _mpo_product_data(srcA, srcB, d):
    """SparseMPO product b * a (src/operators/sparsempo/sparsempo.jl:232-264): a is applied first.  srcX[s] = (blocks, chil,
    chir).  New level (i, k) -> i + odim_a * k with dimension chi_a[i] * chi_b[k] (a's index fastest)."""
    oa, ob = len(srcA[0][1]), len(srcB[0][1])
    eye_d = np.eye(d)

    def dense(O, cl, cr):
        if np.isscalar(O):
            return O * np.einsum("wv,ts->wtsv", np.eye(cl, cr), eye_d)
        return np.asarray(O, dtype=float)
    data, dims = [], []
    for (ba, cla, cra), (bb, clb, crb) in zip(srcA, srcB):
        out = {}
        for (i, j), Oa in ba.items():
            for (k, l), Ob in bb.items():
                if np.isscalar(Oa) and np.isscalar(Ob):
                    out[(i + oa * k, j + oa * l)] = Oa * Ob
                    continue
                A, B = dense(Oa, cla[i], cra[j]), dense(Ob, clb[k], crb[l])
                t = np.einsum("ausx,btuy->abtsxy", A, B)                                   # :256-259
                out[(i + oa * k, j + oa * l)] = t.reshape(A.shape[0] * B.shape[0], d, d, A.shape[3] * B.shape[3], order="F")
        data.append(out)
        dims.append([cla[i] * clb[k] for k in range(ob) for i in range(oa)])
    dims.append([srcA[-1][2][i] * srcB[-1][2][k] for k in range(ob) for i in range(oa)])
    return data, dims


def _pbc_blocks(src, L, d):
    """periodic_boundary_conditions(H::MPOHamiltonian, len)  (src/algorithms/toolbox.jl:186-307) as a state machine on
    integer levels.  src[s] = (blocks {(j, k): scalar | [chi_j, d, d, chi_k]}, chil, chir) of the periodic slices.
    New level (a, b, c): a = progress of the upper layer, b = the level 'lent' across the closing bond, c = progress of
    the lower layer (the part of a wrapped term that sits at the beginning of the chain); b = odim-1 means nothing is lent
    (the open-boundary terms).  Level dimension chi[a] * chi0[b] * chi[c], fused with a fastest.  A term never wraps twice.
    Returns (data, nlev): data[site] = {(J, K): dense [chiJ, d, d, chiK]} for the L sites of the ring."""
    p = len(src)
    if L % p != 0:
        raise ValueError(f"{L} is not a multiple of the unit cell")                       # :190-191
    chi = len(src[0][1])
    top = chi - 1
    ind, n = {}, 0
    for b in range(top, 0, -1):                                                           # :208-211
        for c in range(b, chi):
            ind[(0, b, c)] = n
            n += 1
    for a in range(1, chi):                                                               # :213-216
        for b in range(top, a - 1, -1):
            ind[(a, b, top)] = n
            n += 1
    chi0 = list(src[0][1])                               # dimensions of the lent leg: the bond that closes the ring
    eye_d = np.eye(d)

    def dense(O, cl, cr):
        if np.isscalar(O):
            return O * np.einsum("wv,ts->wtsv", np.eye(cl, cr), eye_d)
        return np.asarray(O, dtype=float)

    def upper(O, cb):          # O on the a leg, identity on the lent leg:  [(alpha, beta), t, s, (alpha', beta')]
        ca, _, _, ca2 = O.shape
        out = np.einsum("atsx,by->abtsxy", O, np.eye(cb))
        return out.reshape(ca * cb, d, d, ca2 * cb, order="F")

    def lower(O, cb):          # identity on the lent leg, O on the c leg:  [(beta, gamma), t, s, (beta', gamma')]
        cc, _, _, cc2 = O.shape
        out = np.einsum("by,gtsz->bgtsyz", np.eye(cb), O)
        return out.reshape(cb * cc, d, d, cb * cc2, order="F")

    data = []
    for site in range(L):
        blocks, chil, chir = src[site % p]
        out = {}
        for (j, k), O in blocks.items():
            Od = dense(O, chil[j], chir[k])
            if site == 0:                                                                 # starter  :258-281
                if j == 0:
                    out[(0, ind[(k, top, top)])] = Od
                elif j < top:
                    out[(0, ind[(0, j, k)])] = np.einsum("btsg->tsbg", Od).reshape(d, d, chil[j] * chir[k], order="F")[None]
                continue
            if site == L - 1:                                                             # ender  :283-296
                if k >= 1:
                    out[(ind[(j, k, top)], n - 1)] = np.einsum("atsb->abts", Od).reshape(chil[j] * chir[k], d, d, order="F")[..., None]
                continue
            for i in range(1, chi):                                                       # bulk, (j, k) above  :225-238
                if k <= i:
                    out[(ind[(j, i, top)], ind[(k, i, top)])] = upper(Od, chi0[i])
            for l in range(1, top):                                                       # bulk, (j, k) below  :240-254
                if l <= j:
                    out[(ind[(0, l, j)], ind[(0, l, k)])] = lower(Od, chi0[l])
        data.append(out)
    return data, n


def _pbc_level_dims(src, L, nlev):
    """dimension of every new level on the L + 1 bonds of the ring MPO (see _pbc_blocks)."""
    p, chi = len(src), len(src[0][1])
    top = chi - 1
    chi0 = list(src[0][1])
    states = []
    for b in range(top, 0, -1):
        for c in range(b, chi):
            states.append((0, b, c))
    for a in range(1, chi):
        for b in range(top, a - 1, -1):
            states.append((a, b, top))
    assert len(states) == nlev
    dims = []
    for s in range(L + 1):
        cs = src[s % p][1] if s < L else src[(L - 1) % p][2]
        dims.append([cs[a] * chi0[b] * cs[c] for (a, b, c) in states])
    return dims


def periodic_boundary_conditions(H: "MPOHamiltonian", L=None):
    """periodic_boundary_conditions(H, len)  (toolbox.jl:181-307): the MPOHamiltonian of a ring of `len` sites, as an open
    chain of `len` site-dependent slices with odim (odim - 1) levels."""
    L = H.period if L is None else L
    src = [(H[s].blocks, H[s].chil, H[s].chir) for s in range(H.period)]
    data, nlev = _pbc_blocks(src, L, H.d)
    return MPOHamiltonian(data, d=H.d, chis=_pbc_level_dims(src, L, nlev), be=H.be)


# ---- models -----------------------------------------------------------------------------------

def spin_ops(spin=0.5):
    d = int(round(2 * spin + 1))
    m = spin - np.arange(d)
    Sz = np.diag(m)
    Sp = np.zeros((d, d))
    for k in range(1, d):
        Sp[k - 1, k] = math.sqrt(spin * (spin + 1) - m[k] * (m[k] + 1))
    return Sz, Sp, Sp.T.copy()


def heisenberg_XXX(spin=0.5, J=1.0, be=None):
    """Real 5-level Heisenberg MPO: H = J sum Sz Sz + (S+ S- + S- S+)/2 (block pattern of
    docs/src/man/operators.md:67-78)."""
    Sz, Sp, Sm = spin_ops(spin)
    return MPOHamiltonian({(0, 0): 1.0, (4, 4): 1.0, (0, 1): J * Sz, (1, 4): Sz, (0, 2): 0.5 * J * Sp,
                           (2, 4): Sm, (0, 3): 0.5 * J * Sm, (3, 4): Sp}, be=be)


def transverse_field_ising(J=1.0, g=1.0, be=None):
    """H = -J sum Z Z - g sum X (Pauli), docs/src/man/operators.md:52-58."""
    X = np.array([[0.0, 1], [1, 0]])
    Z = np.array([[1.0, 0], [0, -1]])
    return MPOHamiltonian({(0, 0): 1.0, (2, 2): 1.0, (0, 1): -J * Z, (1, 2): Z, (0, 2): -g * X}, be=be)


def from_twosite(h2, tol=1e-12, be=None):
    """MPOHamiltonian(h::TensorMap two-site): SVD split (mpohamiltonian.jl:16-31, utility.jl:42-54)."""
    d = h2.shape[0]
    M = np.transpose(h2, (0, 2, 1, 3)).reshape(d * d, d * d)
    U, S, Vh = np.linalg.svd(M)
    keep = S > tol
    U, S, Vh = U[:, keep], S[keep], Vh[keep, :]
    r = len(S)
    A = (U * S).reshape(d, d, r)[None, :, :, :]
    B = Vh.reshape(r, d, d)[:, :, :, None]
    return MPOHamiltonian({(0, 0): 1.0, (0, 1): A, (1, 2): B, (2, 2): 1.0}, be=be)


def hubbard(t=1.0, U=4.0, be=None):
    """Spinful Hubbard chain via Jordan-Wigner, d = 4, 6 MPO levels (model of BASELINE config 4)."""
    c = np.array([[0.0, 1], [0, 0]])
    P = np.diag([1.0, -1.0])
    I2 = np.eye(2)
    cu, cd, F = np.kron(c, I2), np.kron(P, c), np.kron(P, P)
    nu, nd = cu.T @ cu, cd.T @ cd
    return MPOHamiltonian({(0, 0): 1.0, (5, 5): 1.0, (0, 5): U * (nu @ nd),
                           (0, 1): -t * (cu.T @ F), (1, 5): cu, (0, 2): t * (cu @ F), (2, 5): cu.T,
                           (0, 3): -t * (cd.T @ F), (3, 5): cd, (0, 4): t * (cd @ F), (4, 5): cd.T}, be=be)


class LazySum(list):
    """LazySum(ops[, fs])  (src/operators/lazysum.jl:16-59): a sum of operators that is never formed; every
    effective Hamiltonian / expectation value is the (weighted) sum of the terms' own (multipleenv.jl,
    derivatives.jl:310-323).  fs: optional scalar prefactors (MultipliedOperator with constant factors)."""

    def __init__(self, ops, fs=None):
        super().__init__(ops)
        self.fs = [1.0] * len(self) if fs is None else [float(f) for f in fs]
        if len(self.fs) != len(self):
            raise ValueError("LazySum: one prefactor per operator")

    def __add__(self, other):
        if isinstance(other, LazySum):
            return LazySum(list(self) + list(other), self.fs + other.fs)
        return LazySum(list(self) + [other], self.fs + [1.0])
