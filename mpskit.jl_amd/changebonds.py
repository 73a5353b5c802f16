"""changebonds (src/algorithms/changebonds/optimalexpand.jl:72-102, svdcut.jl:14-23) on the HIP path: pure reuse
of the hot-path operators (ddAC2 matvec, QRpos / LQpos, tsvd, GEMMs) -- SURVEY §8(f) item 1."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .backend import DTensor
from .derivatives import ddAC2
from .environments import FinEnv, environments
from .states import FiniteMPS, leftorth, _qr, _lq


@dataclass
class OptimalExpand:  # optimalexpand.jl:12-14 ; trscheme = truncdim(1)
    trunc_dim: int = 1


@dataclass
class SvdCut:  # svdcut.jl:9-11 ; trscheme = notrunc()
    trunc_dim: int = 0
    trunc_err: float = 0.0


def _rand(rng, r, c, cplx):
    """random block; complex (embedded) states: the embedding of a complex Gaussian block, so that everything derived
    from it by products with embedded tensors and structured QRpos / LQpos stays an embedding (cplx.py)."""
    if not cplx:
        return rng.standard_normal((r, c))
    from .cplx import embed
    return embed(rng.standard_normal((r // 2, c // 2)) + 1j * rng.standard_normal((r // 2, c // 2)))


def _complement_cols(be, Q: DTensor, rng, cplx=False):
    """Orthonormal basis N (m x (m-n)) of the complement of the orthonormal columns Q (m x n): QRpos of a
    projected random block, projected twice (TensorKit leftnull returns the trailing columns of the full QR
    factor; every orthonormal basis of the complement gives the same expansion)."""
    m, n = Q.shape
    if m == n:
        return None
    Y = be.upload(_rand(rng, m, m - n, cplx))
    for _ in range(2):
        t = be.gemm(Q, Y, transA=True)
        be.gemm(Q, t, alpha=-1.0, beta=1.0, out=Y)
        Y, _ = _qr(be, Y, cplx)
    return Y


def _complement_rows(be, B: DTensor, rng, cplx=False):
    """Orthonormal rows N ((n-m) x n) spanning the complement of the orthonormal rows of B (m x n)."""
    m, n = B.shape
    if m == n:
        return None
    Y = be.upload(_rand(rng, n - m, n, cplx))
    for _ in range(2):
        t = be.gemm(Y, B, transB=True)                 # (n-m) x m
        be.gemm(t, B, alpha=-1.0, beta=1.0, out=Y)
        _, Y = _lq(be, Y, cplx)
    return Y


def _tail_matrix(be, A: DTensor):
    """A[m, s, b] -> M[m, (b, s)] (the column order of the two-site tensor theta[a, s1, b, s2])."""
    Dm, d, Dr = A.shape
    M = be.empty(Dm, Dr * d)
    for s in range(d):
        be.copy2d(Dm, Dr, A.ptr + 8 * s * Dm, Dm * d, M.ptr + 8 * s * Dm * Dr, Dm)
    return M


def _from_tail_matrix(be, M: DTensor, d, Dr, out: DTensor, row0):
    """rows of M[k, (b, s)] -> out[row0 + k, s, b]."""
    k = M.shape[0]
    Dn = out.shape[0]
    for s in range(d):
        be.copy2d(k, Dr, M.ptr + 8 * s * k * Dr, k, out.ptr + 8 * (row0 + s * Dn), Dn * d)


def _optimal_expand_cplx(psi: FiniteMPS, H, alg: OptimalExpand, envs, rng):
    """optimalexpand.jl:72-102 on an EMBEDDED complex state (cplx.py).  Same steps as the real version; what changes:
    * the complements NL / NR are built from embedded random blocks with the structured QRpos / LQpos, so they are
      embeddings of complex isometries;
    * the dominant right singular SUBSPACE of NL^dag (H_AC2 AC2) NR^dag comes from cplx.split_two_site (the plain tsvd of
      an embedding returns an arbitrary basis inside every doubled singular value); any orthonormal basis of that subspace
      gives the same expanded state as the reference's V rows, because AC is padded with zeros;
    * leftorth([AC | 0]) is assembled instead of factored: [AL | N_k] [[C, 0], [0, 0]] with N_k the first k (complex)
      columns of NL -- a QRpos of the padded matrix whose completion is an embedding by construction (a numerical QR of
      a rank-deficient matrix completes it with arbitrary, unstructured columns)."""
    be, L = psi.be, len(psi)
    from .algorithms import _two_site_tensor
    from .cplx import split_two_site
    for i in range(L - 1):
        ac, ar = psi.AC(i), psi.AR(i + 1)
        Dl, d1, Dm = ac.shape                                        # (embedded: all bond dimensions doubled)
        _, d2, Dr = ar.shape
        nl, nr = Dl * d1 - Dm, d2 * Dr - Dm
        if nl <= 0 or nr <= 0:
            continue
        ac2 = ddAC2(i, psi, H, envs)(_two_site_tensor(be, ac, ar))
        al_old, c_old = leftorth(be, ac, True)
        NL = _complement_cols(be, al_old.reshape(Dl * d1, Dm), rng, True)     # (Dl d1) x nl
        Bm = _tail_matrix(be, ar)                                    # Dm x (Dr d2), orthonormal rows, (b, s) columns
        NR = _complement_rows(be, Bm, rng, True)                     # nr x (Dr d2)
        t = be.gemm(NL, ac2.reshape(Dl * d1, Dr * d2), transA=True)
        inter = be.gemm(t, NR, transB=True)                          # nl x nr (embedded)
        k = min(alg.trunc_dim, nl // 2, nr // 2)                     # complex dimensions
        _, _, ars, _, _ = split_two_site(be, inter.reshape(nl, 1, nr, 1), trunc_dim=k, rng=rng)
        K2 = ars.shape[0]                                            # 2 k real rows: an embedded isometry K2 x nr
        are = be.gemm(ars.reshape(K2, nr), NR)                       # K2 x (Dr d2) in (b, s) column order
        nal = be.empty(Dl, d1, Dm + K2)                              # [AL | N_k]
        be.copy2d(Dl * d1, Dm, al_old.ptr, Dl * d1, nal.ptr, Dl * d1)
        be.copy2d(Dl * d1, K2, NL.ptr, Dl * d1, nal.ptr + 8 * Dl * d1 * Dm, Dl * d1)
        nc = be.zeros(Dm + K2, Dm + K2)
        be.copy2d(Dm, Dm, c_old.ptr, Dm, nc.ptr, Dm + K2)
        nar = be.empty(Dm + K2, d2, Dr)
        be.copy2d(Dm, d2 * Dr, ar.ptr, Dm, nar.ptr, Dm + K2)
        _from_tail_matrix(be, are, d2, Dr, nar, Dm)
        psi.set_AC(i, (nal, nc))
        psi.set_AC(i + 1, (nc, nar))
    return psi, envs


def _svd_cut_cplx(psi: FiniteMPS, alg: SvdCut):
    """svdcut.jl:14-23 on an EMBEDDED complex state: the truncated decomposition of the bond matrix comes from
    cplx.split_two_site (structured: kept subspace invariant under multiplication by i), so the new bond matrix is
    TRIANGULAR with the Schmidt values as singular values instead of diag(S) -- the state is the same truncated state."""
    be, L = psi.be, len(psi)
    from .cplx import split_two_site
    for i in range(L - 2, -1, -1):
        c = psi.CR(i)
        al, ar = psi.AL(i), psi.AR(i + 1)
        Dl, d, Dm = al.shape
        _, d2, Dr = ar.shape
        als, cs, ars, _, _ = split_two_site(be, c.reshape(Dm, 1, c.shape[1], 1), alg.trunc_dim, alg.trunc_err)
        k = cs.shape[0]
        nal = be.gemm(al.reshape(Dl * d, Dm), als.reshape(Dm, k)).reshape(Dl, d, k)
        nar = be.gemm(ars.reshape(k, c.shape[1]), ar.reshape(c.shape[1], d2 * Dr)).reshape(k, d2, Dr)
        psi.set_AC(i, (nal, cs))
        psi.set_AC(i + 1, (cs, nar))
    new = be.copy(psi.AC(L - 1))
    be.scal(1.0 / psi.norm(), new)
    psi.set_AC(L - 1, new)
    return psi


def _optimal_expand(psi: FiniteMPS, H, alg: OptimalExpand, envs, rng):
    be, L = psi.be, len(psi)
    from .algorithms import _two_site_tensor
    for i in range(L - 1):
        ac, ar = psi.AC(i), psi.AR(i + 1)
        Dl, d1, Dm = ac.shape
        _, d2, Dr = ar.shape
        nl, nr = Dl * d1 - Dm, d2 * Dr - Dm
        if nl <= 0 or nr <= 0:
            continue
        ac2 = ddAC2(i, psi, H, envs)(_two_site_tensor(be, ac, ar))
        Qac, _ = be.qrpos(ac.reshape(Dl * d1, Dm))
        NL = _complement_cols(be, Qac, rng)                          # (Dl d1) x nl
        Bm = _tail_matrix(be, ar)                                    # Dm x (Dr d2), orthonormal rows
        NR = _complement_rows(be, Bm, rng)                           # nr x (Dr d2)
        t = be.gemm(NL, ac2.reshape(Dl * d1, Dr * d2), transA=True)  # nl x (Dr d2)
        inter = be.gemm(t, NR, transB=True)                          # nl x nr
        k = min(alg.trunc_dim, nl, nr)
        _, _, Vh, kept, _ = be.tsvd(inter, max_keep=k)
        k = kept
        kmax = Vh.shape[0]
        are = be.empty(k, Dr * d2)                                   # ar_re = V NR  in (b, s) column order
        be.gemm_raw(False, False, k, Dr * d2, nr, 1.0, Vh.ptr, kmax, NR.ptr, nr, 0.0, are.ptr, k)
        # [AC | 0] -> leftorth ;  [AR ; ar_re]
        ext = be.zeros(Dl, d1, Dm + k)
        be.copy2d(Dl * d1, Dm, ac.ptr, Dl * d1, ext.ptr, Dl * d1)
        nal, nc = leftorth(be, ext)
        nar = be.empty(Dm + k, d2, Dr)
        be.copy2d(Dm, d2 * Dr, ar.ptr, Dm, nar.ptr, Dm + k)
        _from_tail_matrix(be, are, d2, Dr, nar, Dm)
        psi.set_AC(i, (nal, nc))
        psi.set_AC(i + 1, (nc, nar))
    return psi, envs


def _svd_cut(psi: FiniteMPS, alg: SvdCut):
    be, L = psi.be, len(psi)
    for i in range(L - 2, -1, -1):
        c = psi.CR(i)
        al, ar = psi.AL(i), psi.AR(i + 1)
        U, S, Vh, k, _ = be.tsvd(c, max_keep=alg.trunc_dim, trunc_err=alg.trunc_err)
        Dl, d, Dm = al.shape
        _, d2, Dr = ar.shape
        nal = be.empty(Dl, d, k)
        be.gemm_raw(False, False, Dl * d, k, Dm, 1.0, al.ptr, Dl * d, U.ptr, U.shape[0], 0.0, nal.ptr, Dl * d)
        nar = be.empty(k, d2, Dr)
        be.gemm_raw(False, False, k, d2 * Dr, Dm, 1.0, Vh.ptr, Vh.shape[0], ar.ptr, Dm, 0.0, nar.ptr, k)
        cm = be.upload(np.diag(be.download(DTensor(S.buf, (k,)))))
        psi.set_AC(i, (nal, cm))
        psi.set_AC(i + 1, (cm, nar))
    last = psi.AC(L - 1)
    new = be.copy(last)
    be.scal(1.0 / psi.norm(), new)
    psi.set_AC(L - 1, new)
    return psi


def changebonds(psi, H=None, alg=None, envs=None, rng=None):
    """changebonds(psi, H, alg[, envs]) -> (psi', envs)  /  changebonds(psi, SvdCut(...)) -> psi'   (copying versions)."""
    if isinstance(H, (OptimalExpand, SvdCut)) and alg is None:
        H, alg = None, H
    if not isinstance(psi, FiniteMPS):
        raise NotImplementedError("changebonds is built for FiniteMPS (optimalexpand.jl:72-102, svdcut.jl:14-23)")
    cx = bool(getattr(psi, "cplx", False))
    psi = psi.copy()
    if isinstance(alg, SvdCut):
        out = _svd_cut_cplx(psi, alg) if cx else _svd_cut(psi, alg)
        return out if H is None else (out, envs)
    if isinstance(alg, OptimalExpand):
        envs = environments(psi, H) if envs is None else envs
        fn = _optimal_expand_cplx if cx else _optimal_expand
        return fn(psi, H, alg, envs, np.random.default_rng(0) if rng is None else rng)
    raise TypeError(f"unknown changebonds algorithm {alg!r}")
