"""complex128 states on the real fp64 kernels: every complex number z = r + i m is carried as the 2x2 real matrix
[[r, -m], [m, r]] on the BOND indices of the MPS tensors (interleaved embedding):

    E[2a + alpha, s, 2b + beta] = [[Re A, -Im A], [Im A, Re A]][alpha][beta]   at (a, s, b).

Products of embedded matrices are the embeddings of the complex products and transposition is the embedding of
the conjugate transpose, so every contraction of the hot path (dAC / dC / transfers with a REAL MPO), run
unchanged on the embedded tensors, returns the embedding of the complex result; QRpos / LQpos of an embedded
matrix are the embeddings of the complex QRpos / LQpos (the embedded R is upper triangular with a positive
diagonal and the factorization is unique); a Krylov solver started from an embedded tensor stays in the embedded
subspace.  Cost: 8x the real flops (a native complex kernel needs 4x) and 4x the memory -- the price of reusing
the fp64 MFMA path of this round unchanged (DESIGN.md section 7 has the native plan).  The chain's boundary bond
dimension becomes 2.  Supported: FiniteMPS / InfiniteMPS, DMRG, DMRG2, VUMPS, TDVP / TDVP2 incl. real time,
calc_galerkin, expectation_value, real MPO Hamiltonians.  The two-site split of an embedded tensor is NOT the plain
tsvd of the embedding (its singular vectors are only defined up to a rotation inside each doubled singular value):
see split_two_site below."""
from __future__ import annotations

import numpy as np

from .backend import DTensor


def embed(A):
    """complex (or real) host tensor [Dl, d..., Dr] -> real embedded [2 Dl, d..., 2 Dr] (bond indices first / last)."""
    A = np.asarray(A)
    Ar, Ai = np.real(A), np.imag(A)
    Dl, Dr = A.shape[0], A.shape[-1]
    E = np.zeros((2 * Dl,) + A.shape[1:-1] + (2 * Dr,))
    E[0::2, ..., 0::2] = Ar
    E[0::2, ..., 1::2] = -Ai
    E[1::2, ..., 0::2] = Ai
    E[1::2, ..., 1::2] = Ar
    return E


def extract(E):
    """inverse of embed (reads the first column of every 2x2 block)."""
    E = np.asarray(E)
    return E[0::2, ..., 0::2] + 1j * E[1::2, ..., 0::2]


def structure_defect(E):
    """max deviation of E from the embedded form (0 for an exact embedding)."""
    E = np.asarray(E)
    return max(np.abs(E[0::2, ..., 0::2] - E[1::2, ..., 1::2]).max(initial=0.0),
               np.abs(E[0::2, ..., 1::2] + E[1::2, ..., 0::2]).max(initial=0.0))


def times_i(be, x: DTensor, out: DTensor = None):
    """emb(i * z) = (I_Dl (x) J) . emb(z) with J = [[0, -1], [1, 0]] acting on the first (left bond) index:
    rows (2a, 2a+1) -> (-row 2a+1, row 2a)  (mpsk_vtimes_i)."""
    return be.times_i(x, out)


def split_two_site(be, theta: DTensor, trunc_dim=0, trunc_err=0.0, rng=None):
    """Truncated 'SVD split' of an EMBEDDED two-site tensor theta_E[(2 Dl), d1, (2 Dr), d2] (the complex
    tsvd!(theta; trunc) of dmrg.jl:96 / tdvp.jl:124):  theta ~ al . c . ar  with al left-isometric, ar right-isometric,
    all three embedded, and the kept subspace = the dominant complex singular subspace.

    The real tsvd of the embedding returns every complex singular value twice with an ARBITRARY orthonormal basis of
    each doubled (or, for degenerate complex values, 2g-fold) subspace, so its U / V columns are not embeddings.
    What is well defined is the kept subspace as long as it is invariant under J ("times i"):
      1. cut at an even count; if the cut falls inside a cluster of equal singular values, keep a J-invariant part of
         that cluster's subspace (structured random projection + QRpos), which costs no truncation error;
      2. al = QRpos(P_kept X) for a structured random X: an embedded orthonormal basis of the kept subspace;
      3. (c, ar) = LQpos(al^T theta): c is lower triangular instead of diag(S) - the two-site drivers only need
         al . c . ar (dmrg.jl:97-104), the Schmidt values are still returned.
    Returns (al, c, ar, S_complex, discarded_norm) with ar in [k, s2, b] order."""
    rng = np.random.default_rng(0) if rng is None else rng
    Dl2, d1, Dr2, d2 = theta.shape
    m2, n2 = Dl2 * d1, Dr2 * d2
    th = theta.reshape(m2, n2)
    U, S, Vh, kept, _ = be.tsvd(th, max_keep=2 * trunc_dim if trunc_dim > 0 else 0, trunc_err=trunc_err)
    s = be.download(S)
    kmax = len(s)
    K2 = min(kept + (kept & 1), kmax - (kmax & 1))
    tol_c = 1e-8
    lo = K2
    while lo > 0 and abs(s[lo - 1] - s[K2 - 1]) <= tol_c * s[K2 - 1] + 1e-14 * s[0]:
        lo -= 1
    hi = K2
    while hi < kmax and abs(s[hi] - s[K2 - 1]) <= tol_c * s[K2 - 1] + 1e-14 * s[0]:
        hi += 1
    lo -= lo & 1
    hi += hi & 1
    hi = min(hi, kmax)
    B = be.empty(m2, K2)                                       # orthonormal basis of the kept subspace
    if hi == K2:
        be.copy2d(m2, K2, U.ptr, m2, B.ptr, m2)
    else:                                                      # the cut splits the cluster [lo, hi)
        if lo > 0:
            be.copy2d(m2, lo, U.ptr, m2, B.ptr, m2)
        r2 = K2 - lo
        Uc = be.empty(m2, hi - lo)
        be.copy2d(m2, hi - lo, U.ptr + 8 * m2 * lo, m2, Uc.ptr, m2)
        Xc = be.upload(embed(rng.standard_normal((m2 // 2, r2 // 2)) + 1j * rng.standard_normal((m2 // 2, r2 // 2))))
        Wc = be.gemm(Uc, be.gemm(Uc, Xc, transA=True))
        Qc, _ = be.qrpos(Wc)
        be.copy2d(m2, r2, Qc.ptr, m2, B.ptr + 8 * m2 * lo, m2)
    Xs = be.upload(embed(rng.standard_normal((m2 // 2, K2 // 2)) + 1j * rng.standard_normal((m2 // 2, K2 // 2))))
    W = be.gemm(B, be.gemm(B, Xs, transA=True))
    al, _ = be.qrpos(W)                                        # embedded orthonormal basis of the kept subspace
    M = be.gemm(al, th, transA=True)                           # K2 x n2
    c, arm = be.lqpos(M)
    ar = be.empty(K2, d2, Dr2)
    for s2 in range(d2):                                       # arm[k, (b, s2)] -> ar[k, s2, b]
        be.copy2d(K2, Dr2, arm.ptr + 8 * s2 * K2 * Dr2, K2, ar.ptr + 8 * s2 * K2, K2 * d2)
    tot2 = float(np.sum(s * s))
    kept2 = be.norm(c) ** 2
    disc = float(np.sqrt(max(tot2 - kept2, 0.0) / 2.0))
    return al.reshape(Dl2, d1, K2), c, ar, s[0:K2:2].copy(), disc
