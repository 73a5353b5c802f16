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
dimension becomes 2.  Supported: FiniteMPS / InfiniteMPS with 1-site algorithms (DMRG, VUMPS, TDVP incl. real time,
calc_galerkin, expectation_value) and real MPO Hamiltonians; tsvd-based 2-site algorithms are not (singular vectors of the
embedding are only defined up to a rotation inside each doubled singular value)."""
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
