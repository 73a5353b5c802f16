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


class HalfEmbeddedOp:
    """Effective Hamiltonian of an EMBEDDED complex state applied through the NATIVE complex128 kernels (MPSK_C128).

    The state tensors and environments of a complex FiniteMPS / InfiniteMPS are stored embedded (module docstring);
    inside a Krylov solve the iterate only needs ONE column of every 2x2 block -- E[:, ..., 2b] -- and that
    "half-embedded" tensor (2 Dl, d, Dr) IS the interleaved complex128 tensor of the C ABI (include/mpsk.h).  So:
      prepare (once per site visit): GLc / GRc = even columns of the embedded environment slabs (interleaved complex
              environments), the complex twin of the real MPO slice, the prepared operator (mpsk_hac_create);
      encode / decode (once per solve): embedded tensor <-> even columns;  decode rebuilds [h | J h];
      apply: mpsk_hac_apply / mpsk_dC / mpsk_dAC2 on complex operands -- 4x the real flops instead of the 8x of the
             embedded matvec, half the vector length in every Krylov vector operation.
    `h(x)` with an embedded x encodes / decodes around the call, so callers that do not know the protocol still work;
    solvers that do (dmrg_sweep, integrate) call encode once, iterate on half vectors, decode once."""

    def __init__(self, be, kind, slices, GL_E: DTensor, GR_E: DTensor):
        self.be, self.kind = be, kind
        self.slices = [self._cslice(be, h) for h in slices]
        W, Dlo2, Dl2 = GL_E.shape
        Wr, Dr2, _ = GR_E.shape
        self.Dlo, self.Dl, self.Dr = Dlo2 // 2, Dl2 // 2, Dr2 // 2
        # structured part of every slab (see _structured_half): (Dlo2 x W * Dl2) matrix -> (Dlo2 x W * Dl)
        self.GLc = self._structured_half(be, GL_E.ptr, Dlo2, W * self.Dl).reshape(W, Dlo2, self.Dl)
        self.GRc = self._structured_half(be, GR_E.ptr, Dr2, Wr * self.Dr).reshape(Wr, Dr2, self.Dr)
        self._hac = be.hac_create(self.slices[0], self.GLc, self.GRc) if kind == "AC" else None
        self._keep = (GL_E, GR_E)

    @staticmethod
    def _structured_half(be, src_ptr, rows, ncols, out=None):
        """Interleaved complex matrix (rows x ncols, rows = 2 x complex rows) = the STRUCTURED part of the embedded real
        matrix at src_ptr (rows x 2 ncols): with E = [e_0 | e_1] the even / odd columns of a 2x2-block column, an exact
        embedding has e_1 = J e_0, and  h = (e_0 - J e_1) / 2  is the complex column whose embedding is closest to E
        (r = (E00 + E11) / 2, m = (E10 - E01) / 2).  For an exact embedding h = e_0.  Gauge steps on numerically
        rank-deficient tensors (the perturbed CholeskyQR / Householder completion of QRpos) return isometries whose
        columns in the NULL directions are arbitrary, unstructured vectors; they carry no weight in the state, but the
        environments built from them are no embeddings there.  Taking e_0 alone made the native operator non-Hermitian
        in those directions (real-time TDVP at D = 512 from a random state: energy drift 3e-4, norm 0.9995); the
        structured part of a symmetric environment is Hermitian, so the evolution stays unitary."""
        if hasattr(be, "cx_half_raw"):               # one launch on the device backend
            ev = be.empty(rows, ncols) if out is None else out
            be.cx_half_raw(rows // 2, ncols, src_ptr, rows, ev.ptr, rows)
            return ev
        ev = be.empty(rows, ncols)
        od = be.empty(rows, ncols)
        be.copy2d(rows, ncols, src_ptr, 2 * rows, ev.ptr, rows)                   # e_0 columns
        be.copy2d(rows, ncols, src_ptr + 8 * rows, 2 * rows, od.ptr, rows)        # e_1 columns
        jod = be.times_i(od)                                                      # J e_1
        be.axpby(-0.5, jod, 0.5, ev)                                              # (e_0 - J e_1) / 2
        if out is not None:
            be.axpby(1.0, ev, 0.0, out)
            return out
        return ev

    @staticmethod
    def _cslice(be, H):
        c = getattr(H, "_cslice", None)
        if c is None:
            c = be.mposlice(H.odim, H.d, H.chil, H.chir, H.blocks, cplx=True)
            H._cslice = c
        return c

    # ---- layout: the doubled ket bond is the 3rd index (AC, AC2) or the 2nd (C); AC2 has a trailing physical index
    def _geom(self, shape):
        if self.kind == "C":
            return shape[0], shape[1], 1
        if self.kind == "AC":
            return shape[0] * shape[1], shape[2], 1
        return shape[0] * shape[1], shape[2], shape[3]

    def _half_shape(self, shape):
        k = 1 if self.kind == "C" else 2
        return tuple(shape[:k]) + (shape[k] // 2,) + tuple(shape[k + 1:])

    def _full_shape(self, shape):
        k = 1 if self.kind == "C" else 2
        return tuple(shape[:k]) + (shape[k] * 2,) + tuple(shape[k + 1:])

    def is_half(self, x: DTensor):
        k = 1 if self.kind == "C" else 2
        return x.shape[k] == self.Dr

    def encode(self, x: DTensor, out: DTensor = None):
        be = self.be
        rows, n2, tail = self._geom(x.shape)
        out = be.empty(*self._half_shape(x.shape)) if out is None else out
        n = n2 // 2
        for j in range(tail):
            self._structured_half(be, x.ptr + 8 * j * rows * n2, rows, n,
                                  out=DTensor(out.buf[j * rows * n:(j + 1) * rows * n], (rows, n)))
        return out

    def decode(self, xh: DTensor, out: DTensor = None):
        be = self.be
        full = self._full_shape(xh.shape)
        rows, n2, tail = self._geom(full)
        out = be.empty(*full) if out is None else out
        n = n2 // 2
        if hasattr(be, "cx_embed_raw"):                  # one launch per trailing index on the device backend
            for j in range(tail):
                be.cx_embed_raw(rows // 2, n, xh.ptr + 8 * j * rows * n, rows, out.ptr + 8 * j * rows * n2, rows)
            return out
        jx = be.times_i(xh)                              # J x: (re, im) -> (-im, re) on every row pair
        for j in range(tail):
            be.copy2d(rows, n, xh.ptr + 8 * j * rows * n, rows, out.ptr + 8 * j * rows * n2, 2 * rows)
            be.copy2d(rows, n, jx.ptr + 8 * j * rows * n, rows, out.ptr + 8 * (j * rows * n2 + rows), 2 * rows)
        return out

    def apply_half(self, xh: DTensor, out: DTensor = None):
        be = self.be
        if self.kind == "AC":
            return self._hac.apply(xh, out=out)
        if self.kind == "C":
            return be.dC(self.GLc, self.GRc, xh, out=out, cplx=True)
        return be.dAC2(self.slices[0], self.slices[1], self.GLc, self.GRc, xh, out=out)

    def __call__(self, x: DTensor, out: DTensor = None):
        if self.is_half(x):
            return self.apply_half(x, out)
        y = self.decode(self.apply_half(self.encode(x)))
        if out is not None:
            self.be.axpby(1.0, y, 0.0, out)
            return out
        return y

    __mul__ = __call__


class HalfSpaceOp(HalfEmbeddedOp):
    """Any operator on EMBEDDED tensors (op(x_E) -> y_E), iterated on the half-embedded (= interleaved complex) vectors:
    apply = encode(op(decode(h))).  Needed whenever the operator is not the same in both sectors of the doubled real space:
    besides the embeddings X_E, the (2 Dl) x d x (2 Dr) real tensors contain the "anti-structured" X_E Z (Z = sigma_z on
    every right-bond pair), on which an embedded operator acts as its PARTIAL complex conjugate.  A Krylov solve on the
    embedded vectors relies on rounding never populating that sector; a term that lives in the embedded sector only -- the
    projector penalty w |v><v| of excitations(H, FiniteExcited(), psi) -- leaves a LOWER eigenvalue there (the unpenalised
    ground state) and the solve converges to it (measured: the "excited" state came back with the ground energy and
    unstructured tensors).  On half vectors the other sector does not exist.  Real inner products of half vectors are
    Re <x, y>, all a Hermitian Lanczos / Arnoldi solve needs."""

    def __init__(self, be, op, kind, Dr):
        self.be, self.op, self.kind, self.Dr = be, op, kind, int(Dr)

    def apply_half(self, xh: DTensor, out: DTensor = None):
        y = self.encode(self.op(self.decode(xh)))
        if out is not None:
            self.be.axpby(1.0, y, 0.0, out)
            return out
        return y


def structured_part(be, E: DTensor):
    """P(E): the embedded real matrix closest to E (2r x 2c) among embeddings of complex matrices:
    r = (E00 + E11) / 2, m = (E10 - E01) / 2 on every 2x2 block.  Returns (P(E), |E - P(E)|_F)."""
    rows, cols2 = E.shape
    n = cols2 // 2
    h = HalfEmbeddedOp._structured_half(be, E.ptr, rows, n)
    jh = be.times_i(h)
    out = be.empty(rows, cols2)
    be.copy2d(rows, n, h.ptr, rows, out.ptr, 2 * rows)
    be.copy2d(rows, n, jh.ptr, rows, out.ptr + 8 * rows, 2 * rows)
    diff = be.copy(out)
    be.axpby(-1.0, E, 1.0, diff)
    return out, be.norm(diff)


STRUCT_TOL = 1e-13


def qrpos_structured(be, A: DTensor):
    """QRpos of an EMBEDDED complex matrix that returns EMBEDDED factors also when A is ill-conditioned or rank
    deficient.  In exact arithmetic the real QRpos of an embedding is the embedding of the complex QRpos; numerically
    the columns of Q along weakly determined directions (sigma_j << sigma_max: relative perturbation u sigma_max / sigma_j,
    arbitrary for null directions -- the perturbed CholeskyQR / Householder completion) are NOT embeddings, and
    everything built from such an isometry (environments, the tangent-space projector of TDVP) leaves the complex
    manifold: a random complex D = 512 state drifted by 3e-4 in energy per real-time step before this.  Remedy:
    Q_s = P(Q) (structured part), re-orthonormalised by a QRpos that is now well conditioned (and therefore structure
    preserving to rounding), R = triu(Q^T A).  The weakly determined directions carry weight sigma_j, so
    |A - Q R| stays O(u sigma_max): backward stable.  No-op (one projection, one norm) when Q is already structured."""
    Q, R = be.qrpos(A)
    Qs, defect = structured_part(be, Q)
    if defect <= STRUCT_TOL * np.sqrt(Q.shape[1]):
        return Q, R
    Q2, _ = be.qrpos(Qs)
    Q2, d2 = structured_part(be, Q2)          # rounding-level clean-up (d2 ~ 1e-15)
    Rn = be.gemm(Q2, A, transA=True)
    be.triu_(Rn)
    return Q2, Rn


def lqpos_structured(be, A: DTensor):
    """LQpos twin of qrpos_structured (A = L Q, Q with orthonormal rows)."""
    L, Q = be.lqpos(A)
    Qs, defect = structured_part(be, Q)
    if defect <= STRUCT_TOL * np.sqrt(Q.shape[0]):
        return L, Q
    _, Q2 = be.lqpos(Qs)
    Q2, d2 = structured_part(be, Q2)
    Ln = be.gemm(A, Q2, transB=True)
    be.tril_(Ln)
    return Ln, Q2


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
    fast = None
    PAD = 16                                   # singular values / vectors looked at beyond the cut (cluster detection)
    if trunc_dim > 0 and trunc_err == 0.0 and min(m2, n2) > 64 and 2 * trunc_dim + PAD <= min(m2, n2):
        # truncdim scheme: the truncation-aware two-site split (mpsk_tsplit, svd mode 3: subspace iteration + Jacobi on
        # ~1.5 k columns, checked) delivers the 2 k + PAD leading left singular vectors and values -- all this routine
        # needs -- instead of the full decomposition with accumulated rotations (mpsk_tsvd)
        U, _, _, s, _ = be.tsplit(th, max_keep=2 * trunc_dim + PAD)
        kmax, K2 = len(s), 2 * trunc_dim
        fast = be.norm(th) ** 2                # |theta|^2 = sum of ALL squared singular values
    if fast is None:
        U, S, Vh, kept, _ = be.tsvd(th, max_keep=2 * trunc_dim if trunc_dim > 0 else 0, trunc_err=trunc_err)
        s = be.download(S)
        kmax = len(s)
        K2 = min(kept + (kept & 1), kmax - (kmax & 1))
    def cluster(s, K2, kmax):
        """[lo, hi): the (even-aligned) run of singular values equal to s[K2 - 1] to 1e-8 -- the cluster the cut may split"""
        tol_c = 1e-8
        lo = K2
        while lo > 0 and abs(s[lo - 1] - s[K2 - 1]) <= tol_c * s[K2 - 1] + 1e-14 * s[0]:
            lo -= 1
        hi = K2
        while hi < kmax and abs(s[hi] - s[K2 - 1]) <= tol_c * s[K2 - 1] + 1e-14 * s[0]:
            hi += 1
        return lo - (lo & 1), min(hi + (hi & 1), kmax)

    lo, hi = cluster(s, K2, kmax)
    if fast is not None and hi >= kmax and kmax < min(m2, n2):
        # the cluster at the cut runs past the vectors the fast path computed: take the full decomposition instead
        U, S, Vh, kept, _ = be.tsvd(th, max_keep=2 * trunc_dim, trunc_err=0.0)
        s = be.download(S)
        kmax = len(s)
        K2 = min(kept + (kept & 1), kmax - (kmax & 1))
        fast = None
        lo, hi = cluster(s, K2, kmax)
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
    tot2 = float(np.sum(s * s)) if fast is None else fast
    kept2 = be.norm(c) ** 2
    disc = float(np.sqrt(max(tot2 - kept2, 0.0) / 2.0))
    return al.reshape(Dl2, d1, K2), c, ar, s[0:K2:2].copy(), disc
