"""Native complex128 (MPSK_C128, the reference's default scalar type, defaults.jl:18) for the matvec / transfer family:
interleaved complex128 tensors through the C ABI against the oracle's complex arithmetic (derivatives.jl:77-193,
transfer.jl:18-126 with complex tensors; conj(Ab) on the bra).  Bar: 1e-12 relative."""
import numpy as np
import pytest

import mpskit_oracle as mo

pytestmark = pytest.mark.gpu
RTOL = 4e-13


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def rand_cslice(rng, odim, d, chis, density=0.6, scal_prob=0.3):
    blocks = {(0, 0): 1.0, (odim - 1, odim - 1): 1.0}
    for i in range(odim):
        for j in range(i, odim):
            if (i, j) in blocks:
                continue
            if rng.random() < density:
                if chis[i] == chis[j] and rng.random() < scal_prob:
                    blocks[(i, j)] = complex(rng.standard_normal(), rng.standard_normal())
                else:
                    blocks[(i, j)] = crand(rng, chis[i], d, d, chis[j])
    return mo.SparseMPOSlice(odim, d, chis, chis, blocks)


def dev_cslice(be, s):
    return be.mposlice(s.odim, s.d, s.chil, s.chir, dict(s.Os), cplx=True)


CASES = [
    (4, 4, 2, [1, 1, 1]),
    (16, 16, 2, [1, 1, 1, 1, 1]),
    (7, 13, 3, [1, 2, 1]),           # ragged, odd dims (unaligned kernels), chi > 1
    (64, 64, 2, [1, 1, 1, 1, 1]),
    (33, 65, 2, [1, 3, 2, 1]),
    (128, 96, 4, [1, 1, 1, 1, 1, 1]),
    (256, 256, 2, [1, 1, 1, 1, 1]),
    (1, 2, 2, [1, 1, 1]),            # chain edge
]


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_dAC_dC_complex(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(1000 + 7 * Dl + Dr)
    s = rand_cslice(rng, len(chis), d, chis)
    GL = [crand(rng, Dl, c, Dl) for c in chis]
    GR = [crand(rng, Dr, c, Dr) for c in chis]
    x = crand(rng, Dl, d, Dr)
    H, dGL, dGR = dev_cslice(be, s), be.upload_env_c(GL), be.upload_env_c(GR)
    ref = mo.dAC(x, s, GL, GR)
    y = be.download_c(be.dAC(H, dGL, dGR, be.upload_c(x)))
    assert relerr(y, ref) < RTOL * max(Dl, Dr)
    # prepared operator: planar right environment built once, applied twice
    hac = be.hac_create(H, dGL, dGR)
    assert hac.info()["mode"] == 2
    for _ in range(2):
        x = crand(rng, Dl, d, Dr)
        assert relerr(be.download_c(hac.apply(be.upload_c(x))), mo.dAC(x, s, GL, GR)) < RTOL * max(Dl, Dr)
    # dC with the same environments
    c = crand(rng, Dl, Dr)
    yc = be.download_c(be.dC(dGL, dGR, be.upload_c(c), cplx=True))
    assert relerr(yc, mo.dC(c, GL, GR)) < RTOL * max(Dl, Dr)
    # a complex slice whose entries happen to be real == the real slice applied to complex tensors
    sr = mo.SparseMPOSlice(len(chis), d, chis, chis, {k: (np.real(v) if np.isscalar(v) else np.real(v)) for k, v in s.Os.items()})
    yr = be.download_c(be.dAC(dev_cslice(be, sr), dGL, dGR, be.upload_c(x)))
    assert relerr(yr, mo.dAC(x, sr, GL, GR)) < RTOL * max(Dl, Dr)


@pytest.mark.parametrize("Dl,Dr,d,chis", [c for c in CASES if c[0] <= 128])
def test_dAC2_complex(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(2000 + Dl)
    s1, s2 = rand_cslice(rng, len(chis), d, chis), rand_cslice(rng, len(chis), d, chis)
    GL = [crand(rng, Dl, c, Dl) for c in chis]
    GR = [crand(rng, Dr, c, Dr) for c in chis]
    x2 = crand(rng, Dl, d, Dr, d)
    y = be.download_c(be.dAC2(dev_cslice(be, s1), dev_cslice(be, s2), be.upload_env_c(GL), be.upload_env_c(GR), be.upload_c(x2)))
    assert relerr(y, mo.dAC2(x2, s1, s2, GL, GR)) < RTOL * max(Dl, Dr)


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_transfer_left_right_complex(be, Dl, Dr, d, chis):
    """transfer.jl:105-110,166-259 with complex A, Abar (conjugated on the bra) and environments; bra and ket bond
    dimensions differ (Dlb = Dl + 1 ...) as in the excitation / overlap transfers."""
    rng = np.random.default_rng(3000 + Dl * 3 + Dr)
    s = rand_cslice(rng, len(chis), d, chis)
    Dlb, Drb = Dl + 1, Dr + 2
    A, Ab = crand(rng, Dl, d, Dr), crand(rng, Dlb, d, Drb)
    GLin = [crand(rng, Dlb, c, Dl) for c in chis]
    GRin = [crand(rng, Dr, c, Drb) for c in chis]
    H = dev_cslice(be, s)
    dA, dAb = be.upload_c(A), be.upload_c(Ab)
    tl = be.download_env_c(be.transfer_left(H, be.upload_env_c(GLin), dA, dAb), chis)
    for a, b in zip(tl, mo.transfer_left(GLin, s, A, Ab)):
        assert relerr(a, b) < RTOL * max(Dl, Dr) * 2
    tr = be.download_env_c(be.transfer_right(H, be.upload_env_c(GRin), dA, dAb), chis)
    for a, b in zip(tr, mo.transfer_right(GRin, s, A, Ab)):
        assert relerr(a, b) < RTOL * max(Dl, Dr) * 2
    # pass-through legs (H = NULL): W independent slabs  (transfer.jl:18-45,66-75)
    vl = [crand(rng, Dlb, 1, Dl) for _ in range(2)]
    vr = [crand(rng, Dr, 1, Drb) for _ in range(2)]
    pl = be.download_env_c(be.transfer_left(None, be.upload_env_c(vl), dA, dAb, cplx=True), [1, 1])
    pr = be.download_env_c(be.transfer_right(None, be.upload_env_c(vr), dA, dAb, cplx=True), [1, 1])
    for k in range(2):
        assert relerr(pl[k][:, 0, :], mo.transfer_left_bond(vl[k][:, 0, :], A, Ab)) < RTOL * max(Dl, Dr) * 2
        assert relerr(pr[k][:, 0, :], mo.transfer_right_bond(vr[k][:, 0, :], A, Ab)) < RTOL * max(Dl, Dr) * 2


def test_complex_effective_hamiltonian_is_hermitian(be):
    """Hermitian MPO + Hermitian-paired environments -> <u, H v> = conj(<v, H u>) (size-independent identity at D = 512)."""
    rng = np.random.default_rng(9)
    D, d = 512, 2
    Sz = np.diag([0.5, -0.5]).astype(complex)
    Sp = np.array([[0, 1], [0, 0]], dtype=complex)
    Sy = np.array([[0, -0.5j], [0.5j, 0]])
    blocks = {(0, 0): 1.0, (3, 3): 1.0, (0, 1): Sz[None, :, :, None], (1, 3): Sz[None, :, :, None],
              (0, 2): Sy[None, :, :, None], (2, 3): Sy[None, :, :, None], (0, 3): (0.3 * Sz + 0.2 * (Sp + Sp.T))[None, :, :, None]}
    H = be.mposlice(4, d, [1] * 4, [1] * 4, blocks, cplx=True)
    herm = lambda m: m + np.conj(np.transpose(m, (0, 2, 1)))
    gl = herm(crand(rng, 4, D, D))
    gr = herm(crand(rng, 4, D, D))
    GL = be.upload_env_c([m[:, None, :] for m in gl])
    GR = be.upload_env_c([m[:, None, :] for m in gr])
    u, v = crand(rng, D, d, D), crand(rng, D, d, D)
    hac = be.hac_create(H, GL, GR)
    Hu, Hv = be.download_c(hac.apply(be.upload_c(u))), be.download_c(hac.apply(be.upload_c(v)))
    a, b = np.vdot(u, Hv), np.vdot(Hu, v)
    assert abs(a - b) < 1e-11 * abs(a) * D


@pytest.mark.parametrize("name", ["test_complex_states_realtime_tdvp_and_dmrg", "test_complex_two_site_algorithms",
                                  "test_complex_infinite_mps_vumps"])
def test_drivers_on_complex_states_through_the_native_operators(be, monkeypatch, name):
    """The complex-state driver tests of test_gpu_algorithms.py (DMRG / TDVP / TDVP2 / DMRG2 / VUMPS vs the oracle's
    complex arithmetic, dense expm, energy conservation) with the effective Hamiltonians FORCED through the native
    MPSK_C128 kernels (cplx.HalfEmbeddedOp: encode -> Krylov on interleaved complex vectors -> decode); by default
    tensors this small take the embedded path (derivatives.NATIVE_CPLX_MIN_D)."""
    import test_gpu_algorithms as tga
    from mpskit_jl_amd import cplx
    monkeypatch.setenv("MPSK_NATIVE_CPLX", "1")
    calls = {"n": 0}
    orig = cplx.HalfEmbeddedOp.apply_half

    def counted(self, xh, out=None):
        calls["n"] += 1
        return orig(self, xh, out)

    monkeypatch.setattr(cplx.HalfEmbeddedOp, "apply_half", counted)
    getattr(tga, name)(be)
    assert calls["n"] > 10, "the native complex operator was not exercised"


def test_gauge_steps_keep_the_embedding_on_ill_conditioned_states(be, monkeypatch):
    """A random complex MPS with uniform[0,1) entries has strongly graded bonds (numerically rank deficient at D = 64):
    the plain real QRpos of the embedded tensors returns isometries that are NOT embeddings along the weak directions,
    and everything downstream leaves the complex manifold (before cplx.qrpos_structured: AL defects up to O(1), energy
    drift 3e-4 per real-time TDVP step at D = 512).  Now: AL / AR are embeddings to rounding, canonical identities hold,
    and a real-time TDVP step conserves energy and norm through the embedded AND the native operators."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import cplx
    L, D = 14, 64
    H = mk.heisenberg_XXX(0.5, be=be)
    for mode in ("1", "0"):
        monkeypatch.setenv("MPSK_NATIVE_CPLX", mode)
        psi = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
        for i in range(L):
            al, ar = be.download(psi.AL(i)), be.download(psi.AR(i))
            assert cplx.structure_defect(al) < 1e-11 and cplx.structure_defect(ar) < 1e-11, (i, cplx.structure_defect(al), cplx.structure_defect(ar))
            alc, arc = cplx.extract(al), cplx.extract(ar)
            assert np.abs(np.einsum("asb,asc->bc", alc.conj(), alc) - np.eye(alc.shape[2])).max() < 1e-11
            assert np.abs(np.einsum("asb,csb->ac", arc, arc.conj()) - np.eye(arc.shape[0])).max() < 1e-11
            ac = cplx.extract(be.download(psi.AC(i)))
            assert np.abs(np.einsum("asb,bk->ask", alc, cplx.extract(be.download(psi.CR(i)))) - ac).max() < 1e-11
        envs = mk.FinEnv(psi, H)
        e0 = float(np.sum(mk.expectation_value(psi, H, envs)))
        for k in range(2):
            psi, envs = mk.timestep(psi, H, 0.05 * k, 0.05, mk.TDVP(tol=1e-11), envs)
        e1 = float(np.sum(mk.expectation_value(psi, H, envs)))
        assert abs(e1 - e0) < 2e-9 * max(1.0, abs(e0)) and abs(psi.norm() - 1) < 1e-9, (mode, e1 - e0, psi.norm() - 1)


def test_complex_states_in_changebonds_and_finite_excited_gpu(be):
    """changebonds (OptimalExpand / SvdCut) and excitations(H, FiniteExcited(), psi) on COMPLEX states through the C ABI
    (same body as the host-logic test on the CPU stand-in: tests/test_host_logic_cpu.py)."""
    from test_host_logic_cpu import test_complex_states_in_changebonds_and_finite_excited as body
    body(be)


@pytest.mark.parametrize("m,n,k", [(192, 160, 40), (160, 192, 40), (256, 256, 64)])
def test_structured_split_through_the_truncation_aware_tsplit(be, m, n, k):
    """cplx.split_two_site at sizes where it takes mpsk_tsplit (2 k + 16 leading vectors; svd mode 3 runs its subspace
    stage on the doubled spectrum of the embedding) instead of the full mpsk_tsvd: al / ar are EMBEDDED isometries, the kept
    Schmidt values are the complex singular values, al c ar is the optimal rank-k truncation (numpy complex SVD); and a
    degenerate pair of complex singular values across the cut (a 4-fold cluster in the embedding) still gives a structured,
    optimal split."""
    from mpskit_jl_amd import cplx
    rng = np.random.default_rng(m + n)
    r = min(m, n)
    for degenerate in (False, True):
        U, _ = np.linalg.qr(rng.standard_normal((m, r)) + 1j * rng.standard_normal((m, r)))
        V, _ = np.linalg.qr(rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r)))
        sv = np.logspace(0, -5, r)
        if degenerate:
            sv[k] = sv[k - 1]
        th = (U * sv) @ V.conj().T
        E = cplx.embed(th).reshape(2 * m, 1, 2 * n, 1)              # theta_E[(2 Dl), d1, (2 Dr), d2] with d1 = d2 = 1
        al, c, ar, S, disc = cplx.split_two_site(be, be.upload(E), trunc_dim=k)
        assert be.split_stats()["path"] in (0, 1)
        A, Cm, B = be.download(al), be.download(c), be.download(ar)
        K2 = 2 * k
        assert A.shape == (2 * m, 1, K2) and B.shape == (K2, 1, 2 * n)
        A2, B2 = A.reshape(2 * m, K2), B.reshape(K2, 2 * n)
        assert np.abs(A2.T @ A2 - np.eye(K2)).max() < 1e-12 and np.abs(B2 @ B2.T - np.eye(K2)).max() < 1e-12
        assert cplx.structure_defect(A2) < 1e-12 and cplx.structure_defect(B2) < 1e-12 and cplx.structure_defect(Cm) < 1e-12
        assert np.abs(S - sv[:k]).max() < 1e-12
        rec = cplx.extract(A2 @ Cm @ B2)
        best_err = np.linalg.norm(sv[k:])
        assert abs(np.linalg.norm(th - rec) - best_err) < 1e-11
        assert abs(disc - best_err) < 1e-11


@pytest.mark.parametrize("m,n", [(5, 3), (64, 64), (200, 130), (1024, 512)])
def test_qrpos_lqpos_complex128_through_the_abi(be, m, n):
    """mpsk_qrpos / mpsk_lqpos under MPSK_C128 (interleaved complex operands; orthoview.jl:49-60 on ComplexF64 tensors)
    against the oracle's complex QRpos / LQpos: Q^H Q = I, R upper triangular with a REAL positive diagonal, A = Q R, and
    Q, R equal to the oracle's (the factorization is unique) to 1e-12; the same for a 1e-10-conditioned matrix and an
    exactly rank-deficient one (factors compared through A = Q R and the isometry only: the completion is arbitrary)."""
    rng = np.random.default_rng(m * n)
    for kind in ("random", "graded", "rankdef"):
        A = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
        if kind == "graded":
            U, _ = np.linalg.qr(A)
            V, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
            A = (U * np.logspace(0, -10, n)) @ V.conj().T
        elif kind == "rankdef" and n >= 3:
            A[:, n - 1] = A[:, 0] * (0.3 - 0.2j) + A[:, 1]
        Q, R = be.qrpos_c(be.upload_c(A))
        Q, R = be.download_c(Q), be.download_c(R)
        assert Q.shape == (m, n) and R.shape == (n, n)
        assert np.abs(Q.conj().T @ Q - np.eye(n)).max() < 1e-12
        assert np.abs(np.tril(R, -1)).max() == 0.0 and np.abs(np.diag(R).imag).max() == 0.0
        assert np.all(np.diag(R).real >= -1e-13 * np.abs(R).max())          # (a vanishing pivot of rank-deficient input: 0 to rounding)
        assert np.abs(Q @ R - A).max() < 1e-12 * np.abs(A).max() * n
        if kind == "random":
            Qo, Ro = mo.qrpos(A)
            assert np.abs(Q - Qo).max() < 1e-12 and np.abs(R - Ro).max() < 1e-12 * np.abs(Ro).max()
        # LQpos of the conjugate transpose problem (n x m, wide)
        B = A.conj().T.copy()
        L, Ql = be.lqpos_c(be.upload_c(B))
        L, Ql = be.download_c(L), be.download_c(Ql)
        assert L.shape == (n, n) and Ql.shape == (n, m)
        assert np.abs(Ql @ Ql.conj().T - np.eye(n)).max() < 1e-12
        assert np.abs(np.triu(L, 1)).max() == 0.0 and np.abs(np.diag(L).imag).max() == 0.0
        assert np.abs(L @ Ql - B).max() < 1e-12 * np.abs(B).max() * n
        if kind == "random":
            Lo, Qlo = mo.lqpos(B)
            assert np.abs(L - Lo).max() < 1e-12 * np.abs(Lo).max() and np.abs(Ql - Qlo).max() < 1e-12


@pytest.mark.parametrize("m,n,k", [(192, 160, 40), (160, 192, 40), (256, 256, 64), (96, 80, 0), (512, 512, 128), (20, 16, 6),
                                   (24, 24, 12)])
def test_tsplit_complex128_through_the_abi(be, m, n, k):
    """mpsk_tsplit under MPSK_C128 (tsvd!(theta; trunc = truncdim(k)) of a ComplexF64 two-site tensor, dmrg.jl:96-104):
    al / ar complex isometries, c lower triangular with a real positive diagonal, the kept values are the complex singular
    values, al c ar is the optimal rank-k truncation (numpy complex SVD) -- also with a degenerate pair of complex values
    across the cut (a 4-fold cluster of the real embedding the library works on)."""
    rng = np.random.default_rng(m + n + k)
    r = min(m, n)
    kk = k if k > 0 else r
    for degenerate in (False, True):
        U, _ = np.linalg.qr(rng.standard_normal((m, r)) + 1j * rng.standard_normal((m, r)))
        V, _ = np.linalg.qr(rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r)))
        sv = np.logspace(0, -5, r)
        if degenerate and kk < r:
            sv[kk] = sv[kk - 1]
        th = (U * sv) @ V.conj().T
        al, c, ar, S, disc = be.tsplit_c(be.upload_c(th), max_keep=k)
        A, Cm, B = be.download_c(al), be.download_c(c), be.download_c(ar)
        assert A.shape == (m, kk) and Cm.shape == (kk, kk) and B.shape == (kk, n) and len(S) == kk
        assert np.abs(A.conj().T @ A - np.eye(kk)).max() < 1e-12 and np.abs(B @ B.conj().T - np.eye(kk)).max() < 1e-12
        assert np.abs(np.triu(Cm, 1)).max() == 0.0 and np.abs(np.diag(Cm).imag).max() == 0.0
        assert np.abs(S - sv[:kk]).max() < 1e-12
        best = np.linalg.norm(sv[kk:])
        assert abs(np.linalg.norm(th - A @ Cm @ B) - best) < 1e-11
        assert abs(disc - best) < 1e-7 + 1e-11 * (best > 1e-6)      # (|theta|^2 - |c|^2 cancels for tiny discarded weights)
        assert np.abs(np.linalg.svd(Cm, compute_uv=False) - sv[:kk]).max() < 1e-12


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(5, 3, 4), (130, 70, 33), (512, 256, 256)])
def test_gemm_complex128_through_the_abi(be, M, N, K, tA, tB):
    """mpsk_gemm under MPSK_C128: C = alpha op(A) op(B) + beta C on interleaved complex matrices, op = conjugate transpose
    (AC = AL*C, AC = C*AR, AL = Q_AC*Q_C' on ComplexF64 tensors: orthoview.jl:99,103, ortho.jl:130) == numpy."""
    rng = np.random.default_rng(M + N + K + 2 * tA + tB)
    cr = lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)
    A = cr(K, M) if tA else cr(M, K)
    B = cr(N, K) if tB else cr(K, N)
    C0 = cr(M, N)
    ref = 0.7 * (A.conj().T if tA else A) @ (B.conj().T if tB else B) - 1.3 * C0
    out = be.upload_c(C0)
    be.gemm_c(be.upload_c(A), be.upload_c(B), transA=bool(tA), transB=bool(tB), alpha=0.7, beta=-1.3, out=out)
    assert np.abs(be.download_c(out) - ref).max() < 1e-13 * K * max(1.0, np.abs(ref).max())


def _vec(tensors):
    """state vector of an open-boundary MPS given as [Dl, d, Dr] tensors (any gauge)"""
    v = tensors[0]
    for t in tensors[1:]:
        v = np.tensordot(v, t, axes=([-1], [0]))
    return v.reshape(-1)


def test_native_interleaved_dmrg_and_tdvp_match_the_oracle(be):
    """native_cplx: complex128 states on INTERLEAVED storage, every step through the MPSK_C128 entry points of the C ABI
    (prepared operator, mpsk_dC, transfers, mpsk_qrpos / mpsk_lqpos / mpsk_gemm complex).  One-site DMRG sweeps follow the
    oracle's complex128 run sweep by sweep (dmrg.jl:22-55); a real-time TDVP step (tdvp.jl:61-94) lands on the oracle's
    state (overlap 1 - 1e-10, energy conserved); the state takes 2x the memory of a real one (the embedded host: 4x)."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import native_cplx as nc
    rng = np.random.default_rng(5)
    L, d, D = 8, 2, 12
    dims = mo.FiniteMPS.random(L, d, D, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    H = mk.heisenberg_XXX(0.5, be=be)
    Ho = mo.heisenberg_mpo(0.5)
    psi = nc.NativeFiniteMPS(As, be)
    assert psi.bytes() == 16 * sum(int(np.prod(a.shape)) for a in As)
    assert abs(psi.norm() - 1.0) < 1e-13
    # the right-canonical form built through mpsk_lqpos / mpsk_gemm (C128) is the input state
    v0 = _vec(As); v0 /= np.linalg.norm(v0)
    assert abs(abs(np.vdot(v0, _vec(psi.to_host()))) - 1.0) < 1e-12
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(As, normalize=True)
    eo = mo.FinEnv(po, Ho)
    assert abs(nc.energy(psi, envs) - float(np.real(np.sum(mo.expectation_value(po, Ho, eo))))) < 1e-11
    eig = mk.Arnoldi(tol=1e-12, krylovdim=20, maxiter=50)
    for sweep in range(3):
        E = nc.dmrg_sweep(psi, H, envs, eig)
        po, eo, _, log = mo.dmrg(po, Ho, maxiter=1, eig_tol=1e-12, krylovdim=20, eig_maxiter=50, envs=None)
        eo = mo.FinEnv(po, Ho)
        assert abs(E - log[-1][1]) < 1e-9 * abs(E), (sweep, E, log[-1][1])
    E0 = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))[0]
    assert E0 - 1e-9 <= E <= E0 + 1e-3 * abs(E0)
    # real-time TDVP step from a fresh random complex state
    psi = nc.NativeFiniteMPS(As, be)
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(As, normalize=True)
    e_before = nc.energy(psi, envs)
    alg = mk.TDVP(tol=1e-12, krylovdim=20)
    psi, envs = nc.tdvp_step(psi, H, envs, 0.0, 0.05, alg)
    po2, _ = mo.tdvp_timestep(po, Ho, 0.0, 0.05, tol=1e-12, krylovdim=20)
    vo = mo.mps_to_vector(po2)
    vn = _vec(psi.to_host())
    assert abs(np.linalg.norm(vn) - 1.0) < 1e-10
    assert abs(abs(np.vdot(vo, vn)) - 1.0) < 1e-9
    assert abs(nc.energy(psi, envs) - e_before) < 1e-8


def test_fixed_budget_sweeps_on_a_converged_complex_state_stay_variational_gpu(be):
    """The GPU twin of tests/test_host_logic_cpu.py::test_fixed_budget_sweeps_on_a_converged_complex_state_stay_variational
    at the size where it was found (L = 16, D = 64, 8 matvecs per site: -7.13 against a converged -6.9117 before the Krylov
    solvers of embedded complex states moved to half-embedded vectors), on the embedded host state AND the interleaved one."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import algorithms as alg, krylov, native_cplx as nc
    L, D = 16, 64
    H = mk.heisenberg_XXX(0.5, be=be)
    ref = mk.FiniteMPS.random(L, 2, D, np.random.default_rng(5), be=be, dtype=complex)
    As = [ref.download(ref.AL(i)) for i in range(L - 1)] + [ref.download(ref.AC(L - 1))]
    ws = krylov.KrylovWorkspace(be)
    pe = mk.FiniteMPS(As, normalize=True, be=be); ee = mk.FinEnv(pe, H)
    pn = nc.NativeFiniteMPS(As, be); en = nc.NativeFinEnv(pn, H)
    tol = mk.Arnoldi(tol=1e-12, krylovdim=30, maxiter=100)
    for _ in range(2):
        alg.dmrg_sweep(pe, H, ee, tol, ws)
        En = nc.dmrg_sweep(pn, H, en, tol, ws)
    Ec = float(np.sum(mk.expectation_value(pe, H, ee)))
    assert abs(Ec - En) < 1e-10
    fixed = mk.Arnoldi(fixed_matvecs=8, krylovdim=8)
    for _ in range(2):
        alg.dmrg_sweep(pe, H, ee, fixed, ws)
        En = nc.dmrg_sweep(pn, H, en, fixed, ws)
        Ee = float(np.sum(mk.expectation_value(pe, H, ee)))
        assert abs(Ee - Ec) < 1e-8 and abs(En - Ec) < 1e-8, (Ee, En, Ec)


def test_native_interleaved_two_site_dmrg_matches_the_oracle(be):
    """native_cplx.dmrg2_sweep (dmrg.jl:86-120 on interleaved storage: mpsk_dAC2 complex + mpsk_tsplit / mpsk_gemm under
    MPSK_C128) follows the oracle's complex128 DMRG2 sweep by sweep, from bond dimension 4 up to the truncation bound."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import native_cplx as nc
    rng = np.random.default_rng(8)
    L, d, D0, D = 8, 2, 4, 12
    dims = mo.FiniteMPS.random(L, d, D0, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    H, Ho = mk.heisenberg_XXX(0.5, be=be), mo.heisenberg_mpo(0.5)
    psi = nc.NativeFiniteMPS(As, be)
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(As, normalize=True)
    eig = mk.Arnoldi(tol=1e-12, krylovdim=20, maxiter=50)
    for sweep in range(3):
        E = nc.dmrg2_sweep(psi, H, envs, eig, trunc_dim=D)
        po, _, _, log = mo.dmrg2(po, Ho, truncdim=D, maxiter=1, eig_tol=1e-12, krylovdim=20, eig_maxiter=50)
        assert abs(E - log[-1][1]) < 1e-9 * abs(E), (sweep, E, log[-1][1])
    assert max(psi.dims(i)[2] for i in range(L - 1)) == D
    E0 = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))[0]
    assert E0 - 1e-9 <= E <= E0 + 1e-4 * abs(E0)
    # the reference-named entry point: DMRG then DMRG2 objects drive the same sweeps to convergence
    psi2 = nc.NativeFiniteMPS(As, be)
    psi2, envs2, dE = nc.find_groundstate(psi2, H, mk.DMRG2(tol=1e-10, maxiter=8, trunc_dim=D, eigalg=eig))
    psi2, envs2, dE = nc.find_groundstate(psi2, H, mk.DMRG(tol=1e-11, maxiter=8, eigalg=eig), envs2)
    assert dE <= 1e-8 and abs(nc.energy(psi2, envs2) - E0) < 1e-4 * abs(E0)
    psi2, envs2 = nc.timestep(psi2, H, 0.0, 0.02, mk.TDVP(tol=1e-10), envs2)
    assert abs(psi2.norm() - 1.0) < 1e-9


def test_native_interleaved_tdvp2_matches_the_oracle(be):
    """native_cplx.tdvp2_step (tdvp.jl:113-146 on interleaved storage) against the oracle's complex TDVP2 step: same state
    (overlap 1 - 1e-9) with the bond dimension growing from 4 to the truncation bound, energy conserved to the truncation."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import native_cplx as nc
    rng = np.random.default_rng(12)
    L, d, D0, D = 8, 2, 4, 16
    dims = mo.FiniteMPS.random(L, d, D0, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    H, Ho = mk.heisenberg_XXX(0.5, be=be), mo.heisenberg_mpo(0.5)
    psi = nc.NativeFiniteMPS(As, be)
    envs = nc.NativeFinEnv(psi, H)
    e0 = nc.energy(psi, envs)
    po = mo.FiniteMPS(As, normalize=True)
    alg = mk.TDVP2(tol=1e-12, krylovdim=20, trunc_dim=D)
    psi, envs = nc.tdvp2_step(psi, H, envs, 0.0, 0.05, alg, trunc_dim=D)
    po2, _ = mo.tdvp2_timestep(po, Ho, 0.0, 0.05, truncdim=D, tol=1e-12, krylovdim=20)
    vo, vn = mo.mps_to_vector(po2), _vec(psi.to_host())
    assert abs(abs(np.vdot(vo, vn)) / (np.linalg.norm(vo) * np.linalg.norm(vn)) - 1.0) < 1e-9
    assert abs(np.linalg.norm(vn) - np.linalg.norm(vo)) < 1e-9
    assert max(psi.dims(i)[2] for i in range(L - 1)) > D0
    assert abs(nc.energy(psi, envs) - e0) < 1e-6


def test_native_interleaved_dmrg_with_a_complex_mpo(be):
    """A genuinely complex MPOHamiltonian -- the Pauli-matrix XXX chain of the reference's docs (operators.md:67-78: sigma^y is
    complex) -- through native_cplx.ComplexMPOHamiltonian (MPSK_C128 slices): one-site DMRG follows the oracle's complex run
    sweep by sweep and reaches the ED ground energy."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import native_cplx as nc
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    Y = np.array([[0, -1j], [1j, 0]])
    Z = np.array([[1, 0], [0, -1]], dtype=complex)
    H = nc.ComplexMPOHamiltonian({(0, 0): 1.0, (4, 4): 1.0, (0, 1): X, (1, 4): X, (0, 2): Y, (2, 4): Y, (0, 3): Z, (3, 4): Z}, be)
    Ho = mo.heisenberg_pauli_mpo()
    rng = np.random.default_rng(17)
    L, d, D = 8, 2, 16
    dims = mo.FiniteMPS.random(L, d, D, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    psi = nc.NativeFiniteMPS(As, be)
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(As, normalize=True)
    eig = mk.Arnoldi(tol=1e-12, krylovdim=20, maxiter=50)
    for sweep in range(3):
        E = nc.dmrg_sweep(psi, H, envs, eig)
        po, _, _, log = mo.dmrg(po, Ho, maxiter=1, eig_tol=1e-12, krylovdim=20, eig_maxiter=50)
        assert abs(E - log[-1][1]) < 1e-9 * abs(E), (sweep, E, log[-1][1])
    E0 = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))[0]
    assert abs(E - E0) < 1e-8 * abs(E0)          # D = 16 is the exact bond dimension of L = 8
