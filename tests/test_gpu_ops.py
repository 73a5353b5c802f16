"""GPU parity of the hot-path operators (through the C ABI) against the NumPy oracle.

Tolerance: fp64, |y_gpu - y_oracle| <= 1e-12 * ||y_oracle||_inf-scale (north star: energies to
1e-10 relative; the operators themselves are held to ~100 ulp of the accumulated magnitude)."""
import numpy as np
import pytest

import mpskit_oracle as mo

pytestmark = pytest.mark.gpu

RTOL = 2e-13


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rand_slice(rng, odim, d, chis, density=0.6, scal_prob=0.3):
    """random block-sparse slice with the MPOHamiltonian structure (upper triangular, 1 on corners)."""
    blocks = {(0, 0): 1.0, (odim - 1, odim - 1): 1.0}
    for i in range(odim):
        for j in range(i, odim):
            if (i, j) in blocks:
                continue
            if rng.random() < density:
                if chis[i] == chis[j] and rng.random() < scal_prob:
                    blocks[(i, j)] = float(rng.standard_normal())
                else:
                    blocks[(i, j)] = rng.standard_normal((chis[i], d, d, chis[j]))
    return mo.SparseMPOSlice(odim, d, chis, chis, blocks)


def dev_slice(be, s):
    return be.mposlice(s.odim, s.d, s.chil, s.chir, dict(s.Os))


def rand_env(rng, chis, Db, Dk):
    return [rng.standard_normal((Db, c, Dk)) for c in chis]


CASES = [
    # (Dl, Dr, d, chis)
    (4, 4, 2, [1, 1, 1]),
    (16, 16, 2, [1, 1, 1, 1, 1]),
    (7, 13, 3, [1, 2, 1]),          # ragged, chi > 1
    (64, 64, 2, [1, 1, 1, 1, 1]),
    (33, 65, 2, [1, 3, 2, 1]),
    (128, 128, 4, [1, 1, 1, 1, 1, 1]),
    (256, 256, 3, [1, 1, 1, 1, 1]),  # config 2 shape
    (1, 2, 2, [1, 1, 1]),            # chain edge
]


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_dAC(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(20240213 + Dl * 7 + Dr)
    s = rand_slice(rng, len(chis), d, chis)
    GL, GR = rand_env(rng, chis, Dl, Dl), rand_env(rng, chis, Dr, Dr)
    x = rng.standard_normal((Dl, d, Dr))
    ref = mo.dAC(x, s, GL, GR)
    y = be.download(be.dAC(dev_slice(be, s), be.upload_env(GL), be.upload_env(GR), be.upload(x)))
    assert relerr(y, ref) < RTOL * max(Dl, Dr)


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_prepared_operator_matches_dAC(be, Dl, Dr, d, chis):
    """mpsk_hac (the MPO_ddAC object, derivatives.jl:11-15): prepared once, applied many times == mpsk_dAC == oracle,
    whichever factorisation the cost model picks; also with x in the blocked layout and a row-sharded left environment."""
    from mpskit_jl_amd import dist as md
    rng = np.random.default_rng(99 + Dl * 7 + Dr)
    s = rand_slice(rng, len(chis), d, chis)
    GL, GR = rand_env(rng, chis, Dl, Dl), rand_env(rng, chis, Dr, Dr)
    H, dGL, dGR = dev_slice(be, s), be.upload_env(GL), be.upload_env(GR)
    hac = be.hac_create(H, dGL, dGR)
    for rep in range(2):
        x = rng.standard_normal((Dl, d, Dr))
        ref = mo.dAC(x, s, GL, GR)
        assert relerr(be.download(hac.apply(be.upload(x))), ref) < RTOL * max(Dl, Dr)
    if Dl % 2 == 0:
        P = 2
        xb = md.to_blocked(be, be.upload(x), P)
        assert relerr(be.download(hac.apply(xb, nblk=P)), ref) < RTOL * max(Dl, Dr)
        n = Dl // P
        rows = md.rows_of_env(be, dGL, n, 2 * n)                         # second row block of every slab
        hloc = be.hac_create(H, rows, dGR)
        assert relerr(be.download(hloc.apply(xb, nblk=P)), ref[n:]) < RTOL * max(Dl, Dr)


@pytest.mark.parametrize("model", ["heis", "heis1", "tfi", "hubbard", "tfi2"])
def test_prepared_operator_model_slices(be, model):
    """The MPOs of the BASELINE configs through the prepared operator (MPO folded into the right environment: two GEMM
    launches, no slab mix) == oracle."""
    import mpskit_jl_amd as mk
    D = 96
    rng = np.random.default_rng(3)
    if model == "tfi2":
        X = np.array([[0.0, 1], [1, 0]]); Z = np.array([[1.0, 0], [0, -1]]); E = np.eye(2)
        Hg = mk.from_twosite(-(np.kron(Z, Z) + 0.65 * (np.kron(X, E) + np.kron(E, X))).reshape(2, 2, 2, 2), be=be)[0]
        Ho = mo.tfi_twosite_mpo(1.3)[0]
    else:
        Hg = {"heis": lambda: mk.heisenberg_XXX(0.5, be=be), "heis1": lambda: mk.heisenberg_XXX(1.0, be=be),
              "tfi": lambda: mk.transverse_field_ising(1.0, 0.7, be=be), "hubbard": lambda: mk.hubbard(1.0, 4.0, be=be)}[model]()[0]
        Ho = {"heis": lambda: mo.heisenberg_mpo(0.5), "heis1": lambda: mo.heisenberg_mpo(1.0),
              "tfi": lambda: mo.tfi_mpo(1.0, 0.7), "hubbard": lambda: mo.hubbard_mpo(1.0, 4.0)}[model]()[0]
    GL, GR = rand_env(rng, Ho.chil, D, D), rand_env(rng, Ho.chir, D, D)
    x = rng.standard_normal((D, Hg.d, D))
    hac = be.hac_create(Hg, be.upload_env(GL), be.upload_env(GR))
    assert hac.info()["mode"] == 1, hac.info()        # at D = 96 the saved mix pass / launch always pays
    assert relerr(be.download(hac.apply(be.upload(x))), mo.dAC(x, Ho, GL, GR)) < RTOL * D


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_dC(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(5 + Dl)
    GL, GR = rand_env(rng, chis, Dl, Dl), rand_env(rng, chis, Dr, Dr)
    c = rng.standard_normal((Dl, Dr))
    ref = mo.dC(c, GL, GR)
    y = be.download(be.dC(be.upload_env(GL), be.upload_env(GR), be.upload(c)))
    assert relerr(y, ref) < RTOL * max(Dl, Dr)


@pytest.mark.parametrize("Dl,Dr,d,chis", [c for c in CASES if c[0] <= 128])
def test_dAC2(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(11 + Dl)
    s1 = rand_slice(rng, len(chis), d, chis)
    s2 = rand_slice(rng, len(chis), d, chis)
    GL, GR = rand_env(rng, chis, Dl, Dl), rand_env(rng, chis, Dr, Dr)
    x = rng.standard_normal((Dl, d, Dr, d))
    ref = mo.dAC2(x, s1, s2, GL, GR)
    y = be.download(be.dAC2(dev_slice(be, s1), dev_slice(be, s2), be.upload_env(GL), be.upload_env(GR),
                            be.upload(x)))
    assert relerr(y, ref) < RTOL * max(Dl, Dr) * d


@pytest.mark.parametrize("Dl,Dr,d,chis", CASES)
def test_transfer_left_right(be, Dl, Dr, d, chis):
    rng = np.random.default_rng(17 + Dl)
    s = rand_slice(rng, len(chis), d, chis)
    ds = dev_slice(be, s)
    A = rng.standard_normal((Dl, d, Dr))
    Ab = rng.standard_normal((Dl, d, Dr))
    GL, GR = rand_env(rng, chis, Dl, Dl), rand_env(rng, chis, Dr, Dr)
    refL = mo.transfer_left(GL, s, A, Ab)
    outL = be.download_env(be.transfer_left(ds, be.upload_env(GL), be.upload(A), be.upload(Ab)), chis)
    for a, b in zip(outL, refL):
        assert relerr(a, b) < RTOL * max(Dl, Dr) * 10 or np.abs(b).max() == 0 and np.abs(a).max() == 0
    refR = mo.transfer_right(GR, s, A, Ab)
    outR = be.download_env(be.transfer_right(ds, be.upload_env(GR), be.upload(A), be.upload(Ab)), chis)
    for a, b in zip(outR, refR):
        assert relerr(a, b) < RTOL * max(Dl, Dr) * 10 or np.abs(b).max() == 0 and np.abs(a).max() == 0


@pytest.mark.parametrize("D,d", [(5, 2), (32, 3), (96, 2)])
def test_transfer_plain_and_regularize(be, D, d):
    rng = np.random.default_rng(23 + D)
    A, Ab = rng.standard_normal((D, d, D)), rng.standard_normal((D, d, D))
    v = rng.standard_normal((D, D))
    outL = be.download(be.transfer_left(None, be.upload(v[None]).reshape(1, D, D), be.upload(A), be.upload(Ab)))
    assert relerr(outL[0] if outL.ndim == 3 else outL, mo.transfer_left_bond(v, A, Ab)) < RTOL * D * 10
    outR = be.download(be.transfer_right(None, be.upload(v[None]).reshape(1, D, D), be.upload(A), be.upload(Ab)))
    assert relerr(outR[0] if outR.ndim == 3 else outR, mo.transfer_right_bond(v, A, Ab)) < RTOL * D * 10
    # regularize on a 3-level env
    env = [rng.standard_normal((D, 1, D)) for _ in range(3)]
    lvec, rvec = rng.standard_normal((D, D)), rng.standard_normal((D, D))
    dv = be.upload_env(env)
    be.regularize(dv, be.upload(lvec), be.upload(rvec))
    got = be.download_env(dv, [1, 1, 1])
    for g, e in zip(got, env):
        assert relerr(g, mo.regularize_env(e, lvec, rvec)) < 1e-12


@pytest.mark.parametrize("D1,D2,W", [(70, 45, 2), (33, 64, 1), (1024, 1024, 1), (300, 1000, 3)])
def test_regularize_rectangular_and_large(be, D1, D2, W):
    """mpsk_regularize's tiled two-phase reduction: ragged tiles, rectangular slabs, several slabs, and the
    1024 x 1024 case every GMRES step of the VUMPS / quasiparticle environments runs."""
    rng = np.random.default_rng(D1 + D2)
    env = [rng.standard_normal((D1, 1, D2)) for _ in range(W)]
    lvec, rvec = rng.standard_normal((D2, D1)), rng.standard_normal((D1, D2))
    dv = be.upload_env(env)
    be.regularize(dv, be.upload(lvec), be.upload(rvec))
    for g, e in zip(be.download_env(dv, [1] * W), env):
        assert relerr(g, mo.regularize_env(e, lvec, rvec)) < 1e-12


@pytest.mark.parametrize("M,N,K,tA,tB", [(64, 64, 64, 0, 0), (100, 37, 53, 0, 0), (100, 37, 53, 1, 0),
                                          (100, 37, 53, 0, 1), (100, 37, 53, 1, 1), (256, 128, 512, 1, 0),
                                          (128, 256, 64, 0, 1), (1, 1, 1, 0, 0), (130, 2, 1000, 1, 1)])
def test_gemm(be, M, N, K, tA, tB):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    C0 = rng.standard_normal((M, N))
    ref = 0.7 * (A.T if tA else A) @ (B.T if tB else B) - 1.3 * C0
    out = be.upload(C0)
    be.gemm(be.upload(A), be.upload(B), transA=bool(tA), transB=bool(tB), alpha=0.7, beta=-1.3, out=out)
    assert relerr(be.download(out), ref) < RTOL * K


def test_vectors(be):
    rng = np.random.default_rng(3)
    for n in (1, 7, 1000, 2 ** 20 + 3):
        xs = [rng.standard_normal(n) for _ in range(11)]
        y = rng.standard_normal(n)
        dxs = [be.upload(x) for x in xs]
        dy = be.upload(y)
        assert abs(be.dot(dxs[0], dy) - xs[0] @ y) < 1e-12 * n
        assert abs(be.norm(dy) - np.linalg.norm(y)) < 1e-12 * np.sqrt(n)
        md = be.multidot(dxs, dy)
        assert np.allclose(md, [x @ y for x in xs], rtol=0, atol=1e-12 * n)
        h = be.gs_step(dxs, dy)
        assert np.allclose(h, md, rtol=0, atol=1e-12 * n)
        assert relerr(be.download(dy), y - sum(c * x for c, x in zip(md, xs))) < 1e-12
        z = be.lincomb(dxs, np.arange(1, 12))
        assert relerr(be.download(z), sum((i + 1) * x for i, x in enumerate(xs))) < 1e-13
        be.axpby(2.0, dxs[0], -0.5, z)
        be.scal(3.0, z)
        assert relerr(be.download(z), 3 * (2 * xs[0] - 0.5 * sum((i + 1) * x for i, x in enumerate(xs)))) < 1e-13


@pytest.mark.parametrize("k,m", [(1, 1), (3, 2), (8, 5), (9, 9), (16, 10), (17, 3), (30, 18), (32, 32)])
def test_multilincomb_basis_rotation(be, k, m):
    """mpsk_vmultilincomb: outs[j] = sum_i S[i, j] xs[i] in one pass (the thick-restart rotation of the Krylov basis,
    KrylovKit shrink step) == numpy, for every register variant (8 / 16 / 24 / 32 inputs) and odd lengths; aliasing an
    output with an input is refused."""
    from mpskit_jl_amd._lib import MpskError
    rng = np.random.default_rng(10 * k + m)
    for n in (7, 100003):
        X = rng.standard_normal((n, k))
        S = rng.standard_normal((k, m))
        dxs = [be.upload(np.ascontiguousarray(X[:, i])) for i in range(k)]
        outs = [be.upload(np.full(n, np.nan)) for _ in range(m)]
        be.multilincomb(dxs, S, outs)
        ref = X @ S
        for j in range(m):
            assert np.abs(be.download(outs[j]).ravel() - ref[:, j]).max() < 1e-13 * max(1.0, np.abs(ref).max())
    with pytest.raises(MpskError, match="alias"):
        be.multilincomb(dxs, S, [dxs[0]] + outs[1:])


@pytest.mark.parametrize("n", [5, 4099, 2 ** 18])
def test_orth_step_every_basis_length(be, n):
    """mpsk_vorth_step (CGS2 + normalise, the Krylov loops' orthogonalisation: KrylovKit ModifiedGramSchmidt2) for every
    basis length 1 .. 34: k <= 8 runs the fused pair of passes, 9 .. 32 their long twins (16 / 24 / 32-vector register
    variants with zero-padded coefficients), 33+ the chunked fallback -- all against two rounds of classical Gram-Schmidt
    in numpy on a basis that is only roughly orthonormal (so the second round matters)."""
    rng = np.random.default_rng(n)
    for k in range(1, 35):
        Q, _ = np.linalg.qr(rng.standard_normal((max(n, k), k)))
        X = (Q[:n] if n >= k else rng.standard_normal((n, k))) + 1e-3 * rng.standard_normal((n, k))
        y = rng.standard_normal(n)
        h1 = X.T @ y; y1 = y - X @ h1
        h2 = X.T @ y1; y2 = y1 - X @ h2
        dxs = [be.upload(np.ascontiguousarray(X[:, j])) for j in range(k)]
        dy = be.upload(y)
        h, beta = be.orth_step(dxs, dy)
        scale = np.linalg.norm(y)
        assert np.abs(h - (h1 + h2)).max() < 1e-12 * scale, k
        assert abs(beta - np.linalg.norm(y2)) < 1e-12 * scale, k
        if np.linalg.norm(y2) > 1e-8 * scale:
            assert np.abs(be.download(dy).ravel() - y2 / np.linalg.norm(y2)).max() < 1e-11, k


QR_CASES = [(8, 4), (4, 4), (64, 64), (100, 37), (33, 1), (768, 256), (1030, 515), (2048, 1024), (4096, 1024)]


@pytest.fixture(params=[0, 1], ids=["qr-auto", "qr-householder"])
def qr_mode(request, be):
    """0: shifted CholeskyQR3 with Householder fallback (default); 1: Householder only."""
    be.set_qr_mode(request.param)
    yield request.param
    be.set_qr_mode(0)


def test_qrpos_fallback_counts(be):
    rng = np.random.default_rng(0)
    s0 = be.qr_stats()
    be.qrpos(be.upload(rng.random((512, 256))))            # well conditioned -> CholeskyQR3
    A = rng.random((512, 256))
    A[:, 7] = 0.0                                            # rank deficient -> flagged, finished by the robust variant
    Q, R = be.qrpos(be.upload(A))
    s1 = be.qr_stats()
    assert s1["cholqr3"] == s0["cholqr3"] + 1
    assert s1["fallback"] == s0["fallback"] + 1
    assert (s1["robust"] - s0["robust"]) + (s1["householder"] - s0["householder"]) == 1
    Q, R = be.download(Q), be.download(R)
    assert np.abs(Q.T @ Q - np.eye(256)).max() < 1e-12 and relerr(Q @ R, A) < 1e-13
    assert np.all(np.diag(R) >= 0) and np.abs(np.tril(R, -1)).max() == 0.0


@pytest.mark.parametrize("m,n", [(8, 4), (768, 256), (2048, 1024)])
def test_qrpos2_pair(be, m, n):
    """two factorizations in flight on two streams == two single calls (incl. one rank-deficient input)."""
    rng = np.random.default_rng(m + n)
    A1, A2 = rng.random((m, n)), rng.standard_normal((m, n))
    if n > 64:
        A2[:, 3] = 0.0                      # forces the Householder fallback for the second matrix only
    Q1, R1, Q2, R2 = (be.download(t) for t in be.qrpos2(be.upload(A1), be.upload(A2)))
    for A, Q, R in ((A1, Q1, R1), (A2, Q2, R2)):
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12
        assert relerr(Q @ R, A) < 1e-12
        assert np.all(np.diag(R) >= 0) and np.abs(np.tril(R, -1)).max(initial=0.0) == 0.0
    Qs, Rs = (be.download(t) for t in be.qrpos(be.upload(A1)))
    assert relerr(Q1, Qs) < 1e-12 and relerr(R1, Rs) < 1e-12


@pytest.mark.parametrize("m,n", QR_CASES)
def test_qrpos(be, qr_mode, m, n):
    rng = np.random.default_rng(m * 31 + n)
    A = rng.random((m, n))          # uniform[0,1) like the reference's `rand`
    Q, R = be.qrpos(be.upload(A))
    Q, R = be.download(Q), be.download(R)
    Qr, Rr = mo.qrpos(A)
    assert np.all(np.diag(R) > 0)
    assert np.abs(np.tril(R, -1)).max(initial=0.0) == 0.0
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13 * np.sqrt(m)
    assert relerr(Q @ R, A) < 1e-12
    assert relerr(R, Rr) < 1e-11 and relerr(Q, Qr) < 1e-10


def test_qrpos_illconditioned_and_rank_deficient(be, qr_mode):
    rng = np.random.default_rng(7)
    m, n = 512, 256
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -14, n)     # Schmidt-like spectrum, cond = 1e14
    A = (U * s) @ V.T
    Q, R = be.qrpos(be.upload(A))
    Q, R = be.download(Q), be.download(R)
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12
    assert relerr(Q @ R, A) < 1e-13
    # exactly rank deficient: zero columns + duplicated column
    A2 = rng.random((96, 48))
    A2[:, 5] = 0.0
    A2[:, 9] = A2[:, 3]
    Q, R = be.qrpos(be.upload(A2))
    Q, R = be.download(Q), be.download(R)
    assert np.all(np.isfinite(Q)) and np.all(np.isfinite(R))
    assert relerr(Q @ R, A2) < 1e-13
    assert np.abs(Q.T @ Q - np.eye(48)).max() < 1e-12


@pytest.mark.parametrize("m,n", [(4, 8), (64, 64), (37, 100), (256, 768), (1024, 2048)])
def test_lqpos(be, m, n):
    rng = np.random.default_rng(m * 17 + n)
    A = rng.random((m, n))
    L, Q = be.lqpos(be.upload(A))
    L, Q = be.download(L), be.download(Q)
    Lr, Qr = mo.lqpos(A)
    assert np.all(np.diag(L) > 0) and np.abs(np.triu(L, 1)).max(initial=0.0) == 0.0
    assert np.abs(Q @ Q.T - np.eye(m)).max() < 1e-13 * np.sqrt(n)
    assert relerr(L @ Q, A) < 1e-13
    assert relerr(L, Lr) < 1e-11 and relerr(Q, Qr) < 1e-10


@pytest.mark.parametrize("m,n", [(4, 4), (6, 10), (64, 64), (100, 37), (37, 100), (256, 256), (768, 768),
                                 (2048, 1024)])
def test_tsvd_full(be, m, n):
    rng = np.random.default_rng(m * 13 + n)
    A = rng.random((m, n))
    U, S, Vh, kept, disc = be.tsvd(be.upload(A))
    U, S, Vh = be.download(U), be.download(S), be.download(Vh)
    k = min(m, n)
    assert kept == k and disc == 0.0
    Sr = np.linalg.svd(A, compute_uv=False)
    assert np.all(np.diff(S) <= 0)
    assert np.abs(S - Sr).max() < 1e-12 * Sr[0]
    assert np.abs(U.T @ U - np.eye(k)).max() < 1e-12
    assert np.abs(Vh @ Vh.T - np.eye(k)).max() < 1e-12
    assert relerr((U * S) @ Vh, A) < 1e-12


def test_tsvd_truncation_and_small_singular_values(be):
    rng = np.random.default_rng(99)
    m = n = 192
    Uo, _ = np.linalg.qr(rng.standard_normal((m, n)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -12, n)                    # Schmidt-like spectrum
    A = (Uo * s) @ Vo.T
    # truncdim
    U, S, Vh, kept, disc = be.tsvd(be.upload(A), max_keep=40)
    S = be.download(S)
    assert kept == 40
    # backward-stable accuracy: absolute error ~ eps * sigma_max for EVERY singular value
    # (the mixing V makes the small ones ill-determined relatively; LAPACK gives the same)
    assert np.abs(S[:n] - s).max() < 1e-14
    assert abs(disc - np.linalg.norm(s[40:])) < 1e-14
    U, Vh = be.download(U)[:, :kept], be.download(Vh)[:kept]
    assert relerr((U * S[:kept]) @ Vh, (Uo[:, :40] * s[:40]) @ Vo[:, :40].T) < 1e-12
    # truncerr (TensorKit 0.12 semantics: discarded 2-norm <= eps, ABSOLUTE), cf. oracle tsvd; pinned on an
    # UNNORMALISED theta (|theta| = 3.7 |A|), where the absolute and the relative rule keep different numbers of values
    eps = 1e-6
    for scale in (1.0, 3.7e3):
        _, S2, _, kept2, disc2 = be.tsvd(be.upload(scale * A), trunc_err=eps)
        _, So, _, erro = mo.tsvd((scale * A).reshape(m, 1, n, 1), truncerr=eps)
        assert kept2 == len(So) == int(np.sum(np.sqrt(np.cumsum((scale * s[::-1]) ** 2))[::-1] > eps))
        assert abs(disc2 - erro) < 1e-13 * scale


@pytest.mark.parametrize("m,n", [(100, 37), (37, 100), (300, 300), (640, 256), (256, 640)])
def test_tsvd_plain_mode_matches_preconditioned(be, m, n):
    """mpsk_ctx_set_svd_mode(0): Jacobi on theta itself -- same contract as the default QR-preconditioned path."""
    rng = np.random.default_rng(m + 7 * n)
    A = rng.random((m, n))
    k = min(m, n)
    Sr = np.linalg.svd(A, compute_uv=False)
    try:
        for mode in (False, True):
            be.set_svd_mode(mode)
            U, S, Vh, kept, disc = be.tsvd(be.upload(A))
            U, S, Vh = be.download(U), be.download(S), be.download(Vh)
            assert kept == k and np.abs(S - Sr).max() < 1e-12 * Sr[0]
            assert np.abs(U.T @ U - np.eye(k)).max() < 1e-12 and np.abs(Vh @ Vh.T - np.eye(k)).max() < 1e-12
            assert relerr((U * S) @ Vh, A) < 1e-12
    finally:
        be.set_svd_mode(3)          # the ctx default


def test_tsvd_graded_and_rank_deficient_preconditioned(be):
    """The DMRG case: Schmidt-like spectrum over 14 decades behind random orthogonal factors, plus an exactly
    rank-deficient theta (QRpos falls back to Householder, R^T has zero columns).  The preconditioned
    iteration needs few sweeps where plain block Jacobi needs tens."""
    rng = np.random.default_rng(5)
    m, n = 640, 512
    Uo, _ = np.linalg.qr(rng.standard_normal((m, n)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -14, n)
    A = (Uo * s) @ Vo.T
    U, S, Vh, kept, disc = be.tsvd(be.upload(A), max_keep=128)
    sw_pre = be.svd_sweeps()
    S = be.download(S)
    assert np.abs(S[:n] - s).max() < 2e-14
    U, Vh = be.download(U)[:, :kept], be.download(Vh)[:kept]
    assert np.abs(U.T @ U - np.eye(kept)).max() < 1e-12 and np.abs(Vh @ Vh.T - np.eye(kept)).max() < 1e-12
    assert relerr((U * S[:kept]) @ Vh, (Uo[:, :128] * s[:128]) @ Vo[:, :128].T) < 1e-12
    from mpskit_jl_amd._lib import MpskError
    try:
        be.set_svd_mode(False)
        # plain block Jacobi cannot converge on this input, for a mathematical reason (include/mpsk.h, svd mode 0): the
        # columns of A are nearly parallel (cond of the column-scaled matrix 1e14 ~ 1/u); each rotation against an O(1)
        # column injects noise u |a_big| ~ 1e-16 into columns whose norm IS 1e-14..1e-16, so their mutual cosines stay
        # O(1) (measured: max |cos| = 0.41 after 40 sweeps) and the scale-invariant test never passes.  The library
        # says so instead of returning non-isometries (mpsk.h: "did not converge"); modes 1 / 2 iterate on R^T = B D with
        # B well conditioned (Demmel-Veselic), where the same 14-decade spectrum converges in <= 12 sweeps.
        with pytest.raises(MpskError, match="did not converge"):
            be.tsvd(be.upload(A), max_keep=128)
        sw_plain = be.svd_sweeps()
    finally:
        be.set_svd_mode(3)          # the ctx default
    assert sw_pre <= 12 and sw_pre < sw_plain == 40, (sw_pre, sw_plain)
    # exact rank deficiency: rank 100 of 256
    B = rng.standard_normal((384, 100)) @ rng.standard_normal((100, 256))
    U, S, Vh, kept, disc = be.tsvd(be.upload(B))
    U, S, Vh = be.download(U), be.download(S), be.download(Vh)
    Sr = np.linalg.svd(B, compute_uv=False)
    assert np.abs(S - Sr).max() < 1e-12 * Sr[0]
    assert relerr((U * S) @ Vh, B) < 1e-12
    assert np.abs(U[:, :100].T @ U[:, :100] - np.eye(100)).max() < 1e-12


@pytest.mark.parametrize("M,N,K,tA,tB", [(1024, 512, 2048, 0, 0),     # split-K (64 tiles x 128 k-tiles), XCD grid
                                          (512, 512, 4096, 1, 0),      # TN Gram shape, split-K
                                          (2048, 1024, 256, 0, 0),     # 512 tiles, XCD rectangles 4 x 2
                                          (1024, 2048, 128, 0, 1),     # NT, rectangles 2 x 4
                                          (1032, 520, 1040, 0, 0),     # ragged -> unaligned kernel, linear tile order
                                          (4096, 64, 64, 0, 0)])       # tall skinny (SVD update shape)
def test_gemm_large_paths(be, M, N, K, tA, tB):
    """Every launch-path of the GEMM core (split-K + fixup, XCD-rectangle tile map, unaligned loaders) against
    numpy at sizes that trigger it; also with the remap / split switched off via the environment-free knobs."""
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    C0 = rng.standard_normal((M, N))
    ref = 1.1 * (A.T if tA else A) @ (B.T if tB else B) + 0.5 * C0
    dA, dB = be.upload(A), be.upload(B)
    for tile in ((0, 0), (64, 64), (128, 128)):
        be.lib.mpsk_ctx_force_tile(be.ctx, *tile)
        try:
            out = be.upload(C0)
            be.gemm(dA, dB, transA=bool(tA), transB=bool(tB), alpha=1.1, beta=0.5, out=out)
            assert relerr(be.download(out), ref) < RTOL * K
        finally:
            be.lib.mpsk_ctx_force_tile(be.ctx, 0, 0)


def test_abi_error_behaviour(be):
    """Errors are return codes + mpsk_last_error (no exceptions across the boundary): bad arguments are
    rejected before any launch, the message names the check, and the context stays usable."""
    import ctypes as C
    from mpskit_jl_amd._lib import MpskError
    lib = be.lib
    A = be.upload(np.random.default_rng(0).random((8, 4)))
    Q, R = be.empty(8, 4), be.empty(4, 4)
    assert lib.mpsk_qrpos(be.ctx, 4, 8, A.ptr, 4, Q.ptr, 4, R.ptr, 8) == 1          # m < n
    assert b"m >= n" in lib.mpsk_last_error()
    assert lib.mpsk_qrpos(be.ctx, 8, 4, A.ptr, 4, Q.ptr, 8, R.ptr, 4) == 1          # lda < m
    assert lib.mpsk_qrpos(be.ctx, 8, 4, None, 8, Q.ptr, 8, R.ptr, 4) == 1           # NULL
    assert lib.mpsk_lqpos(be.ctx, 8, 4, A.ptr, 8, R.ptr, 8, Q.ptr, 8) == 1          # m > n
    k, disc = C.c_int(0), C.c_double(0.0)
    assert lib.mpsk_tsvd(be.ctx, 8, 4, A.ptr, 8, Q.ptr, 8, R.ptr, R.ptr, 4, 0, -1.0, C.byref(k), C.byref(disc)) == 1
    assert lib.mpsk_gemm(be.ctx, 0, 0, 0, 4, 4, 1.0, A.ptr, 8, A.ptr, 8, 0.0, Q.ptr, 8) == 1
    assert lib.mpsk_ctx_set_qr_mode(be.ctx, 7) == 1
    with pytest.raises(MpskError):
        be.upload(np.ones((2, 2)) * (1 + 1j))                                       # complex128 is not built
    with pytest.raises(AssertionError):
        be.dC(be.empty(3, 4, 4), be.empty(2, 5, 5), be.empty(4, 5))                 # level count mismatch
    Qg, Rg = be.qrpos(A)                                                            # context still fine
    assert relerr(be.download(Qg) @ be.download(Rg), be.download(A)) < 1e-13


@pytest.mark.parametrize("m,n", [(8, 4), (768, 384), (2048, 1024)])
def test_qrlq_pair(be, m, n):
    """mpsk_qrlq_pair == mpsk_qrpos of the first + mpsk_lqpos of the second operand."""
    rng = np.random.default_rng(3 * m + n)
    A1, A2 = rng.random((m, n)), rng.standard_normal((n, m))
    Q1, R1, L2, Q2 = (be.download(t) for t in be.qrlq_pair(be.upload(A1), be.upload(A2)))
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() < 1e-12 and relerr(Q1 @ R1, A1) < 1e-12
    assert np.abs(Q2 @ Q2.T - np.eye(n)).max() < 1e-12 and relerr(L2 @ Q2, A2) < 1e-12
    assert np.all(np.diag(R1) > 0) and np.all(np.diag(L2) > 0)
    assert np.abs(np.tril(R1, -1)).max() == 0.0 and np.abs(np.triu(L2, 1)).max() == 0.0
    Qs, Rs = (be.download(t) for t in be.qrpos(be.upload(A1)))
    Ls, Qls = (be.download(t) for t in be.lqpos(be.upload(A2)))
    assert relerr(Q1, Qs) < 1e-12 and relerr(L2, Ls) < 1e-11 and relerr(Q2, Qls) < 1e-10


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("m,n,k", [(512, 384, 100), (384, 512, 100), (1024, 1024, 256), (200, 130, 0), (1536, 1280, 300),
                                   (1024, 1024, 512)])          # the last: keep HALF (d = 2 chains), stage with r = 5n/8
def test_tsplit(be, m, n, k, mode):
    """mpsk_tsplit (V-free Jacobi + rebuilt factor): al, ar isometries, al c ar = the optimal rank-k truncation of theta
    (same singular values / discarded norm as numpy), c triangular, for both orientations and a graded spectrum.
    Mode 3 (default): the truncation-aware stage (subspace iteration + Jacobi on r = k + max(64, k/2) columns) must have
    produced the result (path 1) and meets the same bounds."""
    rng = np.random.default_rng(m + 3 * n + k)
    r = min(m, n)
    Uo, _ = np.linalg.qr(rng.standard_normal((m, r)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, r)))
    s = np.logspace(0, -9, r)
    A = (Uo * s) @ Vo.T
    be.set_svd_mode(mode)          # 1: QR-preconditioned, 2: QR + QR of R^T (left singular vectors come out of the iteration)
    try:
        al, c, ar, S, disc = be.tsplit(be.upload(A), max_keep=k)
        st = be.split_stats()
    finally:
        be.set_svd_mode(3)
    assert st["path"] == (1 if (mode == 3 and k > 0) else 0)
    assert mode != 3 or k == 0 or st["residual"] <= 1e-12
    kk = k if k > 0 else r
    al, c, ar = be.download(al), be.download(c), be.download(ar)
    assert al.shape == (m, kk) and c.shape == (kk, kk) and ar.shape == (kk, n)
    assert np.abs(S - s[:kk]).max() < 1e-13
    assert abs(disc - np.linalg.norm(s[kk:])) < 1e-13
    assert np.abs(al.T @ al - np.eye(kk)).max() < 1e-12 and np.abs(ar @ ar.T - np.eye(kk)).max() < 1e-12
    best = (Uo[:, :kk] * s[:kk]) @ Vo[:, :kk].T
    assert np.abs(al @ c @ ar - best).max() < 1e-12
    assert np.abs(np.linalg.svd(c, compute_uv=False) - s[:kk]).max() < 1e-13
    assert np.abs(np.tril(c, -1)).max() < 1e-13 or np.abs(np.triu(c, 1)).max() < 1e-13


def test_tsplit_truncation_aware_stage_gives_up_on_flat_spectra_and_backs_off(be):
    """svd mode 3 on a spectrum without decay behind the cut (uniform random theta: sigma_{r+1} / sigma_k ~ 0.9): the
    check fails, the predicted iteration count is over budget, the call falls through to the full iteration (path 2) and
    returns the same result as mode 2; the next calls skip the stage (path 0) -- flat spectra come in runs."""
    rng = np.random.default_rng(5)
    n, k = 768, 128
    A = rng.random((n, n)) - 0.5
    sref = np.linalg.svd(A, compute_uv=False)
    import mpskit_jl_amd as mk
    be2 = mk.Backend(0)                       # own ctx: the back-off counter is per ctx
    dA = be2.upload(A)
    paths = []
    for _ in range(3):
        al, c, ar, S, disc = be2.tsplit(dA, max_keep=k)
        paths.append(be2.split_stats()["path"])
        assert np.abs(S - sref[:k]).max() < 1e-12 * sref[0]
        assert abs(disc - np.linalg.norm(sref[k:])) < 1e-11 * sref[0]
        a_, c_, r_ = be2.download(al), be2.download(c), be2.download(ar)
        assert np.abs(a_.T @ a_ - np.eye(k)).max() < 1e-12 and np.abs(r_ @ r_.T - np.eye(k)).max() < 1e-12
        assert abs(np.linalg.norm(A - a_ @ c_ @ r_) - np.linalg.norm(sref[k:])) < 1e-11 * sref[0]
    be2.close()
    assert paths == [2, 0, 0]


@pytest.mark.parametrize("m,n,k,mode", [(3200, 1024, 256, 2), (3200, 1024, 256, 3), (2048, 2048, 800, 3), (1024, 3200, 256, 3)])
def test_tsplit_workspace_of_the_inner_factorizations_on_a_fresh_ctx(m, n, k, mode):
    """The QRpos workspace is not monotone in the shape (the small regime's in-step solve / Gram adds CQ_GS npad^2 doubles):
    mpsk_tsplit / mpsk_tsvd must size the ctx workspace for the LARGEST of their inner factorizations, not for the first.
    On a fresh ctx (nothing grew the workspace before) a tall theta whose R^T is in the small regime, and a 2048^2 split whose
    1216-column Jacobi stage is (a GPU memory fault before the fix), must simply work."""
    import mpskit_jl_amd as mk
    rng = np.random.default_rng(m + n + k)
    r = min(m, n)
    Uo, _ = np.linalg.qr(rng.standard_normal((m, r)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, r)))
    s = np.logspace(0, -6, r)
    A = (Uo * s) @ Vo.T
    be2 = mk.Backend(0)
    try:
        be2.set_svd_mode(mode)
        al, c, ar, S, disc = be2.tsplit(be2.upload(A), max_keep=k)
        assert be2.split_stats()["path"] == (1 if mode == 3 else 0)
        assert np.abs(S - s[:k]).max() < 1e-13 and abs(disc - np.linalg.norm(s[k:])) < 1e-13
        al, c, ar = be2.download(al), be2.download(c), be2.download(ar)
        assert np.abs(al.T @ al - np.eye(k)).max() < 1e-12 and np.abs(ar @ ar.T - np.eye(k)).max() < 1e-12
        assert np.abs(al @ c @ ar - (Uo[:, :k] * s[:k]) @ Vo[:, :k].T).max() < 1e-12
        U, Sd, Vh, kept, _ = be2.tsvd(be2.upload(A), max_keep=k)          # mpsk_tsvd has the same two factorizations
        assert kept == k and np.abs(be2.download(Sd).ravel()[:k] - s[:k]).max() < 1e-13
    finally:
        be2.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_tsplit_extreme_scales_and_zero(be, mode):
    """The factorizations inside the split square the scale of theta (Gram matrices): tensors of norm 1e-200 or 1e+150 used to
    underflow / overflow there (zero 'isometries' accepted by the residual check, which a collapsed basis passes trivially).
    The front end now splits theta / |theta| and scales C, S and the discarded norm back; theta = 0 gets C = 0 and isometries."""
    rng = np.random.default_rng(3)
    n, k = 512, 96
    Uo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s0 = np.logspace(0, -6, n)
    be.set_svd_mode(mode)
    try:
        for scale in (1e-200, 1e150, 1.0):
            A = (Uo * (scale * s0)) @ Vo.T
            al, c, ar, S, disc = be.tsplit(be.upload(A), max_keep=k)
            a_, c_, r_ = be.download(al), be.download(c), be.download(ar)
            assert np.abs(a_.T @ a_ - np.eye(k)).max() < 1e-12 and np.abs(r_ @ r_.T - np.eye(k)).max() < 1e-12
            assert np.abs(S / scale - s0[:k]).max() < 1e-13
            assert abs(disc / scale - np.linalg.norm(s0[k:])) < 1e-13
            assert np.abs(a_ @ c_ @ r_ / scale - (Uo[:, :k] * s0[:k]) @ Vo[:, :k].T).max() < 1e-12
        al, c, ar, S, disc = be.tsplit(be.upload(np.zeros((n, n))), max_keep=k)
        a_, c_, r_ = be.download(al), be.download(c), be.download(ar)
        assert np.abs(a_.T @ a_ - np.eye(k)).max() == 0.0 and np.abs(r_ @ r_.T - np.eye(k)).max() == 0.0
        assert np.abs(c_).max() == 0.0 and np.abs(S).max() == 0.0 and disc == 0.0
    finally:
        be.set_svd_mode(3)


def test_tsplit_dominance_probe_rejects_an_unconverged_subspace(be):
    """The last line of defence of svd mode 3: with the residual check disabled and a single subspace iteration (test hook
    MPSK_SPLIT_DEBUG_SKIP_CHECK), the kept subspace is far from the dominant one, the remainder theta - AL M contains
    directions above the cut, and the power-iteration probe must send the call to the full iteration (path 2) -- the
    result is then the exact one."""
    import os
    import mpskit_jl_amd as mk
    rng = np.random.default_rng(21)
    n, k = 640, 128
    Uo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -2, n)                  # slow decay: one iteration leaves O(1) errors
    A = (Uo * s) @ Vo.T
    be2 = mk.Backend(0)
    os.environ["MPSK_SPLIT_DEBUG_SKIP_CHECK"] = "1"
    try:
        al, c, ar, S, disc = be2.tsplit(be2.upload(A), max_keep=k)
        st = be2.split_stats()
    finally:
        del os.environ["MPSK_SPLIT_DEBUG_SKIP_CHECK"]
    assert st["path"] == 2 and st["iterations"] == 1
    assert np.abs(S - s[:k]).max() < 1e-13 and abs(disc - np.linalg.norm(s[k:])) < 1e-13
    best = (Uo[:, :k] * s[:k]) @ Vo[:, :k].T
    assert np.abs(be2.download(al) @ be2.download(c) @ be2.download(ar) - best).max() < 1e-12
    be2.close()


def test_tsplit_truncation_aware_rank_deficient_and_truncerr(be):
    """svd mode 3 where the subspace is wider than the rank of theta (rank 150, r = 192: the early two-site sweeps of a
    growing chain) and with the truncerr rule deciding the cut: same kept rank, values and discarded weight as mode 2."""
    rng = np.random.default_rng(9)
    m, n, rank = 640, 512, 150
    s = np.logspace(0, -7, rank)
    Uo, _ = np.linalg.qr(rng.standard_normal((m, rank)))
    Vo, _ = np.linalg.qr(rng.standard_normal((n, rank)))
    A = (Uo * s) @ Vo.T
    dA = be.upload(A)
    res = {}
    for mode in (2, 3):
        be.set_svd_mode(mode)
        try:
            res[mode] = [be.tsplit(dA, max_keep=100), be.tsplit(dA, max_keep=128, trunc_err=1e-4)]
            path = be.split_stats()["path"]
        finally:
            be.set_svd_mode(3)
        assert path == (1 if mode == 3 else 0)
    for (a2, c2, r2, S2, d2), (a3, c3, r3, S3, d3) in zip(res[2], res[3]):
        assert len(S2) == len(S3) and np.abs(S2 - S3).max() < 1e-13
        assert abs(d2 - d3) < 1e-13
        k = len(S3)
        A3 = be.download(a3) @ be.download(c3) @ be.download(r3)
        best = (Uo[:, :k] * s[:k]) @ Vo[:, :k].T
        assert np.abs(A3 - best).max() < 1e-12
        assert np.abs(be.download(a3).T @ be.download(a3) - np.eye(k)).max() < 1e-12
    assert len(res[3][1][3]) < 128                      # truncerr(1e-4) cut below max_keep: tail norm <= 1e-4
    assert np.linalg.norm(s[len(res[3][1][3]):]) <= 1e-4 < np.linalg.norm(s[len(res[3][1][3]) - 1:])


@pytest.mark.parametrize("m,n", [(48, 40), (40, 48), (64, 64)])
def test_tsplit_small_rank_deficient_keeps_isometries(be, m, n):
    """min(m, n) <= 64 goes through mpsk_tsvd: with a rank-deficient theta and max_keep > rank the kept singular values
    include exact zeros, whose one-sided-Jacobi vectors are zero columns; the split must still return isometries
    (LAPACK's tsvd! gives an orthonormal completion; the lazy-gauge FiniteMPS and FinEnv assume AL^T AL = I)."""
    rng = np.random.default_rng(m * n)
    r, keep = 10, 24
    A = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    al, c, ar, S, disc = be.tsplit(be.upload(A), max_keep=keep)
    al, c, ar = be.download(al), be.download(c), be.download(ar)
    assert al.shape == (m, keep) and ar.shape == (keep, n)
    assert np.abs(S[r:]).max() < 1e-12 * S[0]
    assert np.abs(al.T @ al - np.eye(keep)).max() < 1e-12
    assert np.abs(ar @ ar.T - np.eye(keep)).max() < 1e-12
    assert np.abs(al @ c @ ar - A).max() < 1e-11 * np.abs(A).max()


def test_bond_matrix_inverse_on_device(be):
    """inv(CR) of the IDMRG2 edge step (idmrg.jl:118,150) by mpsk_tsvd + two GEMMs, graded spectrum included."""
    from mpskit_jl_amd import algorithms as alg
    rng = np.random.default_rng(3)
    for D, grade in ((12, 0.0), (96, 0.0), (64, 6.0)):
        U, _ = np.linalg.qr(rng.standard_normal((D, D)))
        V, _ = np.linalg.qr(rng.standard_normal((D, D)))
        Cm = (U * np.logspace(0, -grade, D)) @ V.T if grade else rng.standard_normal((D, D))
        inv = be.download(alg._bond_inv(be, be.upload(Cm)))
        assert np.abs(inv @ Cm - np.eye(D)).max() < 1e-9 * np.linalg.cond(Cm)


def test_device_ritz_step_matches_host(be):
    """Fixed-budget eigsolve with the Ritz step on the device (values=False: mpsk_vritz_dev + mpsk_vlincomb_dev, no host
    synchronisation) against the same solve with the host Ritz step (numpy eigh on the downloaded scalars), including an
    invariant-subspace cut (operator of rank 3 -> the Krylov space closes after 3-4 steps)."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import krylov
    rng = np.random.default_rng(11)
    n = 96
    for rank in (None, 3):
        M = rng.standard_normal((n, n)); M = M + M.T
        if rank is not None:
            U = np.linalg.qr(rng.standard_normal((n, rank)))[0]
            M = U @ np.diag([-3.0, 1.0, 2.5]) @ U.T
        Md = be.upload(M)

        def mv(x, out):
            return be.gemm(Md, x.reshape(n, 1), out=out.reshape(n, 1))

        x0 = be.upload(rng.standard_normal((n, 1, 1)))
        for m in (1, 2, 5, 8, 20):
            lam, v1, _, res = krylov.eigsolve_sr(be, mv, x0, fixed_matvecs=m, krylovdim=m)
            none, v2, _, none2 = krylov.eigsolve_sr(be, mv, x0, fixed_matvecs=m, krylovdim=m, values=False)
            assert none is None and none2 is None
            a, b = be.download(v1).ravel(), be.download(v2).ravel()
            assert abs(np.linalg.norm(b) - 1.0) < 1e-13
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-11, (rank, m)
            assert b @ be.download(x0).ravel() > 0            # sign convention: positive overlap with the start vector
            assert abs(b @ M @ b - lam) < 1e-10 * max(1.0, abs(lam))


def test_deferred_gauge_completion(be):
    """mpsk_ctx_qr_defer / mpsk_qr_commit: the factorization returns once enqueued, workspace-free calls are accepted in
    between, workspace users are refused, and after the commit the factors equal those of the immediate calls -- for a
    well-conditioned pair, for an ill-conditioned matrix whose third pass is repeated at commit (redone bit), and for LQ."""
    from mpskit_jl_amd._lib import MpskError
    rng = np.random.default_rng(21)
    m, n = 768, 256
    A1 = rng.standard_normal((m, n))
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A2 = (U * np.logspace(0, -9, n)) @ V.T                       # cond 1e9: the first-order third pass is not enough
    d1, d2 = be.upload(A1), be.upload(A2)
    Qa, Ra = (be.download(t) for t in be.qrpos(d1))
    Qb, Rb = (be.download(t) for t in be.qrpos(d2))
    be.qr_defer()
    Q1, R1, Q2, R2 = be.qrpos2(d1, d2)
    spec = be.gemm(Q1, R1)                                        # workspace-free: accepted while the pair is pending
    with pytest.raises(MpskError):
        be.qrpos(d1)                                              # needs the ctx workspace: refused until the commit
    redone = be.qr_commit()
    assert redone in (0, 1, 2, 3) and not (redone & 1)            # the well-conditioned factor is final at the enqueue
    assert relerr(be.download(spec), A1) < 1e-13
    assert relerr(be.download(Q1), Qa) < 1e-12 and relerr(be.download(R1), Ra) < 1e-12
    assert relerr(be.download(Q2), Qb) < 1e-7 and relerr(be.download(R2), Rb) < 1e-9
    q2 = be.download(Q2)
    assert np.abs(q2.T @ q2 - np.eye(n)).max() < 1e-13
    assert be.qr_commit() == 0                                    # nothing pending: a no-op
    B = A1.T.copy()                                               # 256 x 768
    La, Qla = (be.download(t) for t in be.lqpos(be.upload(B)))
    be.qr_defer()
    L, Ql = be.lqpos(be.upload(B))
    be.gemm(Q1, R1)
    be.qr_commit()
    assert relerr(be.download(L), La) < 1e-12 and relerr(be.download(Ql), Qla) < 1e-12


def test_defer_is_not_inherited_by_composite_entry_points(be):
    """mpsk_ctx_qr_defer arms the NEXT mpsk_qrpos2 / mpsk_lqpos; mpsk_qrlq_pair, mpsk_tsplit and mpsk_tsvd call
    those internally and read the factors right away, so they clear the flag on entry (and refuse a pending deferral)."""
    from mpskit_jl_amd._lib import MpskError
    rng = np.random.default_rng(77)
    m, n = 768, 384
    A1, A2 = rng.random((m, n)), rng.standard_normal((n, m))
    Qs, Rs = (be.download(t) for t in be.qrpos(be.upload(A1)))
    Ls, Qls = (be.download(t) for t in be.lqpos(be.upload(A2)))
    be.qr_defer()
    Q1, R1, L2, Q2 = (be.download(t) for t in be.qrlq_pair(be.upload(A1), be.upload(A2)))
    assert be.qr_commit() == 0                                    # nothing was left pending
    assert relerr(Q1, Qs) < 1e-12 and relerr(R1, Rs) < 1e-12 and relerr(L2, Ls) < 1e-11 and relerr(Q2, Qls) < 1e-10
    th = rng.standard_normal((512, 384))
    be.qr_defer()
    al, c, ar, k, _ = be.tsplit(be.upload(th), max_keep=100)
    assert be.qr_commit() == 0
    al, c, ar = be.download(al), be.download(c), be.download(ar)
    U, S, Vh = np.linalg.svd(th, full_matrices=False)
    assert relerr(al @ c @ ar, (U[:, :100] * S[:100]) @ Vh[:100]) < 1e-11
    # a pending deferral is refused by the composite calls
    be.qr_defer()
    d1, d2 = be.upload(A1), be.upload(A1 + 1.0)
    be.qrpos2(d1, d2)
    with pytest.raises(MpskError):
        be.qrlq_pair(be.upload(A1), be.upload(A2))
    with pytest.raises(MpskError):
        be.tsplit(be.upload(th), max_keep=100)
    be.qr_commit()


def test_cholqr_shift_retry(be):
    """The first CholeskyQR3 attempt uses a shift at the rounding level of the Gram matrix; on a matrix whose Gram matrix is
    numerically indefinite (cond 1e12) the device flags the breakdown and the factorization is repeated with the shift of
    Fukaya et al. -- same QRpos factors as Householder, counted by mpsk_ctx_qr_retries; a benign matrix needs no repeat."""
    rng = np.random.default_rng(8)
    m, n = 640, 192
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    r0 = be.qr_retries()
    Q, R = (be.download(t) for t in be.qrpos(be.upload(rng.standard_normal((m, n)))))
    assert be.qr_retries() == r0
    A = (U * np.logspace(0, -12, n)) @ V.T
    Q, R = (be.download(t) for t in be.qrpos(be.upload(A)))
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13
    assert np.abs(Q @ R - A).max() < 1e-14
    assert np.all(np.diag(R) > 0) and np.abs(np.tril(R, -1)).max() == 0.0
    s = be.qr_stats()
    assert be.qr_retries() > r0 or s["fallback"] > 0        # the rounding-level shift cannot have been enough here


@pytest.mark.parametrize("m,n,logc", [(65, 65, 2), (129, 65, 4), (640, 193, 5), (1100, 700, 7), (257, 256, 5), (900, 899, 2), (1493, 137, 10)])
def test_qr_ragged_shapes_in_step_solve(be, m, n, logc):
    """CholeskyQR3 with the in-step triangular solve (Q = X R^-1 as extra workgroups of the Cholesky step launches) on
    sizes that are not multiples of the 64-wide tiles, against LAPACK with the QRpos sign convention; LQpos and the
    two-stream pair go through the same path."""
    rng = np.random.default_rng(m * 7 + n)
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (U * np.logspace(0, -logc, n)) @ V.T
    Qr, Rr = np.linalg.qr(A)
    sg = np.sign(np.diag(Rr)); sg[sg == 0] = 1
    Qr = Qr * sg
    Q, R = (be.download(t) for t in be.qrpos(be.upload(A)))
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13
    assert np.abs(Q @ R - A).max() < 1e-14 and np.abs(np.tril(R, -1)).max() == 0.0 and np.all(np.diag(R) > 0)
    assert np.abs(Q - Qr).max() < 1e-8 * 10.0 ** logc * 1e-7 + 1e-13 or logc >= 10
    L, Ql = (be.download(t) for t in be.lqpos(be.upload(A.T.copy())))
    assert np.abs(Ql @ Ql.T - np.eye(n)).max() < 1e-13 and np.abs(L @ Ql - A.T).max() < 1e-14
    B = rng.standard_normal((m, n))
    Q1, R1, Q2, R2 = (be.download(t) for t in be.qrpos2(be.upload(A), be.upload(B)))
    assert np.abs(Q1 - Q).max() < 1e-12 and np.abs(Q2 @ R2 - B).max() < 1e-12 and np.abs(Q2.T @ Q2 - np.eye(n)).max() < 1e-13


def test_one_call_fixed_budget_eigsolve(be):
    """mpsk_hac_eigsolve_fixed (the site's whole fixed-budget solve in one library call) against the step-by-step host
    loop with the host Ritz step: same Ritz vector (positive overlap with the start), same captured first image."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import krylov
    rng = np.random.default_rng(17)
    D, d, W = 96, 2, 5
    H = mk.heisenberg_XXX(0.5, be=be)
    g = rng.standard_normal((W, D, D)); g = g + np.transpose(g, (0, 2, 1))
    GL = be.upload_env([m_[:, None, :] for m_ in g])
    GR = be.upload_env([m_[:, None, :] for m_ in g[::-1]])
    x0 = be.upload(rng.standard_normal((D, d, D)))
    for m in (1, 3, 8):
        h1 = mk.MPO_ddAC(be, H[1], GL, GR)
        f1 = be.empty(D, d, D)
        none, v1, _, _ = krylov.eigsolve_sr(be, h1, x0, fixed_matvecs=m, krylovdim=m, values=False, first_image=f1)
        assert none is None
        h2 = mk.MPO_ddAC(be, H[1], GL, GR)
        f2 = be.empty(D, d, D)
        lam, v2, _, _ = krylov.eigsolve_sr(be, lambda x, out=None: h2(x, out=out), x0, fixed_matvecs=m, krylovdim=m, first_image=f2)
        a, b = be.download(v1).ravel(), be.download(v2).ravel()
        assert abs(np.linalg.norm(a) - 1.0) < 1e-13
        assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-11, m
        assert a @ be.download(x0).ravel() > 0
        assert np.abs(be.download(f1) - be.download(f2)).max() < 1e-12 * np.abs(be.download(f2)).max()
