"""HIP path vs the committed golden fixtures (tests/golden/hotpath_vectors.npz) and size-independent
properties at the benchmark's full size (D = 1024): linearity, Hermiticity of the effective
Hamiltonian on symmetric environments, QR / transfer identities."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_golden_case_A(be):
    import mpskit_jl_amd as mk
    g = np.load(os.path.join(GOLD, "hotpath_vectors.npz"))
    H = mk.heisenberg_XXX(0.5, be=be)[0]
    GL = be.upload_env([a[:, None, :] for a in g["A_GL"][:, :, 0, :]])
    GR = be.upload_env([a[:, None, :] for a in g["A_GR"][:, :, 0, :]])
    assert relerr(be.download(be.dAC(H, GL, GR, be.upload(g["A_x"]))), g["A_dAC"]) < 1e-13
    assert relerr(be.download(be.dC(GL, GR, be.upload(g["A_c"]))), g["A_dC"]) < 1e-13
    assert relerr(be.download(be.dAC2(H, H, GL, GR, be.upload(g["A_x2"]))), g["A_dAC2"]) < 1e-13
    tl = be.download_env(be.transfer_left(H, GL, be.upload(g["A_A"]), be.upload(g["A_Ab"])), [1] * 5)
    tr = be.download_env(be.transfer_right(H, GR, be.upload(g["A_A"]), be.upload(g["A_Ab"])), [1] * 5)
    assert relerr(np.stack(tl), g["A_tl"]) < 1e-13 and relerr(np.stack(tr), g["A_tr"]) < 1e-13
    q, r = be.qrpos(be.upload(g["G_M"]))
    assert relerr(be.download(q), g["G_Q"]) < 1e-12 and relerr(be.download(r), g["G_R"]) < 1e-12
    l, qq = be.lqpos(be.upload(g["G_M"].T.copy()))
    assert relerr(be.download(l), g["G_L"]) < 1e-12 and relerr(be.download(qq), g["G_LQ"]) < 1e-12
    th = g["G_theta"]
    U, S, Vh, kept, disc = be.tsvd(be.upload(th).reshape(12, 14), max_keep=5)
    # oracle tsvd orders the columns (s2, b); singular values and discarded weight are order independent
    assert kept == 5 and relerr(be.download(S)[:5], g["G_S"]) < 1e-13 and abs(disc - float(g["G_err"])) < 1e-13


def test_golden_case_B_chi_gt_1(be):
    g = np.load(os.path.join(GOLD, "hotpath_vectors.npz"))
    chis = [int(c) for c in g["B_chis"]]
    blocks = {}
    for k in g.files:
        if k.startswith("B_O_"):
            _, _, i, j = k.split("_")
            blocks[(int(i), int(j))] = g[k]
    s = be.mposlice(len(chis), 2, chis, chis, blocks)
    GL = be.upload_env([g[f"B_GL{i}"] for i in range(len(chis))])
    GR = be.upload_env([g[f"B_GR{i}"] for i in range(len(chis))])
    assert relerr(be.download(be.dAC(s, GL, GR, be.upload(g["B_x"]))), g["B_dAC"]) < 1e-13


def test_full_size_properties(be):
    """D = 1024, d = 2, W = 5 (BASELINE north-star point): no oracle run, only identities."""
    import torch
    import mpskit_jl_amd as mk
    D, d, W = 1024, 2, 5
    # 5-level slice with SYMMETRIC d x d blocks (Sz, Sx) so that symmetric environments give a
    # symmetric effective Hamiltonian (the S+/S- blocks of heisenberg_XXX are each other's transpose)
    Sz = np.diag([0.5, -0.5]); Sx = np.array([[0.0, 0.5], [0.5, 0.0]])
    H = mk.MPOHamiltonian({(0, 0): 1.0, (4, 4): 1.0, (0, 1): Sz, (1, 4): Sz, (0, 2): Sx, (2, 4): Sx,
                           (0, 3): 0.3 * Sz, (3, 4): Sx + Sz, (0, 4): 0.7 * Sx}, be=be)[0]
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: be.upload(torch.rand(*s, generator=g, dtype=torch.float64).numpy() - 0.5)
    # symmetric environments -> the effective Hamiltonian is symmetric: <u, H v> = <H u, v>
    gl = torch.rand(W, D, D, generator=g, dtype=torch.float64).numpy() - 0.5
    gl = gl + np.transpose(gl, (0, 2, 1))
    GL = be.upload_env([m[:, None, :] for m in gl])
    GR = be.upload_env([m[:, None, :] for m in gl[::-1]])
    u, v = rnd(D, d, D), rnd(D, d, D)
    Hu, Hv = be.dAC(H, GL, GR, u), be.dAC(H, GL, GR, v)
    a, b = be.dot(u, Hv), be.dot(Hu, v)
    assert abs(a - b) < 1e-10 * max(abs(a), abs(b), 1.0) * D
    # linearity: H(2u - 3v) = 2 Hu - 3 Hv
    w = be.copy(u)
    be.axpby(-3.0, v, 2.0, w)
    Hw = be.dAC(H, GL, GR, w)
    be.axpby(-2.0, Hu, 1.0, Hw)
    be.axpby(3.0, Hv, 1.0, Hw)
    assert be.norm(Hw) < 1e-11 * be.norm(Hu)
    # QRpos identities at the sweep's size: Q^T Q = I, Q R = A (checked through traces / norms on device)
    A = rnd(2 * D, D)
    Q, R = be.qrpos(A)
    QtQ = be.gemm(Q, Q, transA=True)
    eye = be.upload(np.eye(D))
    be.axpby(-1.0, eye, 1.0, QtQ)
    assert be.norm(QtQ) < 1e-11
    QR = be.gemm(Q, R)
    be.axpby(-1.0, A, 1.0, QR)
    assert be.norm(QR) < 1e-11 * be.norm(A)
    # transfer of the identity through an isometry is the identity (test/states.jl:62-70)
    AL = mk.DTensor(Q.buf, (D, d, D))
    one = be.upload(np.eye(D)[None]).reshape(1, D, D)
    out = be.transfer_left(None, one, AL, AL)
    be.axpby(-1.0, eye, 1.0, mk.DTensor(out.buf, (D, D)))
    assert be.norm(mk.DTensor(out.buf, (D, D))) < 1e-11


def test_full_size_two_site_split_properties(be):
    """Size-independent identities of mpsk_tsplit at the config-4 tensor size (theta 4096 x 4096, keep 1024): al / ar
    isometries, |theta|^2 = |c|^2 + discarded^2 (Pythagoras for an orthogonal projection on singular subspaces), c
    triangular with the Schmidt values as singular values, al c ar = al al^T theta (projection on the kept LEFT subspace)."""
    import torch
    n, k = 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    A = A * torch.logspace(0, -6, n, dtype=torch.float64, device="cuda")[None, :]        # graded columns
    from mpskit_jl_amd import DTensor
    th = DTensor(A.T.contiguous().flatten(), (n, n))                                     # column-major copy of A
    al, c, ar, S, disc = be.tsplit(th, max_keep=k)
    alT = torch.as_strided(al.buf, (n, k), (1, n))
    arT = torch.as_strided(ar.buf, (k, n), (1, k))
    cT = torch.as_strided(c.buf, (k, k), (1, k))
    eye = torch.eye(k, dtype=torch.float64, device="cuda")
    assert float((alT.T @ alT - eye).abs().max()) < 1e-12
    assert float((arT @ arT.T - eye).abs().max()) < 1e-12
    tot2 = float((A * A).sum())
    assert abs(tot2 - float((cT * cT).sum()) - disc ** 2) < 1e-11 * tot2
    assert abs(float((cT * cT).sum()) - float(np.sum(S ** 2))) < 1e-11 * tot2
    # c is triangular: upper from QRpos(theta V_k), lower from LQpos(U_k^T theta) -- which one depends on the orientation
    # and on the preconditioning mode (double QR preconditioning yields the LEFT vectors of the tall orientation)
    assert float(torch.tril(cT, -1).abs().max()) == 0.0 or float(torch.triu(cT, 1).abs().max()) == 0.0
    proj = alT @ (alT.T @ A)
    assert float((alT @ cT @ arT - proj).abs().max()) < 1e-11 * float(A.abs().max())
    sv = torch.linalg.svdvals(cT).cpu().numpy()
    assert np.abs(sv - S).max() < 1e-12 * S[0]


def test_tsplit_4096_singular_values_pinned_by_lapack(be):
    """tsvd!(theta; trunc = truncdim(1024)) at the config-4 split size against LAPACK (dmrg.jl:96-104: the truncation acts
    on exactly these values).  tests/golden/tsplit_4096.npz holds numpy.linalg.svd's 4096 singular values and the discarded
    weight of a seeded graded theta with exact multiplets (one across the cut); theta is regenerated from the seed here.
    Absolute accuracy 1e-12 S[0] on every value, 1e-8 relative on the kept ones (LAPACK's own values are only good to
    ~1e-16 S[0] absolute, i.e. 1e-10 relative at S[1023] = 1.2e-6), discarded weight to 1e-9 relative, isometries."""
    import sys, os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gold)
    import make_sweep_traces as gen
    fx = np.load(os.path.join(gold, "tsplit_4096.npz"))
    Sref, k = fx["S"], int(fx["keep"])
    th = gen.tsplit_theta()
    assert np.abs(th.ravel()[:: 1048583][:16] - fx["theta_samples"]).max() < 1e-12        # same theta as the fixture's
    n = th.shape[0]
    d_th = be.upload(th)
    al, c, ar, S, disc = be.tsplit(d_th, max_keep=k)
    assert len(S) == k
    assert np.abs(S - Sref[:k]).max() <= 1e-12 * Sref[0]
    assert np.abs(S / Sref[:k] - 1.0).max() <= 1e-8
    assert abs(disc - float(fx["disc"])) <= 1e-9 * float(fx["disc"])
    A, Cm, B = be.download(al), be.download(c), be.download(ar)
    assert np.abs(A.T @ A - np.eye(k)).max() < 1e-12 and np.abs(B @ B.T - np.eye(k)).max() < 1e-12
    # al c ar == the rank-k truncation of theta up to the freedom inside the multiplet that straddles the cut:
    # |theta - al c ar|_F = discarded weight (optimal rank-k error) to 1e-9 relative
    assert abs(np.linalg.norm(th - A @ Cm @ B) - float(fx["disc"])) <= 1e-9 * float(fx["disc"]) + 1e-13 * float(fx["theta_fro"])
    # the full decomposition (mpsk_tsvd, rotations accumulated): every one of the 4096 values
    U, Sd, Vh, kept, _ = be.tsvd(d_th)
    Sf = be.download(Sd).ravel()
    assert kept == n and np.abs(Sf - Sref).max() <= 1e-12 * Sref[0]
