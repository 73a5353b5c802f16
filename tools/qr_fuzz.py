"""Randomised shapes through mpsk_qrpos / mpsk_lqpos / mpsk_qrpos2 against LAPACK (positive-diagonal convention):
the CholeskyQR3 path with the in-step triangular solve on ragged sizes (m, n not multiples of 64, m == n, n just above 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpskit_jl_amd as mk
be = mk.Backend(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def ref_qr(A):
    Q, R = np.linalg.qr(A)
    s = np.sign(np.diag(R)); s[s == 0] = 1
    return Q * s, (R.T * s).T
worst = 0.0
shapes = [(65, 65), (129, 65), (200, 130), (640, 193), (1000, 333), (1100, 700), (257, 256), (2048, 1024), (1536, 1530), (900, 899)]
shapes += [(int(m), int(n)) for m, n in zip(rng.integers(70, 1800, 12), rng.integers(66, 900, 12)) if m >= n]
for (m, n) in shapes:
    cond = 10.0 ** rng.uniform(0, 7)
    U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (U * np.logspace(0, -np.log10(cond), n)) @ V.T
    Qr, Rr = ref_qr(A)
    Q, R = (be.download(t) for t in be.qrpos(be.upload(A)))
    e_orth = np.abs(Q.T @ Q - np.eye(n)).max(); e_rec = np.abs(Q @ R - A).max() / np.abs(A).max()
    e_q = np.abs(Q - Qr).max(); e_tri = np.abs(np.tril(R, -1)).max()
    L, Ql = (be.download(t) for t in be.lqpos(be.upload(A.T.copy())))
    e_lq = max(np.abs(Ql @ Ql.T - np.eye(n)).max(), np.abs(L @ Ql - A.T).max() / np.abs(A).max())
    B = rng.standard_normal((m, n))
    Q1, R1, Q2, R2 = (be.download(t) for t in be.qrpos2(be.upload(A), be.upload(B)))
    e_pair = max(np.abs(Q1 - Q).max(), np.abs(Q2 @ R2 - B).max(), np.abs(Q2.T @ Q2 - np.eye(n)).max())
    ok = e_orth < 1e-12 and e_rec < 1e-13 and e_tri == 0.0 and e_lq < 1e-12 and e_pair < 1e-11 and e_q < 1e-9 * cond
    worst = max(worst, e_orth, e_rec, e_lq)
    print(f"{m:5d} x {n:4d} cond {cond:8.1e}: orth {e_orth:.1e} rec {e_rec:.1e} |Q-Qref| {e_q:.1e} lq {e_lq:.1e} pair {e_pair:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    assert ok
print("all ok, worst", worst, be.qr_stats(), "retries", be.qr_retries())
