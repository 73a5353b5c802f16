import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, mpskit_jl_amd as mk
from mpskit_jl_amd._lib import MpskError
be = mk.Backend(0)
rng = np.random.default_rng(1)
n, k = 768, 128
for name, A in (("zero", np.zeros((n, n))), ("rank1", np.outer(rng.standard_normal(n), rng.standard_normal(n))),
                ("rank5", rng.standard_normal((n, 5)) @ rng.standard_normal((5, n))), ("tiny", 1e-200 * rng.standard_normal((n, n))),
                ("huge", 1e150 * rng.standard_normal((n, n)) @ np.diag(np.logspace(0, -8, n)))):
    try:
        al, c, ar, S, disc = be.tsplit(be.upload(A), max_keep=k)
        a_, c_, r_ = be.download(al), be.download(c), be.download(ar)
        sref = np.linalg.svd(A, compute_uv=False)
        sc = max(sref[0], 1e-300)
        print(name, be.split_stats(), "S err", np.abs(S - sref[:k]).max() / sc, "orth", np.abs(a_.T @ a_ - np.eye(k)).max(), np.abs(r_ @ r_.T - np.eye(k)).max(),
              "rec", abs(np.linalg.norm(A - a_ @ c_ @ r_) - np.linalg.norm(sref[k:])) / sc, flush=True)
    except MpskError as e:
        print(name, "error:", str(e)[:120], flush=True)
A = rng.standard_normal((n, n)); A[3, 5] = np.nan
try:
    be.tsplit(be.upload(A), max_keep=k); print("nan: returned")
except MpskError as e:
    print("nan error:", str(e)[:120])
