"""Randomised check of the default (truncation-aware, svd mode 3) two-site split against LAPACK: shapes, kept ranks and spectra
drawn at random -- graded, clustered (exact multiplets, one across the cut), rank deficient, flat, with and without a truncerr.
usage: python tools/split_fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpskit_jl_amd as mk
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {"S": 0.0, "orth": 0.0, "opt": 0.0, "disc": 0.0}
paths = {0: 0, 1: 0, 2: 0}
for case in range(ncase):
    be = mk.Backend(0)                        # fresh ctx per case: no back-off / iteration hints / grown workspace carried over
    m, n = (int(rng.integers(130, 1400)) for _ in range(2))
    r = min(m, n)
    kind = ["graded", "cluster", "rankdef", "flat", "shelf"][case % 5]
    k = int(rng.integers(8, max(9, r // 2)))
    if kind == "graded":
        s = np.logspace(0, -float(rng.uniform(2, 12)), r)
    elif kind == "cluster":
        s = np.logspace(0, -6, r); g = int(rng.integers(2, 6)); j = max(0, k - g // 2 - 1); s[j:j + g] = s[j]       # multiplet across the cut
        s = np.sort(s)[::-1]
    elif kind == "rankdef":
        s = np.logspace(0, -5, r); s[int(rng.integers(k // 2 + 1, r)):] = 0.0
    elif kind == "flat":
        s = np.sort(rng.uniform(0.5, 1.0, r))[::-1]
    else:                                                               # decaying kept part + a flat shelf behind the cut
        s = np.logspace(0, -4, r); s[k:] = s[k - 1] * rng.uniform(0.3, 0.99) * np.sort(rng.uniform(0.5, 1.0, r - k))[::-1]
    U, _ = np.linalg.qr(rng.standard_normal((m, r))); V, _ = np.linalg.qr(rng.standard_normal((n, r)))
    A = (U * s) @ V.T
    te = float(rng.choice([0.0, 0.0, 10.0 ** -rng.uniform(3, 8)]))
    al, c, ar, S, disc = be.tsplit(be.upload(A), max_keep=k, trunc_err=te)
    st = be.split_stats(); paths[st["path"]] += 1
    kk = len(S)
    if te > 0:                                                          # the truncerr rule on the exact values
        kx = k
        while kx > 1 and np.linalg.norm(s[kx - 1:]) <= te: kx -= 1
        assert kk == kx, (case, kind, kk, kx)
    else:
        assert kk == k
    a_, c_, r_ = be.download(al), be.download(c), be.download(ar)
    eS = np.abs(S - s[:kk]).max() / s[0]
    eo = max(np.abs(a_.T @ a_ - np.eye(kk)).max(), np.abs(r_ @ r_.T - np.eye(kk)).max())
    best = np.linalg.norm(s[kk:])
    eopt = abs(np.linalg.norm(A - a_ @ c_ @ r_) - best) / s[0]
    ed = abs(disc - best) / s[0]
    for key, v in (("S", eS), ("orth", eo), ("opt", eopt), ("disc", ed)): worst[key] = max(worst[key], v)
    flag = "" if max(eS, eo, eopt, ed) < 1e-11 else "   <-- CHECK"
    be.close()
    print(f"case {case:3d} {kind:8s} {m:5d} x {n:5d} keep {kk:4d} (max {k}, truncerr {te:.0e}) path {st['path']} iters {st['iterations']:2d}  "
          f"|S| {eS:.1e} orth {eo:.1e} |rec|-opt {eopt:.1e} disc {ed:.1e}{flag}", flush=True)
print("worst", worst, "paths", paths)
assert max(worst.values()) < 1e-11
