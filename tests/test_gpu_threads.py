"""Thread-safety of the library's process-wide state (include/mpsk.h, conventions): the reference applies its
effective Hamiltonians from several Julia tasks at once -- the AC and C eigensolves of one VUMPS site run concurrently
(vumps.jl:39-49), the left / right environment solves too (vumps.jl:78-86).  Two ctxs on two streams driven from two
host threads must give the SAME BITS as the serial run on one ctx."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _site_problem(seed, D, d):
    rng = np.random.default_rng(seed)
    W = 5
    sym = lambda a: a + np.transpose(a, (0, 2, 1))
    GL = sym(rng.random((W, D, D)) - 0.5)
    GR = sym(rng.random((W, D, D)) - 0.5)
    return GL, GR, rng.random((D, d, D)) - 0.5, rng.random((D, D)) - 0.5


def _solve(mk, be, which, GL, GR, x0, reps):
    """`reps` eigensolves (fixed 6-matvec budget: no data-dependent control flow) of H_AC or H_C; returns host copies."""
    from mpskit_jl_amd import krylov
    H = mk.heisenberg_XXX(0.5, be=be)[0]
    gl = be.upload_env([m[:, None, :] for m in GL])
    gr = be.upload_env([m[:, None, :] for m in GR])
    x = be.upload(x0)
    ws = krylov.KrylovWorkspace(be)
    out = []
    for _ in range(reps):
        if which == "AC":
            mv = lambda v, o: be.dAC(H, gl, gr, v, out=o)
        else:
            mv = lambda v, o: be.dC(gl, gr, v, out=o)
        lam, vec, _, _ = krylov.eigsolve_sr(be, mv, x, fixed_matvecs=6, krylovdim=6, ws=ws)
        q, r = be.qrpos(vec.reshape(vec.size // vec.shape[-1], vec.shape[-1]))   # gauge step on the same ctx
        out.append((lam, be.download(vec), be.download(r)))
        x = vec
    return out


@pytest.mark.parametrize("D", [96, 256])
def test_two_ctxs_two_threads_bit_identical_to_serial(be, D):
    import torch
    import mpskit_jl_amd as mk
    d, reps = 2, 4
    GL, GR, x0, c0 = _site_problem(7, D, d)
    serial = {"AC": _solve(mk, be, "AC", GL, GR, x0, reps), "C": _solve(mk, be, "C", GL, GR, c0, reps)}
    results, errors = {}, []

    def worker(which, start):
        try:
            s = torch.cuda.Stream(device=0)
            with torch.cuda.stream(s):
                b = mk.Backend(0)
                try:
                    results[which] = _solve(mk, b, which, GL, GR, start, reps)
                    b.synchronize()
                finally:
                    b.close()
        except Exception as e:  # noqa: BLE001
            errors.append((which, repr(e)))

    for attempt in range(3):                      # several rounds: races are timing dependent
        results.clear()
        ts = [threading.Thread(target=worker, args=("AC", x0)), threading.Thread(target=worker, args=("C", c0))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errors, errors
        for which in ("AC", "C"):
            assert len(results[which]) == reps
            for (l1, v1, r1), (l2, v2, r2) in zip(serial[which], results[which]):
                assert l1 == l2, (which, attempt, l1, l2)
                assert np.array_equal(v1, v2) and np.array_equal(r1, r2), (which, attempt)


def test_ctx_destroy_releases_stream_workspace(be):
    """split-K partial-tile workspaces are keyed by (device, stream) and released with the ctx that used the stream:
    creating and destroying many ctxs that each run a split-K GEMM must not grow device memory."""
    import torch
    import mpskit_jl_amd as mk
    D, d = 256, 2          # stage 3 of dAC at D = 256 takes the split-K path (<= 512 tiles of 64x64, >= 64 k-tiles)
    GL, GR, x0, _ = _site_problem(3, D, d)

    def once():
        s = torch.cuda.Stream(device=0)
        with torch.cuda.stream(s):
            b = mk.Backend(0)
            H = mk.heisenberg_XXX(0.5, be=b)[0]
            y = b.dAC(H, b.upload_env([m[:, None, :] for m in GL]), b.upload_env([m[:, None, :] for m in GR]), b.upload(x0))
            b.synchronize()
            del y, H
            b.close()

    def free_bytes():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()        # torch caches blocks per stream: not what is measured here
        return torch.cuda.mem_get_info()[0]

    once()
    free0 = free_bytes()
    for _ in range(12):
        once()
    free1 = free_bytes()
    assert free0 - free1 < 96 * 2 ** 20, (free0 - free1) / 2 ** 20     # one leaked 64 MiB slot per ctx would be 768 MiB
