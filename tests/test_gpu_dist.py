"""World-size-2 check of the DEVICE-side sharding plumbing on a single-GPU box: two rank processes share GPU 0 and the
collective is staged through the host over gloo (RCCL refuses two ranks on one device), so everything around the
collective -- row-block extraction, mpsk_dAC with Dlo = D / P, re-interleave, the sharded sweep -- is the code the
multi-GPU bench runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_host_staged():
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dist_gpu_check.py"), str(r), "2", port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-2000:]}"
        assert "OK" in o


def test_libmpsk_comm_world1(be):
    """include/mpsk_comm.h on the GPU at world size 1 (the only size a one-GPU box allows: RCCL refuses duplicate
    devices): RCCL communicator from the C ABI, in-place all-gather / all-reduce on the ctx stream, and one sharded
    application of a prepared operator (mpsk_comm_hac_apply) == mpsk_dAC.  The sharded sweep through this communicator
    equals the unsharded sweep."""
    import numpy as np
    import torch
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import dist as md, algorithms as alg, krylov
    comm = md.LibComm(be, 1, 0)
    try:
        t = torch.arange(1000, dtype=torch.float64, device=be.device)
        ref = t.clone()
        comm.all_gather_into(t, t)                  # in place
        comm.all_reduce_sum(t)
        rs = torch.empty_like(t)
        comm.reduce_scatter_sum(rs, t)              # world 1: the single chunk comes back unchanged
        be.synchronize()
        assert torch.equal(t, ref) and torch.equal(rs, ref)
        rng = np.random.default_rng(4)
        D, d, W = 128, 2, 5
        H = mk.heisenberg_XXX(0.5, be=be)
        GL = be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)])
        GR = be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)])
        x = be.upload(rng.standard_normal((D, d, D)))
        hac = be.hac_create(H[1], GL, GR)
        y = comm.hac_apply(hac, md.to_blocked(be, x, 1), be.empty(D, d, D))
        yref = be.dAC(H[1], GL, GR, x)
        assert np.abs(be.download(y) - be.download(yref)).max() <= 1e-13 * np.abs(be.download(yref)).max()
        psi = mk.FiniteMPS.random(12, 2, 64, np.random.default_rng(1), be=be)
        ps = psi.copy()
        eig = mk.Arnoldi(fixed_matvecs=4, krylovdim=4)
        eu, es = mk.FinEnv(psi, H), md.ShardedFinEnv(ps, H, comm, min_block=32, force=True)
        alg.dmrg_sweep(psi, H, eu, eig, krylov.KrylovWorkspace(be))
        alg.dmrg_sweep(ps, H, es, eig, krylov.KrylovWorkspace(be))
        e1 = float(np.sum(mk.expectation_value(psi, H, eu)))
        e2 = float(np.sum(mk.expectation_value(ps, H, es)))
        assert abs(e1 - e2) <= 1e-10 * abs(e1) and comm.n_allgather > 10 and comm.n_reduce_scatter > 3
    finally:
        comm.close()


def test_torch_rccl_world1_sharded_sweep(be):
    """dist.Comm over torch.distributed's RCCL backend at world size 1 (force_collective: every all-gather / all-reduce
    of the sharded sweep is really issued): the collectives are ordered against the ctx stream only through stream
    semantics -- the sweep has no host synchronisation left that could hide a missing dependency -- so the sharded sweep
    must reproduce the unsharded one."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import dist as md, algorithms as alg, krylov
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = False
    if not dist.is_initialized():
        port = 29900 + os.getpid() % 90
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        comm = md.Comm(1, 0, force_collective=True)
        H = mk.heisenberg_XXX(0.5, be=be)
        psi = mk.FiniteMPS.random(14, 2, 128, np.random.default_rng(5), be=be)
        ps = psi.copy()
        eig = mk.Arnoldi(fixed_matvecs=6, krylovdim=6)
        eu, es = mk.FinEnv(psi, H), md.ShardedFinEnv(ps, H, comm, min_block=32, force=True)
        for _ in range(2):
            e_u = alg.dmrg_sweep(psi, H, eu, eig, krylov.KrylovWorkspace(be))
            e_s = alg.dmrg_sweep(ps, H, es, eig, krylov.KrylovWorkspace(be))
        e1 = float(np.sum(mk.expectation_value(psi, H, eu)))
        e2 = float(np.sum(mk.expectation_value(ps, H, es)))
        assert abs(e1 - e2) <= 1e-10 * abs(e1), (e1, e2)
        assert np.abs(np.array(e_u) - np.array(e_s)).max() <= 1e-8
        # (left-environment updates onto a sharded bond: one ncclReduceScatter each; an all-reduce only where the output
        #  bond is too small to shard)
        assert comm.n_allgather > 50 and comm.n_reduce_scatter > 10 and comm.n_allreduce >= 1
    finally:
        if created:
            dist.destroy_process_group()
