"""world_size-2 gloo tests (CPU) of the bond-sharded matvec path (mpskit.jl_amd/dist.py): partition
logic, row-block extraction, all-gather layout and the lock-step of a sharded DMRG sweep."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mpskit_oracle as mo
        import mpskit_jl_amd as mk
        from mpskit_jl_amd import dist as mdist, algorithms as alg, krylov
        from cpu_backend import CpuBackend

        rng = np.random.default_rng(11)          # same seed on both ranks -> replicas
        D, d = 8, 2
        Ho = mo.heisenberg_mpo(0.5)[0]
        GL = [rng.standard_normal((D, 1, D)) for _ in range(5)]
        GR = [rng.standard_normal((D, 1, D)) for _ in range(5)]
        x = rng.standard_normal((D, d, D))
        full = mo.dAC(x, Ho, GL, GR)

        # (a) partition logic with the host plumbing + oracle arithmetic
        pl = mdist.HostPlumbing(lambda H, gl, gr, xx: mo.dAC(xx, H, [gl[w][:, None, :] for w in range(5)], GR))
        GLs = np.stack([g[:, 0, :] for g in GL])                     # (W, D, D)
        sm = mdist.ShardedMatvec(pl, Ho, GLs, None, world, rank)
        assert sm.shard.block == D // world and sm.shard.lo == rank * D // world
        ya = sm(x)
        ok_a = bool(np.abs(ya - full).max() < 1e-12) and sm.n_collectives == 1

        # (b) the REAL device plumbing class (copy2d row blocks, all_gather_into_tensor, re-interleave)
        cb = CpuBackend()
        Hg = mk.heisenberg_XXX(0.5, be=cb)
        dGL, dGR, dx = cb.upload_env(GL), cb.upload_env(GR), cb.upload(x)
        smb = mdist.ShardedMatvec(mdist.DevicePlumbing(cb), Hg[0], dGL, dGR, world, rank)
        yb = cb.download(smb(dx))
        ok_b = bool(np.abs(yb - full).max() < 1e-12)
        glloc = cb.download_env(smb.GLloc, [1] * 5)
        ok_b = ok_b and all(np.abs(glloc[w][:, 0, :] - GL[w][smb.shard.lo:smb.shard.hi, 0, :]).max() == 0 for w in range(5))

        # (c) a sharded DMRG sweep stays in lock-step and equals the unsharded sweep
        L, Dm = 8, 8
        dims = mo.FiniteMPS.random(L, d, Dm, np.random.default_rng(0)).bond_dims()
        As = [rng.random((1 if i == 0 else dims[i - 1], d, dims[i])) for i in range(L)]
        eig = mk.Arnoldi(tol=1e-10, krylovdim=10)
        wrap = mdist.shard_wrapper(cb, world, rank, min_block=2)
        ps, pu = mk.FiniteMPS(As, normalize=True, be=cb), mk.FiniteMPS(As, normalize=True, be=cb)
        es, eu = mk.FinEnv(ps, Hg), mk.FinEnv(pu, Hg)
        for _ in range(3):
            alg.dmrg_sweep(ps, Hg, es, eig, krylov.KrylovWorkspace(cb), wrap)
            alg.dmrg_sweep(pu, Hg, eu, eig, krylov.KrylovWorkspace(cb), None)
        Es = float(np.sum(mk.expectation_value(ps, Hg, es)))
        Eu = float(np.sum(mk.expectation_value(pu, Hg, eu)))
        t = torch.tensor([Es], dtype=torch.float64)
        outs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, t)
        ok_c = abs(Es - Eu) < 1e-10 * abs(Eu) and all(float(o) == Es for o in outs)

        # (d) sharded environment updates (dist.ShardedTransfer): all-reduce / all-gather versions == the kernels,
        #     and a sweep whose FinEnv uses them stays in lock-step with the unsharded one
        st = mdist.ShardedTransfer(mdist.DevicePlumbing(cb), world, rank, min_block=2)
        A = cb.upload(rng.standard_normal((D, d, D)))
        tl_s, tl_u = cb.download(st.transfer_left(Hg[1], dGL, A, A)), cb.download(cb.transfer_left(Hg[1], dGL, A, A))
        tr_s, tr_u = cb.download(st.transfer_right(Hg[1], dGR, A, A)), cb.download(cb.transfer_right(Hg[1], dGR, A, A))
        ok_d = bool(np.abs(tl_s - tl_u).max() < 1e-12 * np.abs(tl_u).max() and np.abs(tr_s - tr_u).max() < 1e-12 * np.abs(tr_u).max())
        ok_d = ok_d and st.n_collectives == 2
        pd = mk.FiniteMPS(As, normalize=True, be=cb)
        ed = mk.FinEnv(pd, Hg, transfer_ops=st)
        for _ in range(3):
            alg.dmrg_sweep(pd, Hg, ed, eig, krylov.KrylovWorkspace(cb), wrap)
        Ed = float(np.sum(mk.expectation_value(pd, Hg, ed)))
        t = torch.tensor([Ed], dtype=torch.float64)
        dist.all_gather(outs, t)
        ok_d = ok_d and abs(Ed - Eu) < 1e-10 * abs(Eu) and all(float(o) == Ed for o in outs)
        ret[rank] = (ok_a, ok_b, ok_c and ok_d, Es, Eu)
    finally:
        dist.destroy_process_group()


def test_sharded_matvec_world2_gloo():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok_a, ok_b, ok_c, Es, Eu = ret[r]
        assert ok_a, "host-plumbing sharded matvec != full matvec"
        assert ok_b, "device-plumbing class (copy2d + all_gather_into_tensor) != full matvec"
        assert ok_c, f"sharded sweep diverged: {Es} vs {Eu}"


def test_bondshard_partition():
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mpskit_jl_amd.dist import BondShard
    s = [BondShard(1024, 8, r) for r in range(8)]
    assert [x.lo for x in s] == [128 * r for r in range(8)] and s[-1].hi == 1024
    assert BondShard.shardable(1024, 8) and not BondShard.shardable(1024, 1)
    assert not BondShard.shardable(100, 8) and not BondShard.shardable(128, 8, min_block=64)
    with pytest.raises(ValueError):
        BondShard(10, 4, 0)
