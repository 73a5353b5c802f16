"""Multi-GPU: bond-index sharding of the effective-Hamiltonian matvec (partition B).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  The output bond
index a' of  y[a',t,b'] = sum GL[w][a',a] x[a,s,b] O[w,t,s,v] GR[v][b,b']  is block-partitioned:
rank p owns rows [p*D/P, (p+1)*D/P) of every left-environment slab, computes y[a'_p,:,:] with the
SAME kernels (mpsk_dAC with Dlo = D/P: both GEMM stages shrink by 1/P) and ONE collective per
matvec (all-gather of D*d*D/P doubles per rank) completes y on every rank; no reduction, so every
rank holds bit-identical iterates and the replicated host-side Krylov loop stays in lock-step.
Everything else of the sweep (gauge steps, environment updates) is replicated in this round
("replicas" for those steps; DESIGN.md lists the fully sharded environment update as next).

The partition / gather-layout logic is independent of the device plumbing so that it is covered
by world_size-2 gloo tests on CPU (tests/test_dist_cpu.py) with a host stand-in for the plumbing.
"""
from __future__ import annotations

import numpy as np


class BondShard:
    """Equal row blocks of the bond index [0, D) over `world` ranks (requires world | D)."""

    def __init__(self, D: int, world: int, rank: int):
        if D % world != 0:
            raise ValueError(f"bond dimension {D} is not divisible by world size {world}")
        self.D, self.world, self.rank = D, world, rank
        self.block = D // world

    @property
    def lo(self):
        return self.rank * self.block

    @property
    def hi(self):
        return self.lo + self.block

    @staticmethod
    def shardable(D, world, min_block=64, force=False):
        return (world > 1 or force) and D % world == 0 and D // world >= min_block


class HostPlumbing:
    """NumPy stand-in for the device plumbing (used by the CPU gloo tests only)."""

    def __init__(self, local_dAC):
        self._dAC = local_dAC

    def row_block(self, env, lo, hi):          # env: ndarray (W, Dbra, Dket)
        return np.ascontiguousarray(env[:, lo:hi, :])

    def local_dAC(self, H, GLloc, GR, x):
        return self._dAC(H, GLloc, GR, x)

    def all_gather_rows(self, yloc, group, world):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(yloc))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        return np.concatenate([o.numpy() for o in outs], axis=0)


class DevicePlumbing:
    """Device implementation: copy2d row-block extraction, mpsk_dAC on the local block, RCCL
    all-gather through torch.distributed on the ctx stream, copy2d re-interleave."""

    def __init__(self, be):
        self.be = be

    def row_block(self, env, lo, hi):
        W, Db, Dk = env.shape
        n = hi - lo
        out = self.be.empty(W, n, Dk)
        # all W slabs at once: rows [lo, hi) of a (Db x W*Dk) column-major matrix
        self.be.copy2d(n, W * Dk, env.ptr + 8 * lo, Db, out.ptr, n)
        return out

    def local_dAC(self, H, GLloc, GR, x):
        return self.be.dAC(H, GLloc, GR, x)

    def row_block_tensor(self, A, lo, hi):
        """rows [lo, hi) of the first index of A[a, s, b]  (column-major: a (Dl x d*Dr) matrix)."""
        Dl, d, Dr = A.shape
        n = hi - lo
        out = self.be.empty(n, d, Dr)
        self.be.copy2d(n, d * Dr, A.ptr + 8 * lo, Dl, out.ptr, n)
        return out

    def _all_reduce_flat(self, t, group):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    def all_reduce_sum(self, part, group, world):
        self._all_reduce_flat(part.buf[: part.size], group)
        return part

    def all_gather_cols(self, cols, group, world):
        """cols: (W, Dl, n) = the column block of this rank for every slab -> (W, Dl, n * world)."""
        W, Dl, n = cols.shape
        gathered = self._all_gather_flat(cols, group, world)
        out = self.be.empty(W, Dl, n * world)
        for p in range(world):   # rank p, slab w: a contiguous Dl*n chunk -> columns [p*n, (p+1)*n) of slab w
            self.be.copy2d(Dl * n, W, gathered.data_ptr() + 8 * p * cols.size, Dl * n,
                           out.ptr + 8 * p * Dl * n, Dl * n * world)
        return out

    def _all_gather_flat(self, yloc, group, world):
        """the collective itself: rank blocks back to back in one device buffer (RCCL on the current stream)."""
        import torch
        import torch.distributed as dist
        gathered = torch.empty(world * yloc.size, dtype=torch.float64, device=self.be.device)
        dist.all_gather_into_tensor(gathered, yloc.buf[: yloc.size], group=group)
        return gathered

    def all_gather_rows(self, yloc, group, world):
        n, d, Dr = yloc.shape
        gathered = self._all_gather_flat(yloc, group, world)
        y = self.be.empty(n * world, d, Dr)
        for p in range(world):   # rank p's block -> rows [p*n, (p+1)*n) of every (s, b) column
            self.be.copy2d(n, d * Dr, gathered.data_ptr() + 8 * p * yloc.size, n, y.ptr + 8 * p * n, n * world)
        return y


class ShardedTransfer:
    """Environment updates with the compute sharded over the process group ("A only where the
    contraction index must be split (env updates with sharded AL)").  Storage stays replicated in this version.

    transfer_left :  GL'[v][b', b] = sum_{a', s', a, s, w} Ab[a', s', b'] GL[w][a', a] A[a, s, b] O[w, s', s, v].
        Rank p takes the rows a' in its block of GL AND of Ab (all three stages shrink by 1/P) -> a partial GL'
        of full shape; ONE all-reduce (sum) completes it on every rank.
    transfer_right:  GR'[w][a, a'] = sum A[a, s, b] GR[v][b, b'] Ab[a', s', b'] O[w, s, s', v].
        Rank p takes the rows a' in its block of Ab -> the COLUMNS a'_p of every output slab, no reduction;
        ONE all-gather of the column blocks + a device re-interleave completes GR'.
    Both leave bit-identical environments on every rank (the collectives return the same bits everywhere)."""

    def __init__(self, plumbing, world, rank, group=None, min_block=64, force=False):
        self.pl, self.be = plumbing, plumbing.be
        self.world, self.rank, self.group, self.min_block, self.force = world, rank, group, min_block, force
        self.n_collectives = 0

    def _ok(self, D):
        return BondShard.shardable(D, self.world, self.min_block, self.force)

    def transfer_left(self, H, GLin, A, Ab):
        be = self.be
        Dlb = Ab.shape[0]
        if H is None or not self._ok(Dlb):
            return be.transfer_left(H, GLin, A, Ab)
        sh = BondShard(Dlb, self.world, self.rank)
        GLloc = self.pl.row_block(GLin, sh.lo, sh.hi)                       # (W, Dlo, Dl)
        Abloc = self.pl.row_block_tensor(Ab, sh.lo, sh.hi)                  # (Dlo, d, Drb)
        part = be.transfer_left(H, GLloc, A, Abloc)
        self.n_collectives += 1
        return self.pl.all_reduce_sum(part, self.group, self.world)

    def transfer_right(self, H, GRin, A, Ab):
        be = self.be
        Dlb = Ab.shape[0]
        if H is None or not self._ok(Dlb):
            return be.transfer_right(H, GRin, A, Ab)
        sh = BondShard(Dlb, self.world, self.rank)
        Abloc = self.pl.row_block_tensor(Ab, sh.lo, sh.hi)
        cols = be.transfer_right(H, GRin, A, Abloc)                         # (W, Dl, Dlo): columns a'_p
        self.n_collectives += 1
        return self.pl.all_gather_cols(cols, self.group, self.world)


class HostStagedPlumbing(DevicePlumbing):
    """DevicePlumbing whose collective is staged through the host over gloo: lets TWO ranks share ONE GPU (RCCL
    refuses duplicate devices), so the device-side row-block extraction and re-interleave are exercised at
    world_size 2 on a single-GPU box (tests/test_gpu_dist.py).  Test plumbing only."""

    def _all_gather_flat(self, yloc, group, world):
        import torch
        import torch.distributed as dist
        h = yloc.buf[: yloc.size].cpu()
        outs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(outs, h, group=group)
        return torch.cat(outs).to(self.be.device)

    def _all_reduce_flat(self, t, group):
        import torch.distributed as dist
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h.to(t.device))
        return t


class ShardedMatvec:
    """Callable y = H_AC x with the bond index sharded over the process group."""

    def __init__(self, plumbing, H, GL, GR, world, rank, group=None):
        self.pl, self.H, self.GR = plumbing, H, GR
        self.world, self.rank, self.group = world, rank, group
        D = GL.shape[1]
        self.shard = BondShard(D, world, rank)
        self.GLloc = plumbing.row_block(GL, self.shard.lo, self.shard.hi)
        self.n_collectives = 0

    def __call__(self, x, out=None):
        yloc = self.pl.local_dAC(self.H, self.GLloc, self.GR, x)
        y = self.pl.all_gather_rows(yloc, self.group, self.world)
        self.n_collectives += 1
        if out is not None:
            self.pl.be.axpby(1.0, y, 0.0, out)
            return out
        return y


def shard_wrapper(be, world, rank, group=None, min_block=64, force=False, plumbing=None):
    """Returns wrap(h: MPO_ddAC) -> callable used by dmrg_sweep: sites whose bond dimension is
    shardable run the sharded matvec, the others (chain edges) run replicated."""
    pl = DevicePlumbing(be) if plumbing is None else plumbing

    def wrap(h):
        D = h.leftenv.shape[1]
        if not BondShard.shardable(D, world, min_block, force):
            return h
        return ShardedMatvec(pl, h.o, h.leftenv, h.rightenv, world, rank, group)
    return wrap
