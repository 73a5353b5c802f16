"""Multi-GPU: bond-index sharding of the DMRG site update, one process per GPU.

Partition (B of the survey): the OUTPUT bond index a' of
    y[a',t,b'] = sum GL[w][a',a] x[a,s,b] O[w,t,s,v] GR[v][b,b']
is block-partitioned over the P ranks.  What lives where (D = bond dimension, n = D / P, p = rank):

  left environments   permanently ROW-sharded: rank p stores GL[w][a'_p, :]  (W, n, D)        -> 1/P of the memory
  right environments  stored COLUMN-sharded over the bra index: GR[w][:, a'_p]  (W, D, n)     -> 1/P of the memory;
                      a full copy is all-gathered ONCE PER SITE VISIT (stage 3 of the matvec reads all of GR) into a
                      transient buffer (at most two alive): ONE collective into the rank-major layout [P][W][D][n], then P
                      strided device copies into the slab layout [W][D][D]
  Krylov vectors      replicated, in the BLOCKED layout [P][n, d, D] (rank blocks back to back): the local matvec
                      (mpsk_dAC_blocked: the blocks are K-segments of the stage-1 GEMM) writes this rank's block
                      straight into the destination vector and ONE in-place all-gather completes it -- no per-matvec
                      allocation, no re-interleave kernels; vector arithmetic (dots, axpys, Gram-Schmidt) is layout
                      agnostic.  A tensor is converted rows <-> blocks once per site visit (encode / decode).
  environment update  transfer_left : all three stages on the local rows; the partial result (contraction over the
                      sharded a') is re-ordered into rank-major row blocks [P][W][n][D] (one device copy) and completed
                      by ONE reduce-scatter -- every rank receives exactly the rows it stores, half the traffic of the
                      all-reduce + slice of round 2 (an all-reduce remains where the output bond is too small to shard);
                      transfer_right: rank p computes the bra columns a'_p from the gathered input -- no reduction.
  gauge steps (QRpos / LQpos), state tensors: replicated ("replicas only" for those steps).

No reduction touches the Krylov vectors, so every rank holds bit-identical iterates and the replicated host-side
Krylov loop stays in lock-step.  Chain-edge bonds that are too small to shard (D < P * min_block) keep replicated
environments and run the plain matvec.

Collectives go through `Comm` (torch.distributed: backend "nccl" == RCCL over xGMI; "gloo" on CPU for the
world_size-2 tests).  A host without torch.distributed (the Julia shim) makes the same calls through
libmpsk_comm (include/mpsk_comm.h: mpsk_comm_allgather / mpsk_comm_allreduce_sum on the ctx stream).
"""
from __future__ import annotations

import numpy as np

from .backend import DTensor
from .derivatives import MPO_ddAC


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


class Comm:
    """The two collectives of the sharded sweep on flat fp64 buffers (torch tensors: device memory on the GPU path,
    host memory under the CPU stand-in).  staged=True stages them through the host over gloo so that TWO ranks can
    share ONE GPU in tests (RCCL refuses duplicate devices)."""

    def __init__(self, world=1, rank=0, group=None, staged=False, force_collective=False):
        """force_collective: issue the torch.distributed calls even at world size 1 (exercises the RCCL path on a
        one-GPU box; needs an initialised process group)."""
        self.world, self.rank, self.group, self.staged = int(world), int(rank), group, staged
        self.force_collective = force_collective
        self.n_allgather = self.n_allreduce = self.n_reduce_scatter = 0
        self.bytes_allgather = self.bytes_allreduce = self.bytes_reduce_scatter = 0
        self._inplace = None

    def _dist(self):
        import torch.distributed as dist
        return dist

    def all_gather_into(self, out, inp):
        """out[r * len(inp) : (r + 1) * len(inp)] = rank r's inp.  `inp` may be the slice of `out` that belongs to
        this rank (in place, what the matvec does)."""
        self.n_allgather += 1
        self.bytes_allgather += out.numel() * 8
        if self.world == 1 and not self.force_collective:
            n = inp.numel()
            if out.data_ptr() + 8 * self.rank * n != inp.data_ptr():
                out[:n].copy_(inp)
            return out
        dist = self._dist()
        if self.staged:
            import torch
            h = inp.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            out.copy_(torch.cat(parts).to(out.device))
            return out
        if self._inplace is None:
            self._inplace = dist.get_backend(self.group) == "nccl"      # RCCL: sendbuff == recvbuff + rank * count
        if not self._inplace and inp.data_ptr() == out.data_ptr() + 8 * self.rank * inp.numel():
            inp = inp.clone()
        dist.all_gather_into_tensor(out, inp, group=self.group)
        return out

    def all_reduce_sum(self, buf):
        self.n_allreduce += 1
        self.bytes_allreduce += buf.numel() * 8
        if self.world == 1 and not self.force_collective:
            return buf
        dist = self._dist()
        if self.staged:
            h = buf.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(h.to(buf.device))
            return buf
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return buf

    def reduce_scatter_sum(self, out, inp):
        """out = sum over ranks of chunk `rank` of inp (inp = world equal contiguous chunks of out.numel() elements).
        RCCL: ncclReduceScatter; gloo has no reduce-scatter (CPU tests / host-staged ranks): all-reduce + slice there --
        the byte count recorded is what the RCCL path moves."""
        self.n_reduce_scatter += 1
        self.bytes_reduce_scatter += inp.numel() * 8
        n = out.numel()
        if self.world == 1 and not self.force_collective:
            out.copy_(inp[:n])
            return out
        dist = self._dist()
        if self.staged or dist.get_backend(self.group) != "nccl":
            h = inp.cpu() if self.staged else inp.clone()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(h[self.rank * n:(self.rank + 1) * n].to(out.device))
            return out
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def all_reduce_scalar(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        if self.world > 1:
            dist = self._dist()
            if not self.staged and dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())


class LibComm(Comm):
    """The same two collectives through libmpsk_comm (include/mpsk_comm.h): RCCL called from the C ABI on the ctx
    stream -- what a host without torch.distributed (the Julia shim) binds.  `uid`: the MPSK_COMM_ID_BYTES of rank 0's
    mpsk_comm_unique_id, shipped to every rank by the host (here: any transport, e.g. a torch.distributed broadcast)."""

    def __init__(self, be, world=1, rank=0, uid=None):
        import ctypes as C
        from . import _lib
        super().__init__(world, rank)
        self.be, self.lib = be, _lib.load_comm()
        if uid is None:
            if world != 1:
                raise ValueError("ranks > 0 need rank 0's unique id")
            uid = self.unique_id()
        self._uid = C.create_string_buffer(bytes(uid), _lib.COMM_ID_BYTES)
        h = C.c_void_p()
        _lib.check_comm(self.lib.mpsk_comm_create(be.ctx, world, rank, self._uid, C.byref(h)), "mpsk_comm_create")
        self.handle = h

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check_comm(_lib.load_comm().mpsk_comm_unique_id(buf), "mpsk_comm_unique_id")
        return buf.raw

    def all_gather_into(self, out, inp):
        from . import _lib
        self.n_allgather += 1
        self.bytes_allgather += out.numel() * 8
        _lib.check_comm(self.lib.mpsk_comm_allgather(self.handle, inp.data_ptr(), out.data_ptr(), inp.numel()),
                        "mpsk_comm_allgather")
        return out

    def all_reduce_sum(self, buf):
        from . import _lib
        self.n_allreduce += 1
        self.bytes_allreduce += buf.numel() * 8
        _lib.check_comm(self.lib.mpsk_comm_allreduce_sum(self.handle, buf.data_ptr(), buf.numel()), "mpsk_comm_allreduce_sum")
        return buf

    def reduce_scatter_sum(self, out, inp):
        from . import _lib
        self.n_reduce_scatter += 1
        self.bytes_reduce_scatter += inp.numel() * 8
        _lib.check_comm(self.lib.mpsk_comm_reduce_scatter_sum(self.handle, inp.data_ptr(), out.data_ptr(), out.numel()),
                        "mpsk_comm_reduce_scatter_sum")
        return out

    def hac_apply(self, hac, xb, out):
        """mpsk_comm_hac_apply: local rows into this rank's block of `out` + the in-place all-gather, one C call."""
        from . import _lib
        Dl, d, Dr = xb.shape
        self.n_allgather += 1
        _lib.check_comm(self.lib.mpsk_comm_hac_apply(self.handle, hac.handle, xb.ptr, out.ptr, Dl, d, Dr), "mpsk_comm_hac_apply")
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mpsk_comm_destroy(self.handle)
            self.handle = None


# ---- layout helpers (device copies through mpsk_copy2d; once per site visit, never per matvec) --------------------

def _sub(t: DTensor, offset, shape):
    """DTensor view of `t` starting `offset` elements in."""
    n = int(np.prod(shape))
    return DTensor(t.buf[offset:offset + n], shape)


def rows_of_tensor(be, A: DTensor, lo, hi):
    """rows [lo, hi) of the first index of A[a, s, b] as a contiguous (hi - lo, d, Dr) tensor."""
    Dl, d, Dr = A.shape
    n = hi - lo
    out = be.empty(n, d, Dr)
    be.copy2d(n, d * Dr, A.ptr + 8 * lo, Dl, out.ptr, n)
    return out


def rows_of_env(be, env: DTensor, lo, hi):
    """rows [lo, hi) of every slab of env (W, Db, Dk) -> (W, hi - lo, Dk) (one launch: a (Db x W Dk) matrix)."""
    W, Db, Dk = env.shape
    n = hi - lo
    out = be.empty(W, n, Dk)
    be.copy2d(n, W * Dk, env.ptr + 8 * lo, Db, out.ptr, n)
    return out


def to_blocked(be, x: DTensor, P, out: DTensor = None):
    """x[a, s, b] -> [P][n, d, Dr] row blocks back to back (same number of elements; logical shape kept)."""
    Dl, d, Dr = x.shape
    n = Dl // P
    out = be.empty(Dl, d, Dr) if out is None else out
    for q in range(P):
        be.copy2d(n, d * Dr, x.ptr + 8 * q * n, Dl, out.ptr + 8 * q * n * d * Dr, n)
    return out


def from_blocked(be, xb: DTensor, P, out: DTensor = None):
    Dl, d, Dr = xb.shape
    n = Dl // P
    out = be.empty(Dl, d, Dr) if out is None else out
    for q in range(P):
        be.copy2d(n, d * Dr, xb.ptr + 8 * q * n * d * Dr, n, out.ptr + 8 * q * n, Dl)
    return out


class ShardedSiteOp:
    """H_AC of one site with the bond index sharded: `y = h(x, out)` on BLOCKED vectors; `encode` / `decode` convert a
    tensor to / from the blocked layout.  One in-place all-gather per application, nothing allocated."""

    def __init__(self, be, comm: Comm, H, GLrows: DTensor, GRfull: DTensor):
        self.be, self.comm, self.o, self.leftenv, self.rightenv = be, comm, H, GLrows, GRfull
        self.P = comm.world
        self.n_apply = 0
        self._hac = None

    def agree(self, flag: bool) -> bool:
        """True iff EVERY rank says True (krylov.eigsolve_sr asks before acting on a convergence decision)."""
        if self.comm.world == 1 and not self.comm.force_collective:
            return bool(flag)
        self.comm.n_agree = getattr(self.comm, "n_agree", 0) + 1
        return self.comm.all_reduce_scalar(1.0 if flag else 0.0) >= self.comm.world - 0.5

    def encode(self, x: DTensor, out: DTensor = None):
        return to_blocked(self.be, x, self.P, out)

    def decode(self, xb: DTensor, out: DTensor = None):
        return from_blocked(self.be, xb, self.P, out)

    def __call__(self, xb: DTensor, out: DTensor = None):
        be, P, r = self.be, self.P, self.comm.rank
        Dl, d, Dr = xb.shape
        out = be.empty(Dl, d, Dr) if out is None else out
        blk = (Dl // P) * d * Dr
        mine = _sub(out, r * blk, (Dl // P, d, Dr))
        if hasattr(be, "hac_create"):
            if self._hac is None:                    # prepared once per site visit (mpsk_hac_create)
                self._hac = be.hac_create(self.o, self.leftenv, self.rightenv)
            self._hac.apply(xb, out=mine, nblk=P)
        else:                                        # host stand-in backend of the CPU tests
            be.dAC_blocked(self.o, self.leftenv, self.rightenv, xb, P, out=mine)
        self.comm.all_gather_into(out.buf[:P * blk], mine.buf[:blk])
        self.n_apply += 1
        return out

    __mul__ = __call__


class ShardedFinEnv:
    """FinEnv (src/environments/FinEnv.jl:9-145) with STORAGE sharded over the process group: same lazy,
    identity-based invalidation as environments.FinEnv (an AL / AR that was recomputed is a new object), but a left
    environment of a shardable bond is kept as this rank's row block and a right environment as this rank's bra-column
    block (module docstring).  `site_op(pos, psi)` returns the matvec the sweep uses; `leftenv` / `rightenv` return
    FULL tensors (gathered on demand: diagnostics, expectation values -- not on the sweep's hot path)."""

    def __init__(self, psi, H, comm: Comm, min_block=64, force=False):
        from .environments import _start_env
        be = self.be = psi.be
        self.comm, self.min_block, self.force = comm, min_block, force
        self.P, self.rank = comm.world, comm.rank
        L = len(psi)
        self.H = H
        self.opp = [H[i] for i in range(L)]
        odim = H.odim
        D0 = psi.AL(0).shape[0]
        t = psi.ARs[L - 1] if psi.ARs[L - 1] is not None else psi.AL(L - 1)
        DL = t.shape[2]
        if self._ok(D0) or self._ok(DL):
            raise ValueError("the chain-edge bonds must not be shardable (open boundary: dimension 1)")
        self.leftenvs = [_start_env(be, self.opp[0].chil, D0, 0)] + [None] * L
        self.rightenvs = [None] * L + [_start_env(be, self.opp[L - 1].chir, DL, odim - 1)]
        self.lkind = ["rep"] + [None] * L         # 'rep' | 'row'
        self.rkind = [None] * L + ["rep"]         # 'rep' | 'col'
        self.ldeps = [None] * L
        self.rdeps = [None] * L
        self.n_transfers = 0
        self._full_right = {}                     # bond index -> (stored tensor it was gathered from, full tensor)

    def _ok(self, D):
        return BondShard.shardable(D, self.P, self.min_block, self.force)

    def _lohi(self, D):
        n = D // self.P
        return self.rank * n, (self.rank + 1) * n

    # ---- environment updates ---------------------------------------------------------------------------------------
    def _transfer_left(self, j, al):
        be, Hs = self.be, self.opp[j]
        GLin, kin = self.leftenvs[j], self.lkind[j]
        Dl, d, Dr = al.shape
        kout = "row" if self._ok(Dr) else "rep"
        if kin == "rep":
            if kout == "rep":
                return be.transfer_left(Hs, GLin, al, al), kout
            # replicated input, shardable output bond: rank p computes ITS rows b'_p from the bra columns b'_p
            # (a contiguous column block of AL as a (Dl d x Dr) matrix) -- no communication
            lo, hi = self._lohi(Dr)
            ab = _sub(al, lo * Dl * d, (Dl, d, hi - lo))
            return be.transfer_left(Hs, GLin, al, ab), kout
        lo, hi = self._lohi(Dl)
        abloc = rows_of_tensor(be, al, lo, hi)
        part = be.transfer_left(Hs, GLin, al, abloc)               # contraction over the local rows a'_p only
        if kout == "rep":
            self.comm.all_reduce_sum(part.buf[:part.size])
            return part, kout
        # row-sharded output: rank q needs rows q of every slab -> rank-major row blocks [P][W][n, Dr] (rows q of all slabs
        # = one (n x W Dr) sub-matrix of the (Drb x W Dr) matrix `part`), then ONE reduce-scatter: half the bytes of an
        # all-reduce, and nothing is received that is thrown away
        W, Drb, Drk = part.shape
        P, n2 = self.P, Drb // self.P
        blocks = be.empty(P, W, n2, Drk)
        for q in range(P):
            be.copy2d(n2, W * Drk, part.ptr + 8 * q * n2, Drb, blocks.ptr + 8 * q * W * n2 * Drk, n2)
        out = be.empty(W, n2, Drk)
        self.comm.reduce_scatter_sum(out.buf[:out.size], blocks.buf[:blocks.size])
        return out, kout

    def _gathered_right(self, idx):
        """full (W, Dk, Db) copy of right environment `idx` (the bond index), gathered if stored column-sharded."""
        t, kind = self.rightenvs[idx], self.rkind[idx]
        if kind == "rep":
            return t
        hit = self._full_right.get(idx)
        if hit is not None and hit[0] is t:
            return hit[1]
        W, Dk, n = t.shape
        be, P = self.be, self.P
        full = be.empty(W, Dk, n * P)
        slab = Dk * n
        # ONE collective (rank-major [P][W][Dk, n]), then P strided device copies: rank q's column block of every slab is
        # one chunk of Dk n contiguous elements per slab (round 2 issued W collectives per gather)
        staged = be.empty(P, W, Dk, n)
        self.comm.all_gather_into(staged.buf[:staged.size], t.buf[:t.size])
        for q in range(P):
            be.copy2d(slab, W, staged.ptr + 8 * q * W * slab, slab, full.ptr + 8 * q * slab, slab * P)
        if len(self._full_right) >= 2:
            for k in sorted(self._full_right, key=lambda k: abs(k - idx), reverse=True)[:len(self._full_right) - 1]:
                del self._full_right[k]
        self._full_right[idx] = (t, full)
        return full

    def _transfer_right(self, j, ar):
        be, Hs = self.be, self.opp[j]
        GRfull = self._gathered_right(j + 1)
        Dl, d, Dr = ar.shape
        if not self._ok(Dl):
            return be.transfer_right(Hs, GRfull, ar, ar), "rep"
        lo, hi = self._lohi(Dl)
        abloc = rows_of_tensor(be, ar, lo, hi)
        return be.transfer_right(Hs, GRfull, ar, abloc), "col"     # (W, Dl, n): the bra columns a'_p, no reduction

    def _update_right(self, ind, psi):  # FinEnv.jl:114-129
        L = len(psi)
        a = None
        for i in range(L - 1, ind, -1):
            if psi.AR(i) is not self.rdeps[i]:
                a = i
                break
        if a is not None:
            for j in range(a, ind, -1):
                ar = psi.AR(j)
                self.rightenvs[j], self.rkind[j] = self._transfer_right(j, ar)
                self._full_right.pop(j, None)
                self.rdeps[j] = ar
                self.n_transfers += 1

    def _update_left(self, ind, psi):  # FinEnv.jl:131-145
        a = None
        for i in range(0, ind):
            if psi.AL(i) is not self.ldeps[i]:
                a = i
                break
        if a is not None:
            for j in range(a, ind):
                al = psi.AL(j)
                self.leftenvs[j + 1], self.lkind[j + 1] = self._transfer_left(j, al)
                self.ldeps[j] = al
                self.n_transfers += 1

    # ---- the sweep's accessors ---------------------------------------------------------------------------------------
    def site_op(self, pos, psi):
        """H_AC of site pos: ShardedSiteOp when the left bond is sharded, the plain MPO_ddAC otherwise."""
        self._update_left(pos, psi)
        self._update_right(pos, psi)
        GR = self._gathered_right(pos + 1)
        if self.lkind[pos] == "row":
            return ShardedSiteOp(self.be, self.comm, self.opp[pos], self.leftenvs[pos], GR)
        return MPO_ddAC(self.be, self.opp[pos], self.leftenvs[pos], GR)

    # ---- FinEnv-compatible accessors (full tensors; diagnostics) ---------------------------------------------------
    def rightenv(self, ind, psi):
        self._update_right(ind, psi)
        return self._gathered_right(ind + 1)

    def leftenv(self, ind, psi):
        self._update_left(ind, psi)
        t = self.leftenvs[ind]
        if self.lkind[ind] == "rep":
            return t
        be, P = self.be, self.P
        W, n, Dk = t.shape
        gathered = be.empty(P, W, n, Dk)
        self.comm.all_gather_into(gathered.buf[:gathered.size], t.buf[:t.size])
        full = be.empty(W, n * P, Dk)
        for q in range(P):      # rank q's rows of every slab: an (n x W Dk) matrix -> rows [q n, (q+1) n) of (D x W Dk)
            be.copy2d(n, W * Dk, gathered.ptr + 8 * q * t.size, n, full.ptr + 8 * q * n, n * P)
        return full

    def poison(self, ind):  # FinEnv.jl:108-111
        self.ldeps[ind] = None
        self.rdeps[ind] = None

    def bytes_local(self):
        """bytes of environment storage held by this rank (persistent tensors; the <= 2 transient gathered right
        environments are reported separately)."""
        tot = sum(t.size for t in self.leftenvs if t is not None) + sum(t.size for t in self.rightenvs if t is not None)
        return 8 * tot, 8 * sum(f.size for _, f in self._full_right.values())


def memory_model(L, D, d, W, P, krylov_vectors=10):
    """Per-GPU bytes of a sharded 1-site DMRG sweep at bulk bond dimension D: environments 1/P,
    state / Krylov vectors replicated, matvec workspace 1/P."""
    env = 2 * L * W * D * D * 8 / P
    transient = 2 * W * D * D * 8
    state = L * D * d * D * 8 * 1.05 + 3 * D * d * D * 8           # one gauge copy per site (+ the centre's AC / AL / AR)
    krylov = (krylov_vectors + 2) * D * d * D * 8
    work = 2 * W * (D // P) * d * D * 8
    return {"environments": env, "gathered_right_envs": transient, "state": state, "krylov": krylov,
            "matvec_workspace": work, "total": env + transient + state + krylov + work}
