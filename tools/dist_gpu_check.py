"""Rank process of tests/test_gpu_dist.py: world_size ranks share GPU 0, collectives staged through the host (gloo).
Checks the device side of the bond-sharded sweep (mpsk_dAC_blocked on the local rows, in-place gather into the blocked
vector, storage-sharded environments and their updates) against the unsharded result.
usage: dist_gpu_check.py RANK WORLD PORT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np, torch, torch.distributed as dist
import mpskit_jl_amd as mk
from mpskit_jl_amd import dist as md, algorithms as alg, krylov

dist.init_process_group("gloo", rank=rank, world_size=world)
be = mk.Backend(0)
comm = md.Comm(world, rank, staged=True)
rng = np.random.default_rng(5)                       # same inputs on every rank
D, d, W = 192, 2, 5
H = mk.heisenberg_XXX(0.5, be=be)
GL, GR = be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)]), be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)])
x = be.upload(rng.standard_normal((D, d, D)))
ref = be.download(be.dAC(H[1], GL, GR, x))
n = D // world
op = md.ShardedSiteOp(be, comm, H[1], md.rows_of_env(be, GL, rank * n, (rank + 1) * n), GR)
y = be.download(op.decode(op(op.encode(x))))
err = np.abs(y - ref).max() / np.abs(ref).max()
# sharded sweep (storage-sharded environments) == unsharded sweep; every rank must hold the same state
L, Dm = 18, 64
psi = mk.FiniteMPS.random(L, 2, Dm, np.random.default_rng(1), be=be)
ps = psi.copy()
eig = mk.Arnoldi(fixed_matvecs=4, krylovdim=4)
eu, es = mk.FinEnv(psi, H), md.ShardedFinEnv(ps, H, comm, min_block=32)
# environments of the SAME state: gathered shards == the unsharded tensors
err_t = 0.0
for pos in (L - 1, 0, 8, 11, 5):
    for a, b in ((es.leftenv(pos, ps), eu.leftenv(pos, psi)), (es.rightenv(pos, ps), eu.rightenv(pos, psi))):
        a, b = be.download(a), be.download(b)
        err_t = max(err_t, np.abs(a - b).max() / np.abs(b).max())
for _ in range(2):
    alg.dmrg_sweep(psi, H, eu, eig, krylov.KrylovWorkspace(be))
    alg.dmrg_sweep(ps, H, es, eig, krylov.KrylovWorkspace(be))
stored, transient = es.bytes_local()
full = 8 * (sum(t.size for t in eu.leftenvs if t is not None) + sum(t.size for t in eu.rightenvs if t is not None))
e1 = float(np.sum(mk.expectation_value(psi, H, eu)))
e2 = float(np.sum(mk.expectation_value(ps, H, es)))
t = torch.tensor([e2], dtype=torch.float64)
lst = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(lst, t)
spread = max(abs(float(a) - e2) for a in lst)
ok = (err < 1e-13 and err_t < 1e-12 and abs(e1 - e2) < 1e-10 * abs(e1) and spread == 0.0 and comm.n_allreduce > 2
      and comm.n_allgather > 20 and stored < 0.6 * full and es.n_transfers == eu.n_transfers)
print(f"rank {rank}: matvec relerr {err:.2e}, env relerr {err_t:.2e}, sweep energy {e1:.12f} vs sharded {e2:.12f}, spread {spread:.1e}, "
      f"env bytes {stored / full:.2f} of replicated, collectives {comm.n_allgather} ag / {comm.n_allreduce} ar -> {'OK' if ok else 'FAIL'}", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
