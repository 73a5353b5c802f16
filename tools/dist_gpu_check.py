"""Rank process of tests/test_gpu_dist.py: world_size ranks share GPU 0, collectives staged through the host (gloo).
Checks the device-side sharding plumbing (row blocks, local mpsk_dAC with Dlo = D / P, re-interleave) and a sharded
DMRG sweep against the unsharded result.  usage: dist_gpu_check.py RANK WORLD PORT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np, torch, torch.distributed as dist
import mpskit_jl_amd as mk
from mpskit_jl_amd import dist as md, algorithms as alg, krylov

dist.init_process_group("gloo", rank=rank, world_size=world)
be = mk.Backend(0)
pl = md.HostStagedPlumbing(be)
rng = np.random.default_rng(5)                       # same inputs on every rank
D, d, W = 192, 2, 5
H = mk.heisenberg_XXX(0.5, be=be)
GL, GR = be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)]), be.upload_env([rng.standard_normal((D, 1, D)) for _ in range(W)])
x = be.upload(rng.standard_normal((D, d, D)))
ref = be.download(be.dAC(H[1], GL, GR, x))
y = be.download(md.ShardedMatvec(pl, H[1], GL, GR, world, rank)(x))
err = np.abs(y - ref).max() / np.abs(ref).max()
# sharded sweep == unsharded sweep (bit-identical collectives: every rank must hold the same state)
L, Dm = 12, 128
psi = mk.FiniteMPS.random(L, 2, Dm, np.random.default_rng(1), be=be)
ps = psi.copy()
eig = mk.Arnoldi(fixed_matvecs=4, krylovdim=4)
alg.dmrg_sweep(psi, H, mk.FinEnv(psi, H), eig, krylov.KrylovWorkspace(be))
alg.dmrg_sweep(ps, H, mk.FinEnv(ps, H), eig, krylov.KrylovWorkspace(be), md.shard_wrapper(be, world, rank, None, 32, plumbing=pl))
# sharded environment updates (all-reduce / all-gather + device re-interleave)
st = md.ShardedTransfer(pl, world, rank, None, 32)
A = be.upload(rng.standard_normal((D, d, D)))
tl_s, tl_u = be.download(st.transfer_left(H[1], GL, A, A)), be.download(be.transfer_left(H[1], GL, A, A))
tr_s, tr_u = be.download(st.transfer_right(H[1], GR, A, A)), be.download(be.transfer_right(H[1], GR, A, A))
err_t = max(np.abs(tl_s - tl_u).max() / np.abs(tl_u).max(), np.abs(tr_s - tr_u).max() / np.abs(tr_u).max())
pt = mk.FiniteMPS.random(L, 2, Dm, np.random.default_rng(1), be=be)
alg.dmrg_sweep(pt, H, mk.FinEnv(pt, H, transfer_ops=st), eig, krylov.KrylovWorkspace(be), md.shard_wrapper(be, world, rank, None, 32, plumbing=pl))
e3 = float(np.sum(mk.expectation_value(pt, H, mk.FinEnv(pt, H))))
e1 = float(np.sum(mk.expectation_value(psi, H, mk.FinEnv(psi, H))))
e2 = float(np.sum(mk.expectation_value(ps, H, mk.FinEnv(ps, H))))
t = torch.tensor([e2], dtype=torch.float64)
lst = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(lst, t)
spread = max(abs(float(a) - e2) for a in lst)
ok = err < 1e-13 and err_t < 1e-13 and abs(e1 - e2) < 1e-10 * abs(e1) and abs(e1 - e3) < 1e-10 * abs(e1) and spread == 0.0 and st.n_collectives > 2
print(f"rank {rank}: matvec relerr {err:.2e}, transfer relerr {err_t:.2e}, sharded-env sweep {e3:.12f}, sweep energy {e1:.12f} vs sharded {e2:.12f}, spread {spread:.1e} -> {'OK' if ok else 'FAIL'}", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
