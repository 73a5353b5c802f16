#!/usr/bin/env python
"""bench.py -- DMRG sweep throughput + dAC matvec TFLOP/s on MI355X (contract: see task brief).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full 1-site DMRG sweep (2L-2 site updates, dmrg.jl:33-38 of the reference) of
the Heisenberg S=1/2 chain, L=100, D=1024, fp64, with a fixed Krylov budget of 8 matvecs per
site, on a seeded synthetic uniform[0,1) random MPS that is already
left-canonical and resident in HBM when the timed region starts.  N > 1: the bond index of the
effective-Hamiltonian matvec is block-sharded over the ranks with one RCCL all-gather per matvec
(strong scaling: the same chain on N GPUs).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (256 CU x 2.4 GHz x 128 flop/clk/CU)


def flops_dAC(D, d, W):
    """algorithmic real flops of one dAC matvec."""
    return 2 * W * D * D * d * D + 2 * W * W * d * d * D * D + 2 * W * D * d * D * D


def cpu_baseline(L, D, d, budget_s=20.0):
    """Oracle (NumPy/OpenBLAS restatement of the reference's per-block evaluation order) timed on
    the host cores on a bounded sample: bulk-site updates (8 Krylov matvecs + galerkin matvec +
    environment update + QRpos of the old and new AC), extrapolated to sweeps/s with the flop-model
    weight of every site of the chain."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import mpskit_oracle as mo
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    rng = np.random.default_rng(20240213)
    H = mo.heisenberg_mpo(0.5)
    slc = H[0]
    W = slc.odim
    GL = [rng.random((D, 1, D)) for _ in range(W)]
    GR = [rng.random((D, 1, D)) for _ in range(W)]
    x = rng.random((D, d, D))
    AL, _ = mo.leftorth(rng.random((D, d, D)))

    def site_update():
        v = x
        for _ in range(8 + 1):                 # 8 Krylov matvecs + 1 galerkin matvec
            v = mo.dAC(v, slc, GL, GR)
            v = v / np.linalg.norm(v)
        mo.leftorth(v)                         # QRpos of the old AC (galerkin projector)
        al, _ = mo.leftorth(v)                 # QRpos of the new AC
        mo.transfer_left(GL, slc, al, al)      # environment update
    t0 = time.time()
    site_update()
    t_first = time.time() - t0
    n = 1
    while time.time() - t0 + t_first < budget_s and n < 8:
        site_update()
        n += 1
    t_site = (time.time() - t0) / n
    # flop-model weight of the whole sweep in units of a bulk site
    dl = [1]                                   # bond dims min(d^i, D, d^(L-i))  (finitemps.jl:182-192)
    for _ in range(1, L):
        dl.append(min(dl[-1] * d, D))
    dl.append(1)
    for k in range(L - 1, 0, -1):
        dl[k] = min(dl[k], dl[k + 1] * d)
    bulk = 2.0 * D * D * D
    order = list(range(0, L - 1)) + list(range(L - 1, 0, -1))
    equiv = sum((dl[p] * dl[p] * dl[p + 1] + dl[p] * dl[p + 1] * dl[p + 1]) / bulk for p in order)
    sweeps_per_s = 1.0 / (t_site * equiv)
    return {"value": sweeps_per_s, "unit": "sweeps/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} bulk-site updates at D={D} (8+1 dAC matvecs, 2 QRpos, 1 transfer_left; "
                      f"{t_site:.2f} s each) of the oracle, extrapolated over the {len(order)} site updates of a "
                      f"sweep by the D^3 flop model ({equiv:.1f} bulk-site equivalents)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--D", type=int, default=1024)
    ap.add_argument("--matvecs", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded sweep (blocked vectors, storage-sharded environments, RCCL collectives) even with "
                         "one rank: exercises the N > 1 code path on a 1-GPU box")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_shard:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import mpskit_jl_amd as mk
    from mpskit_jl_amd import algorithms as alg, krylov
    from mpskit_jl_amd.dist import Comm, ShardedFinEnv

    be = mk.Backend(local_rank)
    L, D, d = args.L, args.D, 2
    H = mk.heisenberg_XXX(0.5, be=be)
    W = H[0].Wl
    rng = np.random.default_rng(20240213)          # same seed on every rank -> identical replicas
    psi = mk.FiniteMPS.random(L, d, D, rng, normalize=True, be=be)
    sharded = world > 1 or args.force_shard
    comm = Comm(world, rank, force_collective=args.force_shard) if sharded else None
    # N > 1: bond-sharded sweep -- storage-sharded environments (1/N per GPU), blocked Krylov vectors, one in-place
    # all-gather per matvec, one all-reduce per left-environment update, one gather per right-environment use
    envs = ShardedFinEnv(psi, H, comm, force=args.force_shard) if sharded else mk.FinEnv(psi, H)
    eig = mk.Arnoldi(fixed_matvecs=args.matvecs, krylovdim=max(args.matvecs, 2))
    ws = krylov.KrylovWorkspace(be)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        alg.dmrg_sweep(psi, H, envs, eig, ws)
    be.prof_enable(True)
    qr0 = be.qr_stats()
    barrier()
    t0 = time.perf_counter()
    eps = None
    for _ in range(args.steps):
        eps = alg.dmrg_sweep(psi, H, envs, eig, ws)
    barrier()
    dt = time.perf_counter() - t0
    be.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = be.prof_summary()

    # dominant kernel = the matvec-stage GEMM instance with the largest total time
    roofline = None
    if prof:
        top = max(prof, key=lambda r: r["total_ms"])
        achieved = top["flops"] / (top["total_ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": top["kernel"], "achieved": round(achieved, 3),
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                    "launches": top["launches"], "avg_ms": round(top["avg_ms"], 5),
                    "flops_per_launch": top["flops"] / top["launches"], "traffic": None}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        # HBM bytes per launch from a separate rocprofv3 --pmc run of the north-star point (tools/dac_only.py: D = 1024,
        # one GPU); for any other workload the counters were not collected -> null
        if os.path.exists(pmc) and args.D == 1024 and world == 1:
            try:
                roofline["traffic"] = json.load(open(pmc)).get(top["kernel"])
            except Exception:
                pass

    # whole-matvec rate at the north-star point (D, d=2, W=5) on the same stream, HIP events
    dac_tflops = None
    if rank == 0:
        GL = mk.DTensor(torch.rand(W * D * D, dtype=torch.float64, device=be.device), (W, D, D))
        GR = mk.DTensor(torch.rand(W * D * D, dtype=torch.float64, device=be.device), (W, D, D))
        x = mk.DTensor(torch.rand(D * d * D, dtype=torch.float64, device=be.device), (D, d, D))
        y = be.empty(D, d, D)
        hop = mk.MPO_ddAC(be, H[0], GL, GR)         # what the sweep applies: prepared once per site, then applied
        for _ in range(3):
            hop(x, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 20
        e0.record()
        for _ in range(nrep):
            hop(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nrep
        dac_tflops = flops_dAC(D, d, W) / (ms * 1e-3) / 1e12
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "DMRG sweeps/sec + ddAC matvec achieved-TFLOP/s, Heisenberg L=100 D=1024 fp64",
            "value": args.steps / dt, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Heisenberg S=1/2 FiniteMPS L={L} D={D} d=2 W={W} fp64, 1-site DMRG sweep "
                                   f"(2L-2 site updates), fixed Krylov budget {args.matvecs} matvecs/site",
                       "parallelism": "single GPU" if not sharded else
                       f"bond index sharded x{world}: storage-sharded environments (1/{world} per GPU), one in-place RCCL "
                       "all-gather per matvec, one all-reduce per left-environment update, gauge steps replicated"},
            "dAC_tflops": None if dac_tflops is None else round(dac_tflops, 3),
            "dAC_frac_of_fp64_mfma_peak": None if dac_tflops is None else round(dac_tflops / FP64_MFMA_PEAK_TFLOPS, 4),
            "max_galerkin_last_sweep": None if eps is None else float(max(eps)),
            "qr_calls_timed": {k: be.qr_stats()[k] - qr0[k] for k in qr0},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, D, d)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_shard:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
