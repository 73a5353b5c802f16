#!/usr/bin/env python
"""bench.py -- DMRG sweep throughput + dAC matvec TFLOP/s on MI355X (contract: see task brief).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full 1-site DMRG sweep (2L-2 site updates, dmrg.jl:33-38 of the reference) of
the Heisenberg S=1/2 chain, L=100, D=1024, fp64, with a fixed Krylov budget of 8 matvecs per
site, on a seeded synthetic uniform[0,1) random MPS that is already
left-canonical and resident in HBM when the timed region starts.  N > 1: the bond index is
block-sharded over the ranks -- storage-sharded environments, one in-place RCCL all-gather per
matvec, one all-reduce per left-environment update (strong scaling: the same chain on N GPUs).
After the timed steps two more legs run on FRESH copies of the initial state psi0 (so that they
do not depend on --steps): `early_sweeps` = sweeps 1-2 from psi0 with the same fixed budget (the
gauge steps of unconverged tensors are worse conditioned: CholeskyQR retries are counted), and
`to_tolerance` = sweeps 1..T from psi0 with the reference's default eigensolver
Arnoldi(tol=1e-12, krylovdim=30, maxiter=100) (defaults.jl:33).  At N = 1 the oracle's CPU
restatement is timed on the host cores (`cpu_baseline`).  Rank 0 prints ONE JSON line.
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


def kernel_source_hash():
    """sha256 over the kernel sources a PMC traffic figure belongs to: a committed profiles/*pmc_traffic.json is only
    attached to the bench line when it was collected on exactly these sources."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mpsk_gemm.hip", "mpsk_api.hip", "mpsk_ops.hip", "mpsk_internal.h"):
        with open(os.path.join(ROOT, "mpskit.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(L, D, d, matvecs, budget_s=24.0, min_samples=3):
    """The reference's CPU path timed on this box's host cores ("port": the oracle's restatement; the reference itself is
    Julia and cannot run here).  One sample = one bulk-site update exactly as the reference performs it: `matvecs` Krylov
    applications of H_AC + 1 more for calc_galerkin (toolbox.jl:18), each evaluated block by block (one pair of
    contractions per non-zero MPO block, derivatives.jl:85-104), QRpos of the old and of the new AC (LAPACK geqrf +
    orgqr through numpy), one transfer_left.  Every contraction is a plain BLAS dgemm on operands that are already in
    GEMM layout (no permutes inside the timed region), OpenBLAS pinned to the physical cores -- so the number reflects
    a BLAS-bound reference run, not NumPy overheads.  Extrapolated over the sweep with the D^3 flop model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import mpskit_oracle as mo
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        cores = os.cpu_count() or 1
    try:                                           # cgroup limit (the GPU box gives a CPU share)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(20240213)
    slc = mo.heisenberg_mpo(0.5)[0]
    W = slc.odim
    keys = list(slc.keys())
    # operands in GEMM layout: GL[i] as (D x D), x as (D x d D), GR[j] as (D x D); the d x d block acts on the middle index
    GLm = [np.asfortranarray(rng.random((D, D))) for _ in range(W)]
    GRm = [np.asfortranarray(rng.random((D, D))) for _ in range(W)]
    Om = {k: (slc.Os[k] if np.isscalar(slc.Os[k]) else np.asarray(slc.Os[k]).reshape(d, d)) for k in keys}
    x0 = np.asfortranarray(rng.random((D, d * D)))
    flop_mv = len(keys) * (2.0 * D * D * d * D + 2.0 * D * d * D * D)          # what this evaluation order executes

    def matvec(x):                                  # x: (D, d D) column-major  ==  x[a, (s, b)]
        y = np.zeros((D * d, D), order="F")
        for (i, j) in keys:
            t1 = GLm[i] @ x                         # dgemm  D x D x dD      -> [p, (s, b)]
            t3 = t1.reshape(D, d, D, order="F")
            O = Om[(i, j)]
            if np.isscalar(O):
                t2 = O * t3
            else:                                   # d x d physical block (O[t, s]): tiny, applied plane by plane
                t2 = np.empty_like(t3)
                for t in range(d):
                    t2[:, t, :] = sum(O[t, s_] * t3[:, s_, :] for s_ in range(d))
            y += t2.reshape(D * d, D, order="F") @ GRm[j]                       # dgemm  dD x D x D
        return y.reshape(D, d * D, order="F")

    def site_update():
        v = x0
        for _ in range(matvecs + 1):               # Krylov matvecs + the galerkin matvec
            v = matvec(v)
            v = v / np.linalg.norm(v)
        a = v.reshape(D * d, D, order="F")
        np.linalg.qr(a)                             # QRpos of the old AC (galerkin projector)
        q, _ = np.linalg.qr(a)                      # QRpos of the new AC
        al = q.reshape(D, d, D, order="F")
        mo.transfer_left([g[:, None, :] for g in GLm], slc, al, al)             # environment update

    with threadpool_limits(limits=int(cores)):
        site_update()                               # warm-up (page faults, thread pool)
        times = []
        t_start = time.time()
        while len(times) < min_samples or (time.time() - t_start < budget_s and len(times) < 12):
            t0 = time.time()
            site_update()
            times.append(time.time() - t0)
        t0 = time.time()
        matvec(x0)
        t_mv = time.time() - t0
    t_site = float(np.median(times))
    dl = [1]                                        # bond dims min(d^i, D, d^(L-i))  (finitemps.jl:182-192)
    for _ in range(1, L):
        dl.append(min(dl[-1] * d, D))
    dl.append(1)
    for k in range(L - 1, 0, -1):
        dl[k] = min(dl[k], dl[k + 1] * d)
    bulk = 2.0 * D * D * D
    order = list(range(0, L - 1)) + list(range(L - 1, 0, -1))
    equiv = sum((dl[p] * dl[p] * dl[p + 1] + dl[p] * dl[p + 1] * dl[p + 1]) / bulk for p in order)
    return {"value": 1.0 / (t_site * equiv), "unit": "sweeps/s", "cores": int(cores), "kind": "port",
            "operator_applications_per_site": matvecs + 1,
            "matvec_gflops": round(flop_mv / t_mv / 1e9, 1),
            "site_seconds": {"median": round(t_site, 3), "min": round(min(times), 3), "max": round(max(times), 3), "n": len(times)},
            "sample": f"{len(times)} bulk-site updates at D={D} ({matvecs}+1 per-block dAC matvecs as plain dgemm calls on "
                      f"GEMM-layout operands, 2 LAPACK QR, 1 transfer_left; OpenBLAS pinned to {int(cores)} threads) of the oracle's "
                      f"restatement of the reference's evaluation order, median extrapolated over the {len(order)} site updates "
                      f"of a sweep by the D^3 flop model ({equiv:.1f} bulk-site equivalents)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--D", type=int, default=1024)
    ap.add_argument("--matvecs", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tolerance-sweep", action="store_true", help="skip the converged-tolerance leg")
    ap.add_argument("--tol-sweeps", type=int, default=2, help="sweeps of the converged-tolerance leg (from psi0)")
    ap.add_argument("--early-sweeps", type=int, default=2, help="fixed-budget sweeps from psi0 reported as `early_sweeps`")
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
        # RCCL prints a version banner on STDOUT when its communicator is created; the contract is ONE JSON line there,
        # so stdout is pointed at stderr (at the descriptor level: the banner comes from C) until the communicator exists
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            t = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
            dist.all_reduce(t)                      # forces the lazy communicator creation now
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    import mpskit_jl_amd as mk
    from mpskit_jl_amd import algorithms as alg, krylov
    from mpskit_jl_amd.dist import Comm, ShardedFinEnv

    be = mk.Backend(local_rank)
    L, D, d = args.L, args.D, 2
    H = mk.heisenberg_XXX(0.5, be=be)
    W = H[0].Wl
    rng = np.random.default_rng(20240213)          # same seed on every rank -> identical replicas
    psi = mk.FiniteMPS.random(L, d, D, rng, normalize=True, be=be)
    psi0 = psi.copy()                              # (stored tensors are never modified in place: a shallow copy IS psi0)
    sharded = world > 1 or args.force_shard
    comm = Comm(world, rank, force_collective=args.force_shard) if sharded else None
    # N > 1: bond-sharded sweep -- storage-sharded environments (1/N per GPU), blocked Krylov vectors, one in-place
    # all-gather per matvec, one all-reduce per left-environment update, one gather per right-environment use
    def make_envs(p):
        return ShardedFinEnv(p, H, comm, force=args.force_shard) if sharded else mk.FinEnv(p, H)

    envs = make_envs(psi)
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
    qr_retry0 = be.qr_retries()
    barrier()
    t0 = time.perf_counter()
    eps = None
    for _ in range(args.steps):
        eps = alg.dmrg_sweep(psi, H, envs, eig, ws)
    barrier()
    dt = time.perf_counter() - t0
    be.prof_enable(False)
    qr_timed = {k: be.qr_stats()[k] - qr0[k] for k in qr0}
    qr_timed["shift_retries"] = be.qr_retries() - qr_retry0
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
        # HBM-side bytes per launch of that kernel from the separate rocprofv3 --pmc passes of tools/pmc_quick.sh
        # (FETCH_SIZE / WRITE_SIZE cannot share a pass with anything else; D = 1024 north-star point).  Attached only
        # if the passes were taken on EXACTLY the kernel sources of this run (hash), else null.
        for tag in ("r03", "r02"):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
            if roofline["traffic"] is None and os.path.exists(pmc) and args.D == 1024 and world == 1:
                try:
                    pj = json.load(open(pmc))
                    if pj.get("kernel_source_sha256_16") == kernel_source_hash() and top["kernel"] in pj:
                        roofline["traffic"] = pj[top["kernel"]]
                        roofline["traffic_source"] = (f"profiles/{tag}_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                      "passes, tools/dac_only.py, same kernel sources)")
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
        # ~70 ms of warm-up: the event read-back above leaves the GPU idle long enough for its clocks to drop, and the
        # first ~50 ms of matvecs after an idle period run 15 % slower (0.764 ms per matvec in the first batch of 20 after
        # an idle period, 0.662 ms in every later one)
        for _ in range(100):
            hop(x, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 50
        e0.record()
        for _ in range(nrep):
            hop(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nrep
        dac_tflops = flops_dAC(D, d, W) / (ms * 1e-3) / 1e12
    if world > 1:
        dist.barrier()

    def qr_snapshot():
        q = dict(be.qr_stats())
        q["shift_retries"] = be.qr_retries()
        return q

    def energy(p, e):
        try:
            return float(np.sum(alg.expectation_value(p, H, e)))
        except Exception:                           # (not every environment flavour serves expectation_value)
            return None

    # early sweeps: the SAME fixed-budget sweep, but on sweeps 1..k from psi0 (the timed steps above run on the state
    # W + K sweeps have already converged; unconverged tensors make the gauge steps' Gram matrices worse conditioned)
    early = None
    if args.early_sweeps > 0:
        p1 = psi0.copy()
        e1 = make_envs(p1)
        rows = []
        for i in range(args.early_sweeps):
            q0 = qr_snapshot()
            barrier()
            t0 = time.perf_counter()
            eps_e = alg.dmrg_sweep(p1, H, e1, eig, ws)
            barrier()
            dt_e = time.perf_counter() - t0
            q1 = qr_snapshot()
            rows.append({"sweep": i + 1, "sweeps_per_s": round(1.0 / dt_e, 4), "ms": round(dt_e * 1e3, 1),
                         "energy": energy(p1, e1), "max_galerkin": float(max(eps_e)),
                         "qr_calls": {k: q1[k] - q0[k] for k in q1}})
        early = {"from": "psi0 (fresh copy)", "fixed_matvecs": args.matvecs, "sweeps": rows}
        del p1, e1

    # converged-tolerance mode (SURVEY 8d / BASELINE.md section 3): sweeps 1..T from a fresh copy of psi0 with the
    # reference's default eigensolver Arnoldi(tol = 1e-12, krylovdim = 30, maxiter = 100, eager) (defaults.jl:33,
    # dmrg.jl:17) instead of the fixed budget -- independent of --steps
    to_tol = None
    # (N = 1 only: a tolerance-mode solve takes data-dependent host decisions per Krylov step; the ranks of a sharded sweep
    #  hold bit-identical scalars by construction, but that lock-step has never met N > 1 on hardware -- the scaling runs
    #  time the fixed-budget sweeps, which have no data-dependent control flow)
    if not args.no_tolerance_sweep and args.tol_sweeps > 0 and world == 1:
        nmv = {"n": 0}
        orig_eig = krylov.eigsolve_sr

        def counting(*a, **kw):
            r = orig_eig(*a, **kw)
            nmv["n"] += r[2]
            return r

        krylov.eigsolve_sr = counting
        p2 = psi0.copy()
        e2 = make_envs(p2)
        rows = []
        try:
            for i in range(args.tol_sweeps):
                n0, q0 = nmv["n"], qr_snapshot()
                barrier()
                t0 = time.perf_counter()
                eps_t = alg.dmrg_sweep(p2, H, e2, mk.Arnoldi(tol=1e-12, krylovdim=30, maxiter=100), ws)
                barrier()
                dt_t = time.perf_counter() - t0
                q1 = qr_snapshot()
                rows.append({"sweep": i + 1, "sweeps_per_s": round(1.0 / dt_t, 4), "ms": round(dt_t * 1e3, 1),
                             "energy": energy(p2, e2), "matvecs": nmv["n"] - n0,
                             "matvecs_per_site_mean": round((nmv["n"] - n0) / (2 * L - 2), 2),
                             "max_galerkin": float(max(eps_t)), "qr_calls": {k: q1[k] - q0[k] for k in q1}})
        finally:
            krylov.eigsolve_sr = orig_eig
        tot_ms = sum(r["ms"] for r in rows)
        to_tol = {"from": "psi0 (fresh copy)", "eigensolver": "Arnoldi(tol=1e-12, krylovdim=30, maxiter=100, eager)",
                  "sweeps_per_s": round(len(rows) / (tot_ms * 1e-3), 4),
                  "matvecs_total": sum(r["matvecs"] for r in rows),
                  "matvecs_per_site_mean": round(sum(r["matvecs"] for r in rows) / (len(rows) * (2 * L - 2)), 2),
                  "sweeps": rows}
        del p2, e2

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
            "operator_applications_per_site": args.matvecs,       # the galerkin image reuses the eigensolver's first matvec
            "early_sweeps": early,
            "to_tolerance": to_tol,
            "max_galerkin_last_sweep": None if eps is None else float(max(eps)),
            "qr_calls_timed": qr_timed,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, D, d, args.matvecs)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_shard:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
