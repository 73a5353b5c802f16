"""Generates tests/golden/sweep_traces.json and tests/golden/projected_D1024.npz.

Both are ORACLE outputs computed once in the build container (the reference is Julia and cannot be
run here or on the GPU box), committed so that the `-m gpu` tests can hold the
HIP path to them at sizes where a live oracle run would take minutes:

  sweep_traces.json   per-sweep (energy, galerkin / fidelity error) of the oracle's DMRG / DMRG2 on the
                      BASELINE configs (config 1 at its stated size; configs 2 and 4 at CPU-feasible
                      cuts) from seeded random initial states, plus the final middle-bond Schmidt spectrum.
                      The initial tensors are NOT stored: tests regenerate them from the seed with the
                      same `numpy.random.default_rng` calls (`initial_tensors` below is the single
                      definition both sides import).
  projected_D1024.npz the oracle's dAC / dAC2 / transfer_left / transfer_right / dC outputs at D = 1024
                      (north-star point d=2, W=5 and the config-4 shape d=4, W=6) reduced to 64 random
                      projections + 512 strided samples each (inputs and projection vectors are
                      regenerated from seeds, `projected_inputs` below).

Round 3 additions:
  sweep_traces.json   + VUMPS per-ITERATION traces `c3_itfi_D64`, `c3_itfi_D128` (config 3: infinite TFI, seeded A, oracle
                      `mo.vumps` with the reference's dynamic tolerances: energy density + galerkin per iteration);
                      + `c4_hubbard_L8_D256_exact`: Hubbard DMRG2 at the bond dimension where L = 8 is exact (4^4 = 256),
                      with the sparse-ED ground energy of the same MPO next to the oracle trace.
  projected_D1024.npz + `c3_tfi_D512`: dAC / dC / transfer_left / transfer_right with the MPO and the MPO-less bond
                      transfers (`tl0`, `tr0`) at the config-3 shape (D = 512, d = 2, W = 3).
  tsplit_4096.npz     all 4096 singular values (LAPACK, numpy.linalg.svd) + the discarded weight at k = 1024 of a seeded
                      graded 4096 x 4096 theta with exact multiplets (`tsplit_theta` below regenerates theta).

Run:  python tests/golden/make_sweep_traces.py [case ...]        (about 20 minutes on 8 cores)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SEED = 20240213

# name -> (model, model args, L, d, D, algorithm, sweeps, truncdim)
SWEEP_CASES = {
    # BASELINE config 1 at its stated size
    "c1_tfi_L16_D4": dict(model="tfi", args=(1.0, 1.0), L=16, d=2, D=4, alg="dmrg", sweeps=12),
    # BASELINE config 2 (Heisenberg S=1, 1-site DMRG, L=100 D=256): full length at D=64, full D at L=20
    "c2_heis1_L100_D64": dict(model="heis", args=(1.0,), L=100, d=3, D=64, alg="dmrg", sweeps=3),
    "c2_heis1_L20_D256": dict(model="heis", args=(1.0,), L=20, d=3, D=256, alg="dmrg", sweeps=2),
    # BASELINE config 4 (Hubbard, 2-site DMRG with tsvd truncation, L=64 D=1024) cut to L=12 D=128
    "c4_hubbard_L12_D128": dict(model="hubbard", args=(1.0, 4.0), L=12, d=4, D=128, alg="dmrg2", sweeps=2),
    # headline model (Heisenberg S=1/2) at a cut
    "hl_heis_L40_D128": dict(model="heis", args=(0.5,), L=40, d=2, D=128, alg="dmrg", sweeps=3),
    # config 4 at the size where the two-site algorithm is EXACT (max bond 4^4 = 256, nothing truncated): pinned by sparse ED
    "c4_hubbard_L8_D256_exact": dict(model="hubbard", args=(1.0, 4.0), L=8, d=4, D=256, alg="dmrg2", sweeps=4),
}

# config 3 (VUMPS on the infinite transverse-field Ising chain): per-iteration traces of the oracle's vumps
VUMPS_CASES = {
    # g = 0.5: the model of the reference's recorded energy density -1.063544409973 (docs/.../3.ising-dqpt/index.md:118);
    # D = 64 is far beyond its entanglement (singular bond matrices: the stress case for the uniform gauge / regularised envs)
    "c3_itfi_D64": dict(model="tfi", args=(1.0, 0.5), d=2, D=64, n=1, iters=10),
    # g = 0.9: close to criticality, every Schmidt value of a D = 128 bond is populated
    "c3_itfi_D128": dict(model="tfi", args=(1.0, 0.9), d=2, D=128, n=1, iters=8),
}


def bond_dims(L, d, D):
    """finitemps.jl:182-192."""
    dims = [1]
    for _ in range(1, L):
        dims.append(min(dims[-1] * d, D))
    dims.append(1)
    for k in range(L - 1, 0, -1):
        dims[k] = min(dims[k], dims[k + 1] * d)
    return dims


# Seed ids are FROZEN per case (rounds 1-2 derived them from the sorted position; adding a case must not move the others)
SWEEP_SEED_ID = {"c1_tfi_L16_D4": 0, "c2_heis1_L100_D64": 1, "c2_heis1_L20_D256": 2, "c4_hubbard_L12_D128": 3,
                 "hl_heis_L40_D128": 4, "c4_hubbard_L8_D256_exact": 5}
PROJ_SEED_ID = {"c4_hubbard_D1024": 0, "ns_heis_D1024": 1, "tfi2_D768x1024": 2, "c3_tfi_D512": 3}
VUMPS_SEED_ID = {"c3_itfi_D64": 1, "c3_itfi_D128": 0}


def initial_tensors(case):
    """Seeded uniform[0,1) site tensors of a case (the reference's `rand`, abstractmps.jl:34-36)."""
    c = SWEEP_CASES[case]
    rng = np.random.default_rng([SEED, SWEEP_SEED_ID[case]])
    dims = bond_dims(c["L"], c["d"], c["D"])
    return [rng.random((dims[i], c["d"], dims[i + 1])) for i in range(c["L"])]


def vumps_initial_tensors(case):
    c = VUMPS_CASES[case]
    rng = np.random.default_rng([SEED, 500 + VUMPS_SEED_ID[case]])
    return [rng.random((c["D"], c["d"], c["D"])) for _ in range(c["n"])]


def run_vumps_case(mo, case):
    c = VUMPS_CASES[case]
    H = {"tfi": mo.tfi_mpo}[c["model"]](*c["args"])
    t0 = time.time()
    psi = mo.InfiniteMPS.from_tensors(vumps_initial_tensors(case))
    psi, envs, eps, log = mo.vumps(psi, H, tol=1e-14, maxiter=c["iters"])
    out = dict(c)
    out["args"] = list(c["args"])
    out.update(trace=[[int(i), float(E), float(e)] for i, E, e in log],
               schmidt=[float(x) for x in np.linalg.svd(psi.CR[0], compute_uv=False)], seconds=round(time.time() - t0, 1))
    return out


def sparse_hamiltonian(H, L):
    """scipy.sparse twin of mo.dense_hamiltonian (same boundary convention, FinEnv.jl:41-70) for ED beyond dense sizes."""
    import scipy.sparse as sp
    d = H.d
    cur = None
    for i in range(L):
        Of = H[i].full()
        if cur is None:
            cur = [sp.csr_matrix(Of[0, :, :, v]) for v in range(Of.shape[3])]
        else:
            dim = cur[0].shape[0]
            new = []
            for v in range(Of.shape[3]):
                acc = sp.csr_matrix((dim * d, dim * d))
                for w in range(Of.shape[0]):
                    if np.any(Of[w, :, :, v] != 0) and cur[w].nnz:
                        acc = acc + sp.kron(cur[w], sp.csr_matrix(Of[w, :, :, v]), format="csr")
                new.append(acc)
            cur = new
    return cur[-1]


def ed_ground_energy(mo, H, L):
    import scipy.sparse.linalg as sla
    Hs = sparse_hamiltonian(H, L)
    if L <= 4:      # the sparse builder against the oracle's dense one
        assert np.abs(Hs.toarray() - mo.dense_hamiltonian(H, L)).max() < 1e-14
    Hs = (Hs + Hs.T) * 0.5
    w = sla.eigsh(Hs, k=2, which="SA", tol=1e-13, ncv=64, return_eigenvectors=False)
    return float(np.min(w))


def oracle_hamiltonian(mo, case):
    c = SWEEP_CASES[case]
    return {"tfi": mo.tfi_mpo, "heis": mo.heisenberg_mpo, "hubbard": mo.hubbard_mpo}[c["model"]](*c["args"])


def run_sweep_case(mo, case):
    c = SWEEP_CASES[case]
    H = oracle_hamiltonian(mo, case)
    psi = mo.FiniteMPS(initial_tensors(case), normalize=True)
    t0 = time.time()
    if c["alg"] == "dmrg":
        psi, envs, eps, log = mo.dmrg(psi, H, tol=1e-12, maxiter=c["sweeps"])
    else:
        psi, envs, eps, log = mo.dmrg2(psi, H, truncdim=c["D"], tol=1e-12, maxiter=c["sweeps"])
    mid = c["L"] // 2 - 1
    spec = np.linalg.svd(psi.CR(mid), compute_uv=False)
    out = dict(c)
    out["args"] = list(c["args"])
    out.update(trace=[[int(i), float(E), float(e)] for i, E, e in log],
               mid_bond=mid, schmidt=[float(s) for s in spec], seconds=round(time.time() - t0, 1))
    if case.endswith("_exact"):
        Hed = oracle_hamiltonian(mo, case)
        ed_ground_energy(mo, Hed, 4)                       # self-check of the sparse builder at a dense-checkable size
        out["ed_ground_energy"] = ed_ground_energy(mo, Hed, c["L"])
    if c["model"] == "tfi":
        # exactly solvable: pins the oracle itself (variational bound, and D = 4 at L = 16 is within 1e-3 of it)
        out["exact_ground_energy"] = _tfi_obc_exact(c["L"], *c["args"])
    return out


def _tfi_obc_exact(L, J, g):
    """Free-fermion ground energy of H = -J sum_{i<L} Z_i Z_{i+1} - g sum X_i (open chain): minus the sum of the
    singular values of the L x L bidiagonal matrix with g on the diagonal and J on the sub-diagonal."""
    M = np.diag(np.full(L, g)) + np.diag(np.full(L - 1, J), -1)
    return float(-np.sum(np.linalg.svd(M, compute_uv=False)))


# ---- D = 1024 projected operator outputs ---------------------------------------------------------

PROJ_CASES = {
    # name: (model, args, D, d, ops)
    "ns_heis_D1024": dict(model="heis", args=(0.5,), D=1024, d=2, ops=("dAC", "dC", "tl", "tr")),
    "c4_hubbard_D1024": dict(model="hubbard", args=(1.0, 4.0), D=1024, d=4, ops=("dAC", "dAC2")),
    # two-site-decomposed MPO with chi > 1 levels (dense blocks), ragged bond dims
    "tfi2_D768x1024": dict(model="tfi2", args=(1.3,), D=(768, 1024), d=2, ops=("dAC", "tl", "tr")),
    # BASELINE config 3 shape (infinite TFI, D = 512, d = 2, W = 3): matvecs + MPO and MPO-less transfers
    "c3_tfi_D512": dict(model="tfi", args=(1.0, 0.5), D=512, d=2, ops=("dAC", "dC", "tl", "tr", "tl0", "tr0")),
}
NPROJ, NSAMP = 64, 512


def projected_inputs(case, chis):
    """Seeded inputs of a projected case: environments as lists of [D, chi_i, D] blocks, x, c, x2, A, Ab (entries
    uniform[-0.5, 0.5) so that no output is dominated by the all-positive mean)."""
    c = PROJ_CASES[case]
    rng = np.random.default_rng([SEED, 1000 + PROJ_SEED_ID[case]])
    Dl, Dr = c["D"] if isinstance(c["D"], tuple) else (c["D"], c["D"])
    d = c["d"]
    r = lambda *s: rng.random(s) - 0.5
    inp = dict(GL=[r(Dl, ch, Dl) for ch in chis], GR=[r(Dr, ch, Dr) for ch in chis], x=r(Dl, d, Dr))
    if "dC" in c["ops"]:
        inp["c"] = r(Dl, Dr)
    if "dAC2" in c["ops"]:
        inp["x2"] = r(Dl, d, Dr, d)
    if any(o in c["ops"] for o in ("tl", "tr", "tl0", "tr0")):
        inp["A"], inp["Ab"] = r(Dl, d, Dr), r(Dl, d, Dr)
    return inp


def reduce_output(case, op, y):
    """64 random projections + 512 strided samples of a flattened (column-major) output."""
    y = np.ravel(np.asarray(y), order="F")
    rng = np.random.default_rng([SEED, 2000 + PROJ_SEED_ID[case], sum(map(ord, op))])
    proj = np.empty(NPROJ)
    for k in range(NPROJ):
        p = rng.random(y.size) - 0.5
        proj[k] = p @ y
    idx = (np.arange(NSAMP, dtype=np.int64) * 7919 * 131) % y.size
    return proj, y[idx], float(np.linalg.norm(y))


def proj_hamiltonian(mo, case):
    c = PROJ_CASES[case]
    return {"heis": mo.heisenberg_mpo, "hubbard": mo.hubbard_mpo, "tfi2": mo.tfi_twosite_mpo,
            "tfi": mo.tfi_mpo}[c["model"]](*c["args"])


def run_proj_case(mo, case):
    c = PROJ_CASES[case]
    H = proj_hamiltonian(mo, case)[0]
    inp = projected_inputs(case, H.chil)
    out = {}
    for op in c["ops"]:
        t0 = time.time()
        if op == "dAC":
            y = mo.dAC(inp["x"], H, inp["GL"], inp["GR"])
        elif op == "dC":
            y = mo.dC(inp["c"], inp["GL"], inp["GR"])
        elif op == "dAC2":
            y = mo.dAC2(inp["x2"], H, H, inp["GL"], inp["GR"])
        elif op == "tl":
            y = np.concatenate(mo.transfer_left(inp["GL"], H, inp["A"], inp["Ab"]), axis=1)     # [Drb, W, Dr]
            y = np.transpose(y, (0, 2, 1))                                                      # slabs: [Drb, Dr, W]
        elif op == "tr":
            y = np.concatenate(mo.transfer_right(inp["GR"], H, inp["A"], inp["Ab"]), axis=1)
            y = np.transpose(y, (0, 2, 1))
        elif op == "tl0":      # MPO-less transfer of every slab (transfer.jl:18-25), slabs: [Drb, Dr, W]
            y = np.stack([mo.transfer_left_bond(g[:, k, :], inp["A"], inp["Ab"]) for g in inp["GL"] for k in range(g.shape[1])], axis=2)
        elif op == "tr0":      # (transfer.jl:38-45)
            y = np.stack([mo.transfer_right_bond(g[:, k, :], inp["A"], inp["Ab"]) for g in inp["GR"] for k in range(g.shape[1])], axis=2)
        proj, samp, nrm = reduce_output(case, op, y)
        out[f"{case}.{op}.proj"], out[f"{case}.{op}.samp"], out[f"{case}.{op}.norm"] = proj, samp, np.array(nrm)
        print(f"  {case}.{op}: |y| = {nrm:.6e}  ({time.time() - t0:.1f} s)", flush=True)
    return out


# ---- LAPACK-pinned singular values at the config-4 split size --------------------------------------

TSPLIT_N, TSPLIT_KEEP = 4096, 1024


def tsplit_spectrum(n=TSPLIT_N):
    """DMRG-like graded spectrum: exponential decay reaching 1.2e-6 at index 1023, floor 1e-13 (numerically rank deficient),
    exact multiplets of size 2 / 3 / 4 every 16 values (SU(2)-like degeneracies; one straddles the cut at 1024)."""
    i = np.arange(n, dtype=float)
    s = np.maximum(np.exp(-i / 75.0), 1e-13)
    for start in range(5, n - 4, 16):
        mult = 2 + (start // 16) % 3
        s[start:start + mult] = s[start]
    s[1022:1026] = s[1022]               # a quadruplet across the truncation point
    return s


def tsplit_theta(n=TSPLIT_N):
    """theta = U diag(s) V^T with seeded Haar-like U, V (QR of Gaussian matrices): the SAME calls on both sides."""
    rng = np.random.default_rng([SEED, 4096])
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (U * tsplit_spectrum(n)) @ V.T


def run_tsplit_fixture():
    t0 = time.time()
    th = tsplit_theta()
    S = np.linalg.svd(th, compute_uv=False)
    disc = float(np.sqrt(np.sum(S[TSPLIT_KEEP:] ** 2)))
    print(f"  tsplit_4096: LAPACK svd in {time.time() - t0:.0f} s, S[0] = {S[0]:.6f}, S[1023] = {S[1023]:.6e}, disc = {disc:.6e}, "
          f"max |S - design| / S[0] = {np.abs(S - np.sort(tsplit_spectrum())[::-1]).max():.2e}", flush=True)
    np.savez_compressed(os.path.join(HERE, "tsplit_4096.npz"), S=S, disc=np.array(disc), keep=np.array(TSPLIT_KEEP),
                        theta_fro=np.array(np.linalg.norm(th)), theta_samples=th.ravel()[:: 1048583][:16])


def main(argv):
    import mpskit_oracle as mo
    want = set(argv)
    tr_path = os.path.join(HERE, "sweep_traces.json")
    traces = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
    for case in SWEEP_CASES:
        if want and case not in want:
            continue
        print("sweep case", case, flush=True)
        traces[case] = run_sweep_case(mo, case)
        print("  ", traces[case]["trace"], traces[case]["seconds"], "s", flush=True)
        with open(tr_path, "w") as f:
            json.dump(traces, f, indent=1, sort_keys=True)
    for case in VUMPS_CASES:
        if want and case not in want:
            continue
        print("vumps case", case, flush=True)
        traces[case] = run_vumps_case(mo, case)
        print("  ", traces[case]["trace"], traces[case]["seconds"], "s", flush=True)
        with open(tr_path, "w") as f:
            json.dump(traces, f, indent=1, sort_keys=True)
    if not want or "tsplit_4096" in want:
        run_tsplit_fixture()
    pj_path = os.path.join(HERE, "projected_D1024.npz")
    proj = dict(np.load(pj_path)) if os.path.exists(pj_path) else {}
    for case in PROJ_CASES:
        if want and case not in want:
            continue
        print("projected case", case, flush=True)
        proj.update(run_proj_case(mo, case))
        np.savez_compressed(pj_path, **proj)


if __name__ == "__main__":
    main(sys.argv[1:])
