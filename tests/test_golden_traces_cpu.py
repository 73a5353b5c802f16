"""CPU side of the committed ORACLE fixtures (tests/golden/make_sweep_traces.py): the oracle still reproduces them
(regression pin), config 1 of BASELINE.json (TFI L=16 D=4, "reference CPU path, no GPU") runs at its stated size on the
oracle AND through the product's host code on the stand-in backend, and its energy obeys the exact free-fermion bound."""
import json
import os
import sys

import numpy as np
import pytest

import mpskit_oracle as mo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_sweep_traces as gen  # noqa: E402


def _traces():
    with open(os.path.join(GOLD, "sweep_traces.json")) as f:
        return json.load(f)


def test_fixture_files_cover_the_baseline_configs():
    tr = _traces()
    assert set(tr) == set(gen.SWEEP_CASES) | set(gen.VUMPS_CASES)
    ts = np.load(os.path.join(GOLD, "tsplit_4096.npz"))
    assert ts["S"].shape == (gen.TSPLIT_N,) and int(ts["keep"]) == gen.TSPLIT_KEEP
    fx = np.load(os.path.join(GOLD, "projected_D1024.npz"))
    for case, c in gen.PROJ_CASES.items():
        for op in c["ops"]:
            assert fx[f"{case}.{op}.proj"].shape == (gen.NPROJ,) and fx[f"{case}.{op}.samp"].shape == (gen.NSAMP,)


def test_config1_tfi_L16_D4_oracle_and_exact_bound():
    """BASELINE config 1 at its stated size: the oracle reproduces the committed trace sweep by sweep; the converged
    D = 4 energy is variational w.r.t. the exact free-fermion ground energy and within 2e-3 of it."""
    fx = _traces()["c1_tfi_L16_D4"]
    out = gen.run_sweep_case(mo, "c1_tfi_L16_D4")
    assert len(out["trace"]) == len(fx["trace"])
    for (i, E, e), (io, Eo, eo) in zip(out["trace"], fx["trace"]):
        assert i == io and abs(E - Eo) <= 1e-12 * abs(Eo)
    assert np.abs(np.array(out["schmidt"]) - np.array(fx["schmidt"])).max() < 1e-10
    E, exact = fx["trace"][-1][1], fx["exact_ground_energy"]
    assert abs(exact - gen._tfi_obc_exact(16, 1.0, 1.0)) < 1e-12
    assert exact - 1e-10 <= E <= exact + 2e-3 * abs(exact)
    assert fx["trace"][-1][2] < 1e-12          # converged to the reference's default tol


def test_free_fermion_formula_matches_dense_ed():
    for L, g in ((8, 1.0), (9, 0.7)):
        e0 = np.linalg.eigvalsh(mo.dense_hamiltonian(mo.tfi_mpo(1.0, g), L))[0]
        assert abs(e0 - gen._tfi_obc_exact(L, 1.0, g)) < 1e-12 * abs(e0)


def test_config1_product_host_code_matches_trace():
    """The PRODUCT's DMRG driver (host logic on the CPU stand-in backend) follows the committed config-1 trace."""
    import mpskit_jl_amd as mk
    from cpu_backend import CpuBackend
    cb = CpuBackend()
    fx = _traces()["c1_tfi_L16_D4"]
    H = mk.transverse_field_ising(1.0, 1.0, be=cb)
    psi = mk.FiniteMPS(gen.initial_tensors("c1_tfi_L16_D4"), normalize=True, be=cb)
    got = []

    def record(it, p, Hh, envs):
        got.append(float(np.sum(mk.expectation_value(p, Hh, envs))))
        return p, envs

    mk.find_groundstate(psi, H, mk.DMRG(tol=1e-12, maxiter=len(fx["trace"]), finalize=record))
    assert len(got) == len(fx["trace"])
    for E, (_, Eo, _) in zip(got, fx["trace"]):
        assert abs(E - Eo) <= 1e-10 * abs(Eo)


def test_projected_fixture_oracle_regression():
    """The cheapest projected case is recomputed (north-star dAC, D = 1024: ~2 s of BLAS) and must hit the committed
    projections to rounding: guards the fixture generator and the seeds the GPU test regenerates its inputs from."""
    fx = np.load(os.path.join(GOLD, "projected_D1024.npz"))
    case = "ns_heis_D1024"
    H = gen.proj_hamiltonian(mo, case)[0]
    inp = gen.projected_inputs(case, H.chil)
    y = mo.dAC(inp["x"], H, inp["GL"], inp["GR"])
    proj, samp, nrm = gen.reduce_output(case, "dAC", y)
    assert abs(nrm - float(fx[f"{case}.dAC.norm"])) <= 1e-13 * nrm
    assert np.abs(proj - fx[f"{case}.dAC.proj"]).max() <= 1e-12 * nrm
    assert np.abs(samp - fx[f"{case}.dAC.samp"]).max() <= 1e-12 * nrm / np.sqrt(y.size) * 10


def test_round3_fixtures_are_pinned_by_independent_answers():
    """The fixtures added in round 3 against answers that do not come from the oracle's own drivers:
    * Hubbard L = 8 two-site DMRG at the exact bond dimension (4^4 = 256) == sparse ED of the same MPO (1e-12);
    * VUMPS iTFI g = 0.5 at D = 64 == the energy density the reference's docs record (3.ising-dqpt/index.md:118);
    * the 4096 LAPACK singular values of the seeded theta == the designed spectrum to LAPACK's absolute accuracy, and the
      stored discarded weight is the tail sum at k = 1024."""
    tr = _traces()
    hb = tr["c4_hubbard_L8_D256_exact"]
    assert abs(hb["trace"][-1][1] - hb["ed_ground_energy"]) <= 1e-12 * abs(hb["ed_ground_energy"])
    assert max(hb["schmidt"]) < 1.0 and len(hb["schmidt"]) == 256
    # the sparse ED builder itself against the oracle's dense Hamiltonian (L = 4: 256 x 256) and dense ED
    H = mo.hubbard_mpo(1.0, 4.0)
    e4 = np.linalg.eigvalsh(mo.dense_hamiltonian(H, 4))[0]
    assert abs(gen.ed_ground_energy(mo, H, 4) - e4) <= 1e-11 * abs(e4)
    v = tr["c3_itfi_D64"]
    assert abs(v["trace"][-1][1] - (-1.063544409973)) < 2e-12
    assert v["trace"][-1][2] < 1e-11
    ts = np.load(os.path.join(GOLD, "tsplit_4096.npz"))
    design = np.sort(gen.tsplit_spectrum())[::-1]
    assert np.abs(ts["S"] - design).max() <= 1e-14 * design[0]
    assert abs(float(ts["disc"]) - np.sqrt(np.sum(design[gen.TSPLIT_KEEP:] ** 2))) <= 1e-9 * float(ts["disc"])
    assert design[1022] == design[1025] and design[1021] > design[1022] > design[1026]      # the quadruplet across the cut


def test_vumps_fixture_oracle_regression():
    """The oracle still reproduces the committed config-3 trace at D = 64 iteration by iteration (about 3 s)."""
    fx = _traces()["c3_itfi_D64"]
    out = gen.run_vumps_case(mo, "c3_itfi_D64")
    assert len(out["trace"]) == len(fx["trace"])
    for (i, E, e), (io, Eo, eo) in zip(out["trace"], fx["trace"]):
        assert i == io and abs(E - Eo) <= 1e-12 * abs(Eo)
