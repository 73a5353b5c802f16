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
    assert set(tr) == set(gen.SWEEP_CASES)
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
