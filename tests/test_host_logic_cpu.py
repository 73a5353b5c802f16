"""Host logic of the PRODUCT package on a host stand-in backend (tests/cpu_backend.py): lazy-gauge
state machine, FinEnv invalidation, Krylov solvers, DMRG driver -- compared with the oracle."""
import numpy as np
import pytest

import mpskit_oracle as mo
import mpskit_jl_amd as mk
from mpskit_jl_amd import krylov, algorithms as alg
from cpu_backend import CpuBackend


@pytest.fixture()
def cb():
    return CpuBackend()


def _pair(cb, L=6, d=2, D=6, seed=0):
    rng = np.random.default_rng(seed)
    dims = mo.FiniteMPS.random(L, d, D, np.random.default_rng(0)).bond_dims()
    As = [rng.random((1 if i == 0 else dims[i - 1], d, dims[i])) for i in range(L)]
    return mk.FiniteMPS(As, normalize=True, be=cb), mo.FiniteMPS(As, normalize=True)


def test_lazy_gauge_state_machine_matches_oracle(cb):
    """orthoview.jl:1-143 : same reads/writes -> same tensors and same None-pattern as the oracle."""
    pg, po = _pair(cb)
    pat = lambda xs: [x is None for x in xs]
    for i in (5, 2, 0, 3):
        assert np.abs(cb.download(pg.AC(i)) - po.AC(i)).max() < 1e-13
        assert pat(pg.ALs) == pat(po.ALs) and pat(pg.ARs) == pat(po.ARs) and pat(pg.CLs) == pat(po.CLs)
    new = np.random.default_rng(1).random(po.AC(2).shape)
    pg.set_AC(2, cb.upload(new))
    po.set_AC(2, new)
    assert pat(pg.ALs) == pat(po.ALs) and pat(pg.ARs) == pat(po.ARs) and pat(pg.ACs) == pat(po.ACs)
    for i in range(6):
        assert np.abs(cb.download(pg.AL(i)) - po.AL(i)).max() < 1e-12
        assert np.abs(cb.download(pg.AR(i)) - po.AR(i)).max() < 1e-12


def test_finenv_invalidation_counts_match_oracle(cb):
    """FinEnv.jl:114-145 : identity-based invalidation rebuilds exactly the same environments."""
    pg, po = _pair(cb)
    Hg, Ho = mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)
    eg, eo = mk.FinEnv(pg, Hg), mo.FinEnv(po, Ho)
    rng = np.random.default_rng(2)
    for pos in (0, 3, 5, 2, 2, 4):
        gl = cb.download_env(eg.leftenv(pos, pg), [1] * 5)
        for a, b in zip(gl, eo.leftenv(pos, po)):
            assert np.abs(a - b).max() < 1e-12
        gr = cb.download_env(eg.rightenv(pos, pg), [1] * 5)
        for a, b in zip(gr, eo.rightenv(pos, po)):
            assert np.abs(a - b).max() < 1e-12
        assert eg.n_transfers == eo.n_transfers
        new = rng.random(po.AC(pos).shape)
        pg.set_AC(pos, cb.upload(new))
        po.set_AC(pos, new)
    assert cb.calls["transfer_left"] + cb.calls["transfer_right"] == eg.n_transfers


def test_krylov_eigsolve_matches_dense(cb):
    rng = np.random.default_rng(3)
    n = 40
    M = rng.standard_normal((n, n))
    M = M + M.T
    mv = lambda x, out: cb._set(out, M @ cb.download(x))
    lam, vec, nmv, res = krylov.eigsolve_sr(cb, mv, cb.upload(rng.random(n)), tol=1e-12, krylovdim=12)
    w = np.linalg.eigvalsh(M)
    assert abs(lam - w[0]) < 1e-10
    v = cb.download(vec)
    assert np.linalg.norm(M @ v - lam * v) < 1e-8
    # fixed budget mode does exactly that many matvecs
    _, _, nmv, _ = krylov.eigsolve_sr(cb, mv, cb.upload(rng.random(n)), fixed_matvecs=5, krylovdim=5)
    assert nmv == 5
    # gmres
    Apd = M @ M.T + n * np.eye(n)
    b = rng.random(n)
    x = krylov.gmres(cb, lambda x, out: cb._set(out, Apd @ cb.download(x)), cb.upload(b), cb.upload(np.zeros(n)),
                     tol=1e-12)
    assert np.linalg.norm(Apd @ cb.download(x) - b) < 1e-9
    # dominant eigenpair of a positive matrix
    P = rng.random((n, n))
    lamP, vP = krylov.eigsolve_lm_real(cb, lambda x, out: cb._set(out, P @ cb.download(x)), cb.upload(rng.random(n)))
    assert abs(lamP - np.max(np.abs(np.linalg.eigvals(P)))) < 1e-9


@pytest.mark.parametrize("n", [1, 2, 4, 6])
@pytest.mark.parametrize("sync_free", [True, False])
def test_krylov_fixed_budget_larger_than_space(cb, n, sync_free):
    """A fixed matvec budget larger than the vector-space dimension (chain-edge sites of the benchmark sweep: 4 < 8)
    must stop at the Krylov breakdown instead of renormalising a rounding-level residual into the basis: the Ritz
    value is never below the spectrum and the returned vector is the eigenvector (product AND oracle)."""
    rng = np.random.default_rng(10 + n)
    M = rng.standard_normal((n, n))
    M = M + M.T
    w, U = np.linalg.eigh(M)
    x0 = rng.random(n)
    mv = lambda x, out: cb._set(out, M @ cb.download(x))
    kd = 8 if sync_free else 5       # fixed_matvecs <= krylovdim takes the sync-free recurrence, otherwise the host loop
    lam, vec, _, _ = krylov.eigsolve_sr(cb, mv, cb.upload(x0), fixed_matvecs=8, krylovdim=kd)
    v = cb.download(vec)
    assert lam >= w[0] - 1e-12 and abs(lam - w[0]) < 1e-10
    assert abs(v @ M @ v - w[0]) < 1e-10 and abs(abs(v @ U[:, 0]) - 1) < 1e-9
    lo, vo, _ = mo.eigsolve_sr(lambda x: M @ x, x0, fixed_matvecs=8, krylovdim=kd)
    assert abs(lo - w[0]) < 1e-10 and abs(abs(vo @ U[:, 0]) - 1) < 1e-9


@pytest.mark.parametrize("model", ["heis", "tfi"])
def test_dmrg_driver_matches_oracle(cb, model):
    """dmrg.jl:22-55 on the product's host code == the oracle's restatement (energies to 1e-10)."""
    Hg, Ho = ((mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)) if model == "heis"
              else (mk.transverse_field_ising(1.0, 0.6, be=cb), mo.tfi_mpo(1.0, 0.6)))
    pg, po = _pair(cb, L=6, D=8, seed=5)
    p, e, eps = mk.find_groundstate(pg, Hg, mk.DMRG(tol=1e-10, maxiter=8))
    _, _, epso, logo = mo.dmrg(po, Ho, tol=1e-10, maxiter=8)
    E = float(np.sum(mk.expectation_value(p, Hg, e)))
    assert eps < 1e-9
    assert abs(E - logo[-1][1]) < 1e-10 * abs(E)
    e0 = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, 6))[0]
    assert abs(E - e0) < 1e-9


def test_energy_slice_reproduces_expval(cb):
    """expval.jl:92-109 through the 'energy slice' trick == the oracle's block-by-block sum."""
    Hg, Ho = mk.transverse_field_ising(1.0, 0.8, be=cb), mo.tfi_mpo(1.0, 0.8)
    pg, po = _pair(cb, seed=7)
    eg = mk.expectation_value(pg, Hg, mk.FinEnv(pg, Hg))
    eo = mo.expectation_value(po, Ho, mo.FinEnv(po, Ho))
    assert np.abs(eg - eo).max() < 1e-12


def test_mpohamiltonian_from_twosite_matches_oracle(cb):
    """mpohamiltonian.jl:16-31 (SVD split) in the product == oracle; chi > 1 blocks."""
    X = np.array([[0.0, 1], [1, 0]])
    Z = np.array([[1.0, 0], [0, -1]])
    h2 = -(np.kron(Z, Z) + 0.4 * (np.kron(X, np.eye(2)) + np.kron(np.eye(2), X))).reshape(2, 2, 2, 2)
    Hg, Ho = mk.from_twosite(h2, be=cb), mo.tfi_twosite_mpo(0.8)
    assert Hg[0].chil == Ho[0].chil
    assert np.abs(Hg[0].oracle.full() - Ho[0].full()).max() < 1e-12


def test_tdvp_driver_matches_oracle(cb):
    """tdvp.jl:61-94 on the product's host code (CRView.setindex!, integrate, exponentiate) == oracle,
    imaginary-time step at truncated bond dimension; exact vs dense expm at full bond dimension."""
    import scipy.linalg as sla
    Hg, Ho = mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)
    pg, po = _pair(cb, L=6, D=4, seed=9)
    p1, e1 = mk.timestep(pg, Hg, 0.0, -0.1j, mk.TDVP())
    q1, f1 = mo.tdvp_timestep(po, Ho, 0.0, -0.1j)
    assert abs(p1.norm() - q1.norm()) < 1e-11
    Eg = float(np.sum(mk.expectation_value(p1, Hg, e1)))
    Eo = float(np.sum(mo.expectation_value(q1, Ho, f1)).real)
    assert abs(Eg - Eo) < 1e-10 * abs(Eo)
    for i in range(6):
        assert np.abs(cb.download(p1.AC(i)) - q1.AC(i)).max() < 1e-10
    # full bond dimension: exact
    pg, po = _pair(cb, L=6, D=64, seed=10)
    v0 = mo.mps_to_vector(po)
    p1, _ = mk.timestep(pg, Hg, 0.0, -0.2j, mk.TDVP())
    host = mo.FiniteMPS.__new__(mo.FiniteMPS)
    vec = np.ones((1, 1))
    for i in range(6):
        A = cb.download(p1.AL(i)) if i < 5 else cb.download(p1.AC(5))
        vec = np.tensordot(vec, A, axes=([vec.ndim - 1], [0])).reshape(-1, A.shape[2])
    ex = sla.expm(-0.2 * mo.dense_hamiltonian(Ho, 6)) @ v0
    assert np.abs(vec.reshape(-1) - ex).max() < 1e-11
    with pytest.raises(NotImplementedError):
        mk.timestep(pg, Hg, 0.0, 0.1, mk.TDVP())


def test_lazysum_matches_summed_hamiltonian(cb):
    """lazysum.jl / multipleenv.jl / derivatives.jl:310-323 (reference test: test/operators.jl:173-280):
    LazySum([H_zz, H_x]) acts like the TFI Hamiltonian H_zz + H_x -- matvec linearity, expectation values and the
    DMRG ground state energy."""
    g = 0.7
    Hzz, Hx = mk.transverse_field_ising(1.0, 0.0, be=cb), mk.transverse_field_ising(0.0, g, be=cb)
    Hfull, Ho = mk.transverse_field_ising(1.0, g, be=cb), mo.tfi_mpo(1.0, g)
    Hl = mk.LazySum([Hzz, Hx])
    pg, po = _pair(cb, L=6, D=8, seed=13)
    el, ef = mk.environments(pg, Hl), mk.environments(pg, Hfull)
    assert isinstance(el, mk.MultipleEnvironments)
    x = pg.AC(2)
    yl, yf = cb.download(mk.ddAC(2, pg, Hl, el)(x)), cb.download(mk.ddAC(2, pg, Hfull, ef)(x))
    assert np.abs(yl - yf).max() < 1e-12 * max(1.0, np.abs(yf).max())
    assert np.abs(mk.expectation_value(pg, Hl, el) - mk.expectation_value(pg, Hfull, ef)).max() < 1e-12
    p, e, eps = mk.find_groundstate(pg, Hl, mk.DMRG(tol=1e-10, maxiter=8))
    E = float(np.sum(mk.expectation_value(p, Hl, e)))
    e0 = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, 6))[0]
    assert eps < 1e-9 and abs(E - e0) < 1e-9
    # prefactors: LazySum([H, H], [0.25, 0.75]) == H
    H2 = mk.LazySum([Hfull, Hfull], [0.25, 0.75])
    y2 = cb.download(mk.ddAC(2, pg, H2, mk.environments(pg, H2))(x))
    assert np.abs(y2 - yf).max() < 1e-12 * max(1.0, np.abs(yf).max())


def test_complex_states_by_bond_embedding(cb):
    """cplx.py: complex128 FiniteMPS carried as real 2x2 blocks on the bond indices.  The lazy gauge (QRpos / LQpos),
    the environments, expectation values, 1-site DMRG and a REAL-TIME TDVP step on the embedded tensors reproduce
    the oracle's complex arithmetic (the reference's default scalar type)."""
    from mpskit_jl_amd import cplx
    rng = np.random.default_rng(23)
    L, D = 6, 4
    dims = mo.FiniteMPS.random(L, 2, D, np.random.default_rng(0)).bond_dims()
    As = [rng.random((1 if i == 0 else dims[i - 1], 2, dims[i])) + 1j * rng.random((1 if i == 0 else dims[i - 1], 2, dims[i]))
          for i in range(L)]
    M1, M2 = rng.random((3, 4)) + 1j * rng.random((3, 4)), rng.random((4, 5)) + 1j * rng.random((4, 5))
    assert np.abs(cplx.extract(cplx.embed(M1) @ cplx.embed(M2)) - M1 @ M2).max() < 1e-14
    assert np.abs(cplx.embed(M1).T - cplx.embed(M1.conj().T)).max() == 0.0
    Hg, Ho = mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)
    pg, po = mk.FiniteMPS(As, normalize=True, be=cb), mo.FiniteMPS(As, normalize=True)
    assert pg.cplx and pg.bond_dims() == po.bond_dims() and abs(pg.norm() - 1) < 1e-13
    for i in range(L):                                   # same canonical form, structure kept by QRpos / LQpos
        for gt, ot in ((pg.AC(i), po.AC(i)), (pg.AR(i), po.AR(i)), (pg.AL(i), po.AL(i))):
            assert cplx.structure_defect(cb.download(gt)) < 1e-13
            assert np.abs(pg.download(gt) - ot).max() < 1e-12
    eg, eo = mk.FinEnv(pg, Hg), mo.FinEnv(po, Ho)
    Eg, Eo = mk.expectation_value(pg, Hg, eg), mo.expectation_value(po, Ho, eo)
    assert np.abs(Eg - Eo.real).max() < 1e-12
    # real-time TDVP step: tensors, norm and energy
    p1, e1 = mk.timestep(pg, Hg, 0.0, 0.1, mk.TDVP())
    q1, f1 = mo.tdvp_timestep(po, Ho, 0.0, 0.1)
    for i in range(L):
        assert np.abs(p1.download(p1.AC(i)) - q1.AC(i)).max() < 1e-10
    assert abs(p1.norm() - 1) < 1e-12
    assert abs(np.sum(mk.expectation_value(p1, Hg, e1)) - np.sum(Eo).real) < 1e-8      # energy conserved
    # mixed complex step and 1-site DMRG on the complex state
    p2, _ = mk.timestep(pg, Hg, 0.0, 0.05 - 0.02j, mk.TDVP())
    q2, _ = mo.tdvp_timestep(po, Ho, 0.0, 0.05 - 0.02j)
    assert np.abs(p2.download(p2.AC(2)) - q2.AC(2)).max() < 1e-10
    p3, e3, eps = mk.find_groundstate(pg, Hg, mk.DMRG(tol=1e-10, maxiter=8))
    _, _, epso, logo = mo.dmrg(po, Ho, tol=1e-10, maxiter=8)
    assert abs(np.sum(mk.expectation_value(p3, Hg, e3)) - logo[-1][1]) < 1e-10 * abs(logo[-1][1])


def test_complex_infinite_mps_vumps(cb):
    """complex128 InfiniteMPS via the bond embedding: uniform gauge (AL equals the oracle's complex AL; C only up to the
    gauge phase of the fixed point) and VUMPS energy."""
    from mpskit_jl_amd import cplx
    rng = np.random.default_rng(4)
    A = rng.random((6, 2, 6)) + 1j * rng.random((6, 2, 6))
    H, Ho = mk.transverse_field_ising(1.0, 0.5, be=cb), mo.tfi_mpo(1.0, 0.5)
    psi, po = mk.InfiniteMPS.from_tensors([A], be=cb), mo.InfiniteMPS.from_tensors([A])
    assert psi.cplx and cplx.structure_defect(cb.download(psi.AL[0])) < 1e-13
    assert np.abs(cplx.extract(cb.download(psi.AL[0])) - po.AL[0]).max() < 1e-12
    p, e, eps = mk.find_groundstate(psi, H, mk.VUMPS(tol=1e-10, maxiter=60))
    _, _, _, logo = mo.vumps(po, Ho, tol=1e-10, maxiter=60)
    assert eps < 1e-9 and abs(float(np.sum(mk.expectation_value(p, H, e))) - logo[-1][1]) < 1e-10


def test_finite_excited_states_match_ed(cb):
    """dmrgexcitation.jl:13-36 on the product's host code: the first two excited energies of an L = 8 TFI chain
    (H + w sum |psi_i><psi_i|, overlap environments, rank-one projector derivative) equal dense ED and the oracle."""
    L = 8
    Hg, Ho = mk.transverse_field_ising(1.0, 1.3, be=cb), mo.tfi_mpo(1.0, 1.3)
    ev = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))
    psi = mk.FiniteMPS.random(L, 2, 16, np.random.default_rng(0), be=cb)
    p0, e0, eps0 = mk.find_groundstate(psi, Hg, mk.DMRG(tol=1e-11, maxiter=30))
    assert abs(float(np.sum(mk.expectation_value(p0, Hg, e0))) - ev[0]) < 1e-10
    ens, sts = mk.excitations(Hg, mk.FiniteExcited(gsalg=mk.DMRG(tol=1e-10, maxiter=30)), p0, num=2)
    assert abs(ens[0] - ev[1]) < 1e-8 and abs(ens[1] - ev[2]) < 1e-8
    ens_o, _ = mo.excitations_finite(Ho, mo.FiniteMPS.random(L, 2, 16, np.random.default_rng(0)) if False else
                                     mo.dmrg(mo.FiniteMPS.random(L, 2, 16, np.random.default_rng(0)), Ho, tol=1e-11, maxiter=30)[0], num=2)
    assert abs(ens[0] - ens_o[0]) < 1e-8 and abs(ens[1] - ens_o[1]) < 1e-8


def _same_up_to_phase(a, b):
    ov = np.vdot(a, b)
    return np.abs(a * (ov / abs(ov)) - b).max()


def test_quasiparticle_infinite_matches_oracle_and_exact_dispersion(cb):
    """quasiparticleexcitation.jl:39-125 + qpenv.jl:55-144 on the product's host code (real / imaginary parts as separate
    real tensors): on the SAME uniform ground state the excitation energies equal the oracle's complex-arithmetic ones,
    the B tensors agree up to the eigenvector phase, and both reproduce the exact TFI dispersion
    2 sqrt(J^2 + g^2 - 2 J g cos p) at momenta 0 (real path), 1.0 (two-part path) and pi (real path, negative phases)."""
    J, g = 1.0, 2.0
    Ho, Hg = mo.tfi_mpo(J, g), mk.transverse_field_ising(J, g, be=cb)
    po, eo, _, _ = mo.vumps(mo.InfiniteMPS.random(2, 6, np.random.default_rng(1)), Ho, tol=1e-11, maxiter=100)
    psi = mk.InfiniteMPS(*[[cb.upload(t) for t in lst] for lst in (po.AL, po.AR, po.CR, po.AC)], cb)
    envs = mk.environments(psi, Hg)
    for p in (0.0, 1.0, np.pi):
        ens, phis = mk.excitations(Hg, mk.QuasiparticleAnsatz(), p, psi, envs, num=2 if p == 1.0 else 1)
        exact = 2 * np.sqrt(J * J + g * g - 2 * J * g * np.cos(p))
        ens_o, phis_o, M = mo.excitations_qp(Ho, mo.LeftGaugedQP.random(np.random.default_rng(0), po, momentum=p), eo, num=2, dense=True)
        assert abs(ens[0] - ens_o[0]) < 1e-8 and abs(ens[0] - exact) < 1e-5, (p, ens, ens_o, exact)
        assert phis[0].nparts == (2 if p == 1.0 else 1)
        assert _same_up_to_phase(phis[0].B_host(0), phis_o[0].B(0)) < 1e-6
        if p == 1.0:
            assert abs(ens[1] - ens_o[1]) < 1e-7


def test_quasiparticle_two_site_cell_and_finite(cb):
    """(a) two-site unit cell (the `repeat(H, 2)` case of test/algorithms.jl:204-211): the same energy as the oracle and
    as the exact dispersion folded into the halved Brillouin zone; (b) FiniteQP at full bond dimension: the tangent space is the
    whole Hilbert space minus the ground state, so the quasiparticle energies are the exact gaps (dense ED)."""
    J, g = 1.0, 1.7
    Ho1, Hg1 = mo.tfi_mpo(J, g), mk.transverse_field_ising(J, g, be=cb)
    Ho = mo.MPOHamiltonian([Ho1[0], Ho1[0]])
    po, eo, _, _ = mo.vumps(mo.InfiniteMPS.random(2, 5, np.random.default_rng(2), n=2), Ho, tol=1e-11, maxiter=200)
    H2 = mk.transverse_field_ising(J, g, be=cb)
    H2.slices = [H2.slices[0], H2.slices[0]]
    H2.period = 2
    psi = mk.InfiniteMPS(*[[cb.upload(t) for t in lst] for lst in (po.AL, po.AR, po.CR, po.AC)], cb)
    for p in (0.7, np.pi):
        ens, _ = mk.excitations(H2, mk.QuasiparticleAnsatz(), p, psi, mk.environments(psi, H2))
        ens_o, _ = mo.excitations_qp(Ho, mo.LeftGaugedQP.random(np.random.default_rng(0), po, momentum=p), eo)
        assert abs(ens[0] - ens_o[0]) < 1e-8, (p, ens, ens_o)
        eps = lambda k: 2 * np.sqrt(J * J + g * g - 2 * J * g * np.cos(k))          # noqa: E731
        assert abs(ens[0] - min(eps(p), eps(p + np.pi))) < 1e-4                      # two-site cell: the band is folded
    L = 6
    ev = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho1, L))
    p0, e0, _ = mk.find_groundstate(mk.FiniteMPS.random(L, 2, 8, np.random.default_rng(0), be=cb), Hg1, mk.DMRG(tol=1e-12, maxiter=30))
    ens, phis = mk.excitations(Hg1, mk.QuasiparticleAnsatz(), p0, e0, num=2)
    assert abs(ens[0] - (ev[1] - ev[0])) < 1e-8 and abs(ens[1] - (ev[2] - ev[0])) < 1e-8, (ens, ev[:3] - ev[0])


def test_periodic_boundary_conditions_host(cb):
    """periodic_boundary_conditions on the product's MPOHamiltonian: the same block table as the oracle's (dense operator
    of the ring, chi > 1 middle level included) and the reference's known answer test/algorithms.jl:512-540 (ring of 10
    sites, D = 10: DMRG energy == exact diagonalization) through the product's DMRG driver."""
    L = 6
    h2 = np.random.default_rng(0).standard_normal((2, 2, 2, 2))
    h2 = h2 + np.transpose(h2, (2, 3, 0, 1))
    Hp = mk.periodic_boundary_conditions(mk.from_twosite(h2, be=cb), L)
    Ho = mo.periodic_boundary_conditions(mo.mpoham_from_twosite(h2), L)
    assert Hp.odim == Ho.odim == 6 and Hp.period == L
    as_oracle = mo.MPOHamiltonian([mo.SparseMPOSlice(Hp.odim, 2, Hp[s].chil, Hp[s].chir, Hp[s].blocks) for s in range(L)])
    assert np.abs(mo.dense_hamiltonian(as_oracle, L) - mo.dense_hamiltonian(Ho, L)).max() < 1e-13
    L = 10
    X, Z, E = np.array([[0., 1], [1, 0]]), np.diag([1., -1]), np.eye(2)
    h2 = -(np.kron(Z, Z) + 0.5 * (np.kron(X, E) + np.kron(E, X))).reshape(2, 2, 2, 2)
    Hp = mk.periodic_boundary_conditions(mk.from_twosite(h2, be=cb), L)
    e0 = np.linalg.eigvalsh(mo.dense_hamiltonian(mo.periodic_boundary_conditions(mo.tfi_twosite_mpo(1.0), L), L))[0]
    psi, envs, eps = mk.find_groundstate(mk.FiniteMPS.random(L, 2, 10, np.random.default_rng(0), be=cb), Hp, mk.DMRG(tol=1e-10, maxiter=30))
    assert abs(float(np.sum(mk.expectation_value(psi, Hp, envs))) - e0) < 1e-5


def test_mpo_arithmetic_and_variance_host(cb):
    """MPOHamiltonian.__mul__ / __add__ / scalar * / repeat and toolbox.variance on the product's host code: same numbers as
    the oracle (random finite state incl. fused chi > 1 levels, random uniform state, truncated finite quasiparticle)."""
    rng = np.random.default_rng(0)
    L = 7
    h2 = rng.standard_normal((2, 2, 2, 2))
    h2 = h2 + np.transpose(h2, (2, 3, 0, 1))
    psi = mo.FiniteMPS.random(L, 2, 5, rng)
    pg = mk.FiniteMPS([psi.AC(i) if i == L - 1 else psi.AL(i) for i in range(L)], be=cb)
    for Hg, Ho in ((mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)), (mk.from_twosite(h2, be=cb), mo.mpoham_from_twosite(h2))):
        assert abs(mk.variance(pg, Hg) - mo.variance_finite(psi, Ho)) < 1e-11
        e1 = float(np.sum(mk.expectation_value(pg, Hg, mk.environments(pg, Hg))))
        H3 = 2.0 * Hg + 0.25
        assert abs(float(np.sum(mk.expectation_value(pg, H3, mk.environments(pg, H3)))) - (2 * e1 + 0.25 * L)) < 1e-11
    Hi, Hio = mk.transverse_field_ising(1.0, 2.0, be=cb), mo.tfi_mpo(1.0, 2.0)
    pr = mo.InfiniteMPS.random(2, 4, rng)
    pgi = mk.InfiniteMPS(*[[cb.upload(t) for t in lst] for lst in (pr.AL, pr.AR, pr.CR, pr.AC)], cb)
    assert abs(mk.variance(pgi, Hi) - mo.variance_infinite(pr, Hio)) < 1e-10
    assert Hi.repeat(2).period == 2
    Ht, Hto, L = mk.transverse_field_ising(1.0, 1.5, be=cb), mo.tfi_mpo(1.0, 1.5), 9
    p0, e0, _ = mk.find_groundstate(mk.FiniteMPS.random(L, 2, 4, np.random.default_rng(1), be=cb), Ht, mk.DMRG(tol=1e-10, maxiter=30))
    ens, phis = mk.excitations(Ht, mk.QuasiparticleAnsatz(), p0, e0)
    var = mk.variance(phis[0], Ht, e0)
    assert 1e-7 < var < 1e-2
    po = mo.FiniteMPS([cb.download(p0.AC(i)) if i == L - 1 else cb.download(p0.AL(i)) for i in range(L)])
    eo = mo.FinEnv(po, Hto)
    _, phis_o = mo.excitations_qp(Hto, mo.LeftGaugedQP.random(np.random.default_rng(0), po, dtype=np.float64), eo)
    assert abs(var - mo.variance_qp_finite(phis_o[0], Hto, eo)) < 1e-9


def test_transfer_spectrum_and_correlation_length_host(cb):
    """transfer_spectrum / marek_gap / correlation_length (toolbox.jl:44-125) on the host Arnoldi: the leading eigenvalues of the
    mixed transfer matrix equal those of the dense D^2 x D^2 matrix built from AL."""
    from mpskit_jl_amd import toolbox
    rng = np.random.default_rng(5)
    po = mo.InfiniteMPS.random(2, 5, rng, n=2)
    psi = mk.InfiniteMPS(*[[cb.upload(t) for t in lst] for lst in (po.AL, po.AR, po.CR, po.AC)], cb)
    T = np.eye(25)
    for a in po.AL:
        T = T @ np.einsum("asb,psq->paqb", a, a.conj()).reshape(25, 25)
    ref = np.linalg.eigvals(T)
    ref = ref[np.argsort(-np.abs(ref))]
    vals = toolbox.transfer_spectrum(psi, num_vals=4, krylovdim=25)
    assert abs(vals[0] - 1) < 1e-10
    assert np.abs(np.sort(np.abs(vals)) - np.sort(np.abs(ref[:4]))).max() < 1e-8
    assert abs(toolbox.correlation_length(psi, num_vals=4, krylovdim=25) + 1 / np.log(np.abs(ref[1]))) < 1e-6


def test_exact_diagonalization_host(cb):
    """exact_diagonalization (ED.jl:4-53): full-bond-dimension FiniteMPS + the middle-site effective Hamiltonian == dense ED,
    for an even and an odd length, the three lowest levels; the returned states carry the energies."""
    from mpskit_jl_amd import toolbox
    for L, Hg, Ho in ((6, mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)), (5, mk.transverse_field_ising(1.0, 0.8, be=cb), mo.tfi_mpo(1.0, 0.8))):
        ev = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))
        vals, states = toolbox.exact_diagonalization(Hg, len=L, num=2)
        assert abs(vals[0] - ev[0]) < 1e-10
        assert min(abs(vals[1] - e) for e in ev[1:4]) < 1e-9
        assert abs(float(np.sum(mk.expectation_value(states[0], Hg, mk.environments(states[0], Hg)))) - ev[0]) < 1e-10


def _tfi_symmetry_broken_pair(be, g, D, seed):
    """the two Z2-related ground states of the ferromagnetic TFI chain (oracle VUMPS + spin flip), as oracle and product states."""
    Ho = mo.tfi_mpo(1.0, g)
    po, eo, eps, _ = mo.vumps(mo.InfiniteMPS.random(2, D, np.random.default_rng(seed)), Ho, tol=1e-11, maxiter=200)
    assert eps < 1e-8
    X = np.array([[0., 1], [1, 0]])
    flip = lambda A: np.einsum("ts,asb->atb", X, A)     # noqa: E731
    po2 = mo.InfiniteMPS([flip(a) for a in po.AL], [flip(a) for a in po.AR], [c.copy() for c in po.CR], [flip(a) for a in po.AC])
    up = lambda st: mk.InfiniteMPS(*[[be.upload(t) for t in lst] for lst in (st.AL, st.AR, st.CR, st.AC)], be)   # noqa: E731
    return Ho, po, eo, po2, mo.MPOHamInfEnv(po2, Ho), up(po), up(po2)


def test_quasiparticle_domain_wall_host(cb):
    """Topologically non-trivial quasiparticles (quasiparticle_state.jl:9-11, qpenv.jl:68,85: no regularisation; energies
    renormalised by the mean of the two ground states): the kink between the two symmetry-broken TFI ground states has the
    exact dispersion 2 sqrt(1 + g^2 - 2 g cos p); product host code == oracle == exact."""
    g = 0.5
    Ho, po, eo, po2, eo2, pl, pr = _tfi_symmetry_broken_pair(cb, g, 8, 4)
    Hg = mk.transverse_field_ising(1.0, g, be=cb)
    for p in (0.0, 0.8):
        ens, phis = mk.excitations(Hg, mk.QuasiparticleAnsatz(), p, pl, mk.environments(pl, Hg), pr, mk.environments(pr, Hg))
        assert not phis[0].trivial
        VLs = [mo.leftnull(a) for a in po.AL]
        phi_o = mo.LeftGaugedQP(po, po2, VLs, [np.random.default_rng(0).random((VLs[0].shape[2], 8)) + 0j], momentum=p)
        ens_o, _ = mo.excitations_qp(Ho, phi_o, eo, eo2)
        exact = 2 * np.sqrt(1 + g * g - 2 * g * np.cos(p))
        assert abs(ens[0] - ens_o[0]) < 1e-8 and abs(ens[0] - exact) < 1e-6, (p, ens, ens_o, exact)


def test_quasiparticle_long_range_hamiltonian(cb):
    """A Hamiltonian with a decaying (non-identity diagonal) MPO level: -sum Z Z - g sum X + sum_{r>=1} c lam^(r-1) X_i X_{i+r}.
    The quasiparticle transfer systems then solve (1 - e^{-+ip} lam T) on that level (exci_transfer_system.jl:29-31,72-74);
    the oracle's effective Hamiltonian stays Hermitian and the product host code gives the same energies."""
    X, Z = np.array([[0., 1], [1, 0]]), np.diag([1., -1])
    g, c, lam = 2.0, 0.3, 0.5
    blocks = {(0, 0): 1.0, (3, 3): 1.0, (0, 1): -Z, (1, 3): Z, (0, 2): c * X, (2, 2): lam, (2, 3): X, (0, 3): -g * X}
    Ho = mo.MPOHamiltonian([mo.mpoham_from_chain(blocks, 2)])
    Hg = mk.MPOHamiltonian(blocks, be=cb)
    po, eo, eps, _ = mo.vumps(mo.InfiniteMPS.random(2, 6, np.random.default_rng(3)), Ho, tol=1e-11, maxiter=200)
    assert eps < 1e-8
    psi = mk.InfiniteMPS(*[[cb.upload(t) for t in lst] for lst in (po.AL, po.AR, po.CR, po.AC)], cb)
    envs = mk.environments(psi, Hg)
    for p in (0.0, 1.3):
        ev, _, M = mo.excitations_qp(Ho, mo.LeftGaugedQP.random(np.random.default_rng(0), po, momentum=p), eo, num=1, dense=True)
        assert np.abs(M - M.conj().T).max() < 1e-9
        ens, _ = mk.excitations(Hg, mk.QuasiparticleAnsatz(), p, psi, envs)
        assert abs(ens[0] - ev[0]) < 1e-8, (p, ens, ev)


def test_two_site_drivers_host_logic(cb):
    """DMRG2 (dmrg.jl:80-137) and TDVP2 (tdvp.jl:100-151) host logic on the stand-in backend: the truncating two-site sweep
    reaches the ED ground-state energy of an L = 8 Heisenberg chain and grows the bond dimension from 2 to the cap; one
    imaginary-time TDVP2 step lowers the energy of the same chain."""
    L = 8
    Hg, Ho = mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)
    ev = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))
    psi = mk.FiniteMPS.random(L, 2, 2, np.random.default_rng(0), be=cb)
    p, e, eps = mk.find_groundstate(psi, Hg, mk.DMRG2(trunc_dim=16, tol=1e-10, maxiter=20))
    assert abs(float(np.sum(mk.expectation_value(p, Hg, e))) - ev[0]) < 1e-9
    assert max(p.AL(i).shape[2] for i in range(L - 1)) == 16
    psi = mk.FiniteMPS.random(L, 2, 4, np.random.default_rng(1), be=cb)
    envs = mk.environments(psi, Hg)
    e_before = float(np.sum(mk.expectation_value(psi, Hg, envs)))
    psi2, envs2 = mk.timestep(psi, Hg, 0.0, -0.1j, mk.TDVP2(trunc_dim=8), envs)
    e_after = float(np.sum(mk.expectation_value(psi2, Hg, envs2)))        # expectation_value divides by <psi|psi>
    assert e_after < e_before - 1e-3


def test_calc_galerkin_value_parity_host(cb):
    """toolbox.jl:17-25 through the product's host code on the stand-in backend (the GPU twin is
    tests/test_gpu_traces.py::test_calc_galerkin_value_parity, same body)."""
    from test_gpu_traces import test_calc_galerkin_value_parity as body
    body(cb)


@pytest.mark.parametrize("L,m", [(21, 4), (39, 4)])
def test_sweep_galerkin_slot_does_not_alias_solver_buffers(cb, L, m):
    """Regression: the sweep's galerkin slot (2L-2, 1) and the fixed-budget solver's scalar / Ritz buffers were
    pooled by shape alone; at L = 21 (RITZ_BUF = 40) and L = 39 (m (2m+1) + 40 = 76 for m = 4) the shapes coincide and
    the solver overwrote the stored galerkin values (max eps 2.0 on this very input before the fix).  Pools are tagged
    now: eps_s of a fixed-budget sweep must equal the values of a sweep whose galerkin slot lives in a workspace of its own."""
    Hg = mk.heisenberg_XXX(0.5, be=cb)
    rng = np.random.default_rng(3)
    dims = mo.FiniteMPS.random(L, 2, 4, np.random.default_rng(0)).bond_dims()
    As = [rng.random((1 if i == 0 else dims[i - 1], 2, dims[i])) for i in range(L)]
    eig = mk.Arnoldi(fixed_matvecs=m, krylovdim=m)
    out = []
    for shared in (True, False):
        psi = mk.FiniteMPS(As, normalize=True, be=cb)
        envs = mk.FinEnv(psi, Hg)
        ws = krylov.KrylovWorkspace(cb)
        if not shared:
            orig = ws.get
            side = krylov.KrylovWorkspace(cb)
            ws.get = lambda shape, n, tag=None: (side.get(shape, n, tag) if tag == "galerkin" else orig(shape, n, tag))
        out.append(np.array(alg.dmrg_sweep(psi, Hg, envs, eig, ws)))
    assert np.all(np.isfinite(out[0])) and out[0].max() < 1.5
    assert np.allclose(out[0], out[1], rtol=0, atol=1e-12)


def test_thick_restart_eigsolve_converges_with_fewer_matvecs(cb):
    """fixedpoint.jl:19-30 -> KrylovKit eigsolve (Krylov-Schur restart): on a spectrum whose low end is clustered the
    product's thick-restarted solver reaches tol = 1e-12 with a fraction of the matvecs of a single-vector restart (the
    oracle's restatement), returns the same eigenpair, and its residual estimate is the true residual norm."""
    rng = np.random.default_rng(0)
    n = 600
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.concatenate([[-1.0, -0.999, -0.998], np.linspace(-0.99, 5, n - 3)])
    M = (Q * ev) @ Q.T

    def mv(x, out):
        return cb._set(out, M @ cb.download(x).ravel())

    x0 = rng.standard_normal(n)
    lam, vec, nmv, res = krylov.eigsolve_sr(cb, mv, cb.upload(x0), tol=1e-12, krylovdim=30, maxiter=100)
    lo, vo, nmo = mo.eigsolve_sr(lambda x: M @ x, x0, tol=1e-12, krylovdim=30, maxiter=100)     # the oracle's twin of it

    def single_vector_restart():             # what rounds 1-2 did: restart from the Ritz vector alone
        v, count = x0 / np.linalg.norm(x0), 0
        for _ in range(200):
            V, H = [v], np.zeros((31, 30))
            for k in range(30):
                w = M @ V[k]
                count += 1
                for _ in range(2):
                    for i in range(k + 1):
                        c = V[i] @ w
                        H[i, k] += c
                        w = w - c * V[i]
                H[k + 1, k] = np.linalg.norm(w)
                e, S = np.linalg.eigh((H[:k + 1, :k + 1] + H[:k + 1, :k + 1].T) / 2)
                if abs(H[k + 1, k] * S[-1, 0]) < 1e-12:
                    return count
                V.append(w / H[k + 1, k])
            v = sum(S[i, 0] * V[i] for i in range(30))
            v /= np.linalg.norm(v)
        return count

    v = cb.download(vec).ravel()
    assert abs(lam + 1.0) < 1e-11 and abs(lo + 1.0) < 1e-11
    assert abs(abs(v @ Q[:, 0]) - 1.0) < 1e-9
    assert np.linalg.norm(M @ v - lam * v) < 5e-12 and res < 1e-12
    assert nmv == nmo                                          # same algorithm, same decisions
    assert nmv > 30 and nmv < 0.5 * single_vector_restart()     # restarted at least once, and far cheaper


def test_complex_states_in_changebonds_and_finite_excited(cb):
    """SURVEY 8(f).1 / 8(f).4 on COMPLEX states (VERDICT r2 missing #5): changebonds OptimalExpand / SvdCut
    (optimalexpand.jl:72-102, svdcut.jl:14-23) and excitations(H, FiniteExcited(), psi) (dmrgexcitation.jl:13-36) on the
    embedded representation (cplx.py) against the oracle's complex arithmetic."""
    from mpskit_jl_amd import cplx
    from mpskit_jl_amd.changebonds import changebonds, OptimalExpand, SvdCut
    rng = np.random.default_rng(29)
    L, D = 6, 3
    dims = mo.FiniteMPS.random(L, 2, D, np.random.default_rng(0)).bond_dims()
    As = [rng.random((1 if i == 0 else dims[i - 1], 2, dims[i])) + 1j * rng.random((1 if i == 0 else dims[i - 1], 2, dims[i]))
          for i in range(L)]
    Hg, Ho = mk.heisenberg_XXX(0.5, be=cb), mo.heisenberg_mpo(0.5)
    pg, po = mk.FiniteMPS(As, normalize=True, be=cb), mo.FiniteMPS(As, normalize=True)
    v0 = mo.mps_to_vector(po)
    # OptimalExpand: bond dimensions grow by k, the state is unchanged, every tensor stays an embedding
    pe, ee = changebonds(pg, Hg, OptimalExpand(trunc_dim=1))
    qe, _ = mo.changebonds_optimalexpand(po, Ho, truncdim=1)
    assert pe.cplx and pe.bond_dims() == qe.bond_dims() and max(pe.bond_dims()) > D
    ve = mo.mps_to_vector(mo.FiniteMPS([pe.download(pe.AL(i)) for i in range(L - 1)] + [pe.download(pe.AC(L - 1))]))
    assert abs(abs(np.vdot(v0, ve)) - 1.0) < 1e-12
    for i in range(L):
        assert cplx.structure_defect(cb.download(pe.AC(i))) < 1e-12
    # the appended RIGHT directions are the oracle's (dominant right singular subspace of NL^dag H_AC2 AC2 NR^dag; checked
    # basis-independently below); the completion of leftorth([AC | 0]) is an arbitrary isometry in the reference as well
    # (whatever QR returns for zero columns), so sweeps from the two expanded states are compared at convergence
    p1, e1, _ = mk.find_groundstate(pe, Hg, mk.DMRG(tol=1e-11, maxiter=30))
    q1, f1, _, logq = mo.dmrg(qe, Ho, tol=1e-11, maxiter=30)
    E1 = float(np.sum(mk.expectation_value(p1, Hg, e1)))
    Eq = float(np.sum(mo.expectation_value(q1, Ho, f1)).real)
    E0 = float(np.sum(mk.expectation_value(pg, Hg, mk.FinEnv(pg, Hg))))
    e_exact = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))[0]
    # (1-site DMRG in the expanded manifold is a local optimisation: the two completions may end in different optima --
    #  here the oracle's run stalls 8e-4 above the exact energy and the product's reaches it; both must be variational
    #  and must improve on the unexpanded state)
    assert e_exact - 1e-9 <= E1 < E0 and e_exact - 1e-9 <= Eq < E0 + 1e-12
    # SvdCut: truncated state == the oracle's truncated state (up to a phase), Schmidt values kept
    pc = changebonds(pe, SvdCut(trunc_dim=2))            # (the expanded states are the SAME state on both sides)
    qc = mo.changebonds_svdcut(qe, truncdim=2)
    assert pc.bond_dims() == qc.bond_dims() and abs(pc.norm() - 1.0) < 1e-12
    vc = mo.mps_to_vector(mo.FiniteMPS([pc.download(pc.AL(i)) for i in range(L - 1)] + [pc.download(pc.AC(L - 1))]))
    assert abs(abs(np.vdot(mo.mps_to_vector(qc), vc)) - 1.0) < 1e-10
    mid = L // 2 - 1
    sg = np.linalg.svd(pc.download(pc.CR(mid)), compute_uv=False)
    assert np.abs(sg - np.linalg.svd(qc.CR(mid), compute_uv=False)).max() < 1e-10
    # FiniteExcited on a complex ground state: first excited energy == the oracle's (and == dense ED)
    Lx = 6
    dx = mo.FiniteMPS.random(Lx, 2, 8, np.random.default_rng(0)).bond_dims()
    Ax = [rng.random((1 if i == 0 else dx[i - 1], 2, dx[i])) + 1j * rng.random((1 if i == 0 else dx[i - 1], 2, dx[i])) for i in range(Lx)]
    g0, ge, _ = mk.find_groundstate(mk.FiniteMPS(Ax, normalize=True, be=cb), Hg, mk.DMRG(tol=1e-11, maxiter=20))
    ens, sts = mk.excitations(Hg, mk.FiniteExcited(gsalg=mk.DMRG(tol=1e-10, maxiter=30), weight=10.0), g0, num=1)
    w = np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, Lx))
    assert abs(ens[0] - w[1]) < 1e-7 and sts[0].cplx
    assert abs(float(np.sum(mk.expectation_value(g0, Hg, ge))) - w[0]) < 1e-9


def test_structured_split_fast_path_host_logic(cb):
    """cplx.split_two_site through Backend.tsplit (the 2 k + 16 leading vectors of the truncation-aware split) instead of the
    full tsvd, on the CPU stand-in: embedded isometries, complex Schmidt values and the optimal truncation, with and without
    a degenerate pair of complex singular values across the cut; the stand-in counts the calls."""
    from mpskit_jl_amd import cplx
    rng = np.random.default_rng(77)
    m, n, k = 44, 40, 10
    for degenerate in (False, True):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        sv = np.logspace(0, -4, n)
        if degenerate:
            sv[k] = sv[k - 1]
        th = (U * sv) @ V.conj().T
        E = cplx.embed(th).reshape(2 * m, 1, 2 * n, 1)
        before = dict(cb.calls)
        al, c, ar, S, disc = cplx.split_two_site(cb, cb.upload(E), trunc_dim=k)
        assert cb.calls.get("tsplit", 0) == before.get("tsplit", 0) + 1          # the fast path ran
        A2, Cm, B2 = cb.download(al).reshape(2 * m, 2 * k), cb.download(c), cb.download(ar).reshape(2 * k, 2 * n)
        assert np.abs(A2.T @ A2 - np.eye(2 * k)).max() < 1e-12 and np.abs(B2 @ B2.T - np.eye(2 * k)).max() < 1e-12
        assert cplx.structure_defect(A2) < 1e-12 and cplx.structure_defect(B2) < 1e-12
        assert np.abs(S - sv[:k]).max() < 1e-12
        rec = cplx.extract(A2 @ Cm @ B2)
        assert abs(np.linalg.norm(th - rec) - np.linalg.norm(sv[k:])) < 1e-11


def test_fixed_budget_sweeps_on_a_converged_complex_state_stay_variational(cb):
    """Embedded complex states: a fixed-budget eigensolve that restarts from a CONVERGED tensor normalises rounding noise to
    O(1); on embedded Krylov vectors that noise populates the anti-structured sector, where the embedded operator has
    eigenvalues below the ground state, and the sweep energy dropped below the exact ground-state energy (GPU, L = 16,
    D = 64: -7.13 against -6.9117).  The solvers now iterate on half-embedded vectors (derivatives._half_space): the energy
    stays at the converged value and above the ED bound."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import algorithms as alg, krylov
    rng = np.random.default_rng(11)
    L, d, D = 8, 2, 8
    dims = mo.FiniteMPS.random(L, d, D, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    H = mk.heisenberg_XXX(0.5, be=cb)
    psi = mk.FiniteMPS(As, normalize=True, be=cb)
    envs = mk.FinEnv(psi, H)
    ws = krylov.KrylovWorkspace(cb)
    for _ in range(3):
        alg.dmrg_sweep(psi, H, envs, mk.Arnoldi(tol=1e-12, krylovdim=20, maxiter=50), ws)
    Ec = float(np.sum(mk.expectation_value(psi, H, envs)))
    E0 = np.linalg.eigvalsh(mo.dense_hamiltonian(mo.heisenberg_mpo(0.5), L))[0]
    assert E0 - 1e-10 <= Ec <= E0 + 1e-3 * abs(E0)
    for _ in range(3):
        alg.dmrg_sweep(psi, H, envs, mk.Arnoldi(fixed_matvecs=6, krylovdim=6), ws)
        E = float(np.sum(mk.expectation_value(psi, H, envs)))
        assert E >= E0 - 1e-9 and abs(E - Ec) < 1e-8, (E, Ec, E0)


def _vec(tensors):
    v = tensors[0]
    for t in tensors[1:]:
        v = np.tensordot(v, t, axes=([-1], [0]))
    return v.reshape(-1)


def test_interleaved_complex_host_logic():
    """mpskit_jl_amd.native_cplx on the complex CPU stand-in: the HOST logic of the interleaved-storage drivers (mixed-canonical
    bookkeeping, environment extension, index permutations of the two-site tensors, the reference-named entry points) against
    the oracle's complex128 DMRG / DMRG2 / TDVP / TDVP2; the GPU twins in tests/test_gpu_complex.py run the same drivers
    through the C ABI."""
    import mpskit_jl_amd as mk
    from mpskit_jl_amd import native_cplx as nc
    from cpu_backend import CpuComplexBackend
    cb = CpuComplexBackend()
    rng = np.random.default_rng(5)
    L, d, D = 6, 2, 8
    dims = mo.FiniteMPS.random(L, d, D, np.random.default_rng(0)).bond_dims()
    As = [rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i])) + 1j * rng.standard_normal((1 if i == 0 else dims[i - 1], d, dims[i]))
          for i in range(L)]
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.array([[1, 0], [0, -1]], dtype=complex)
    H = nc.ComplexMPOHamiltonian({(0, 0): 1.0, (4, 4): 1.0, (0, 1): X, (1, 4): X, (0, 2): Y, (2, 4): Y, (0, 3): Z, (3, 4): Z}, cb)
    Ho = mo.heisenberg_pauli_mpo()
    eig = mk.Arnoldi(tol=1e-12, krylovdim=16, maxiter=40)
    # one-site DMRG, sweep by sweep
    psi = nc.NativeFiniteMPS(As, cb)
    assert psi.bytes() == 16 * sum(int(np.prod(a.shape)) for a in As)
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(As, normalize=True)
    for sweep in range(2):
        E = nc.dmrg_sweep(psi, H, envs, eig)
        po, _, _, log = mo.dmrg(po, Ho, maxiter=1, eig_tol=1e-12, krylovdim=16, eig_maxiter=40)
        assert abs(E - log[-1][1]) < 1e-9 * abs(E)
    # two-site DMRG growing the bond dimension from 2
    dims2 = mo.FiniteMPS.random(L, d, 2, np.random.default_rng(0)).bond_dims()
    Bs = [rng.standard_normal((1 if i == 0 else dims2[i - 1], d, dims2[i])) + 1j * rng.standard_normal((1 if i == 0 else dims2[i - 1], d, dims2[i]))
          for i in range(L)]
    psi = nc.NativeFiniteMPS(Bs, cb)
    envs = nc.NativeFinEnv(psi, H)
    po = mo.FiniteMPS(Bs, normalize=True)
    for sweep in range(2):
        E = nc.dmrg2_sweep(psi, H, envs, eig, trunc_dim=D)
        po, _, _, log = mo.dmrg2(po, Ho, truncdim=D, maxiter=1, eig_tol=1e-12, krylovdim=16, eig_maxiter=40)
        assert abs(E - log[-1][1]) < 1e-9 * abs(E)
    assert abs(E - np.linalg.eigvalsh(mo.dense_hamiltonian(Ho, L))[0]) < 1e-8 * abs(E)
    # real-time TDVP and TDVP2 steps
    psi = nc.NativeFiniteMPS(As, cb)
    psi, envs = nc.timestep(psi, H, 0.0, 0.05, mk.TDVP(tol=1e-12, krylovdim=16))
    po2, _ = mo.tdvp_timestep(mo.FiniteMPS(As, normalize=True), Ho, 0.0, 0.05, tol=1e-12, krylovdim=16)
    vo, vn = mo.mps_to_vector(po2), _vec(psi.to_host())
    assert abs(abs(np.vdot(vo, vn)) - 1.0) < 1e-9 and abs(np.linalg.norm(vn) - 1.0) < 1e-10
    psi = nc.NativeFiniteMPS(Bs, cb)
    psi, envs = nc.timestep(psi, H, 0.0, 0.05, mk.TDVP2(tol=1e-12, krylovdim=16, trunc_dim=D))
    po2, _ = mo.tdvp2_timestep(mo.FiniteMPS(Bs, normalize=True), Ho, 0.0, 0.05, truncdim=D, tol=1e-12, krylovdim=16)
    vo, vn = mo.mps_to_vector(po2), _vec(psi.to_host())
    assert abs(abs(np.vdot(vo, vn)) / (np.linalg.norm(vo) * np.linalg.norm(vn)) - 1.0) < 1e-9
