"""excitations(H, FiniteExcited(), psi)  (src/algorithms/excitation/dmrgexcitation.jl:13-36): excited states of a finite
chain as ground states of  H + weight * sum_i |psi_i><psi_i|.  The projection operators are LazySum terms whose
environments are plain overlap transfers <psi_i| . |psi> (mpsk_transfer_left/right without an MPO) and whose effective
"Hamiltonian" at a site is the rank-one operator |v><v| with v = <psi_i| contracted into the current mixed-gauge
basis (two GEMMs) -- operators/projection.jl in the reference."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .backend import DTensor
from .algorithms import DMRG, find_groundstate, expectation_value
from .environments import FinEnv, MultipleEnvironments
from .operators import LazySum
from .states import FiniteMPS


@dataclass
class FiniteExcited:  # dmrgexcitation.jl:10-13
    gsalg: object = field(default_factory=DMRG)
    weight: float = 10.0


class ProjectionOperator:
    """|psi><psi| for a fixed FiniteMPS (operators/projection.jl)."""

    def __init__(self, psi: FiniteMPS):
        self.psi = psi


class OverlapEnv:
    """environments(psi, ProjectionOperator(target)): left / right overlap environments [1][bra = target, ket = psi]
    with the identity-based invalidation of FinEnv."""

    def __init__(self, psi, op: ProjectionOperator):
        be = self.be = psi.be
        L = len(psi)
        self.H, self.t = op, op.psi
        self.cplx = bool(getattr(psi, "cplx", False))
        # complex (embedded) states: the boundary "1" is the 2 x 2 identity, every overlap is the embedding of the complex one
        one = np.eye(2).reshape(1, 2, 2) if self.cplx else np.ones((1, 1, 1))
        self.lefts = [be.upload(one)] + [None] * L
        self.rights = [None] * L + [be.upload(one)]
        self.ldeps, self.rdeps = [None] * L, [None] * L

    def leftenv(self, ind, psi):
        a = next((i for i in range(ind) if psi.AL(i) is not self.ldeps[i]), None)
        if a is not None:
            for j in range(a, ind):
                al = psi.AL(j)
                self.lefts[j + 1] = self.be.transfer_left(None, self.lefts[j], al, self.t.AL(j))
                self.ldeps[j] = al
        return self.lefts[ind]

    def rightenv(self, ind, psi):
        L = len(psi)
        a = next((i for i in range(L - 1, ind, -1) if psi.AR(i) is not self.rdeps[i]), None)
        if a is not None:
            for j in range(a, ind, -1):
                ar = psi.AR(j)
                self.rights[j] = self.be.transfer_right(None, self.rights[j + 1], ar, self.t.AR(j))
                self.rdeps[j] = ar
        return self.rights[ind + 1]

    def vector(self, pos, psi):
        """|target> in the current mixed-gauge basis at site pos:  v[a, s, b] = GL[a', a] ACt[a', s, b'] GR[b, b']."""
        be = self.be
        GL, GR = self.leftenv(pos, psi), self.rightenv(pos, psi)        # (1, Dt, D), (1, D', Dt')
        act = self.t.AC(pos)
        Dt, d, Dtr = act.shape
        D, Dp = GL.shape[2], GR.shape[1]
        t1 = be.gemm(DTensor(GL.buf, (Dt, D)), act.reshape(Dt, d * Dtr), transA=True)          # (D, d Dt')
        v = be.gemm(t1.reshape(D * d, Dtr), DTensor(GR.buf, (Dp, Dtr)), transB=True)           # (D d, D')
        return v.reshape(D, d, Dp)


class Proj_ddAC:
    """effective operator of |target><target| at a site: y = v <v, x>.
    Embedded complex tensors (cplx.py): <v, x> = zr + i zi with  2 zr = v_E . x_E  and  2 zi = (i v)_E . x_E  (real
    Frobenius products of the embeddings), and  y_E = zr v_E + zi (i v)_E  -- the projector is complex-linear, i.e. it
    projects on span{v, i v} of the real embedded space."""

    def __init__(self, be, v: DTensor, cplx=False):
        self.be, self.v, self.cplx = be, v, cplx
        self.iv = be.times_i(v) if cplx else None

    def overlap(self, x: DTensor):
        if not self.cplx:
            return self.be.dot(self.v, x), 0.0
        return 0.5 * self.be.dot(self.v, x), 0.5 * self.be.dot(self.iv, x)

    def __call__(self, x: DTensor, out: DTensor = None):
        y = self.be.empty(*x.shape) if out is None else out
        zr, zi = self.overlap(x)
        self.be.axpby(zr, self.v, 0.0, y)
        if self.cplx:
            self.be.axpby(zi, self.iv, 1.0, y)
        return y

    __mul__ = __call__


def _expval_projection(psi, op, envs):
    """<psi| target><target |psi> / <psi|psi>, reported on site 0 (the other sites carry 0)."""
    pos = 0
    cx = bool(getattr(psi, "cplx", False))
    ac = psi.AC(pos)
    be = psi.be
    zr, zi = Proj_ddAC(be, envs.vector(pos, psi), cx).overlap(ac)
    out = np.zeros(len(psi))
    out[0] = (zr * zr + zi * zi) / (be.norm(ac) ** 2 / (2.0 if cx else 1.0))
    return out


def excitations(H, alg, *args, **kw):
    """excitations(H, alg, args...; num)  (excitation/excitations.jl) -> (energies, states), dispatching on `alg` and the
    argument types the way the reference's methods do:
      FiniteExcited:        excitations(H, alg, psi0::FiniteMPS; num, init)                     dmrgexcitation.jl:13-36
      QuasiparticleAnsatz:  excitations(H, alg, momentum | momenta, psi::InfiniteMPS[, envs]; num)   quasiparticleexcitation.jl:84-125
                            excitations(H, alg, psi::FiniteMPS[, envs]; num)                    :163-169
                            excitations(H, alg, phi0::LeftGaugedQP[, envs]; num)                :39-53,127-143"""
    from .quasiparticle import QuasiparticleAnsatz, LeftGaugedQP, excitations_qp, excitations_momenta
    if isinstance(alg, FiniteExcited):
        return _excitations_finite_excited(H, alg, *args, **kw)
    if not isinstance(alg, QuasiparticleAnsatz):
        raise TypeError(f"excitations: unknown algorithm {type(alg).__name__}")
    if isinstance(args[0], LeftGaugedQP):
        return excitations_qp(H, alg, *args, **kw)
    if isinstance(args[0], FiniteMPS):
        psi, rest = args[0], args[1:]
        if getattr(psi, "cplx", False):
            raise NotImplementedError("QuasiparticleAnsatz on complex (embedded) ground states")
        rng = kw.pop("rng", None)
        rpsi = rest[1] if len(rest) > 1 else None            # (lmps, lenvs, rmps, renvs) as in the reference
        rest = rest[:1] + rest[2:]
        return excitations_qp(H, alg, LeftGaugedQP.random(psi, 0.0, rng, rpsi), *rest, **kw)
    if getattr(args[1], "cplx", False):
        raise NotImplementedError("QuasiparticleAnsatz on complex (embedded) ground states")
    return excitations_momenta(H, alg, *args, **kw)


def _excitations_finite_excited(H, alg: FiniteExcited, psi0: FiniteMPS, num=1, init=None):
    """excitations(H, FiniteExcited(gsalg, weight), psi0; num) -> (energies, states)."""
    be = psi0.be
    cx = bool(getattr(psi0, "cplx", False))
    L = len(psi0)
    states, ens, out = [psi0], [], []
    for _ in range(num):
        start = FiniteMPS([be.copy(psi0.AC(i)) for i in range(L)], normalize=True, be=be, cplx=cx) if init is None else init.copy()
        ops = [H] + [ProjectionOperator(s) for s in states]
        Hs = LazySum(ops, [1.0] + [alg.weight] * len(states))
        envs = MultipleEnvironments(Hs, [FinEnv(start, H)] + [OverlapEnv(start, o) for o in ops[1:]])
        ne, _, _ = find_groundstate(start, Hs, alg.gsalg, envs)
        states.append(ne)
        out.append(ne)
        ens.append(float(np.sum(expectation_value(ne, H, FinEnv(ne, H)))))
    return ens, out
