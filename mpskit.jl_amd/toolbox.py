"""variance(state, H, envs)  (src/algorithms/toolbox.jl:128-172): energy variance of a FiniteMPS / InfiniteMPS / finite
quasiparticle state.  H * H is the odim^2-level product MPOHamiltonian (operators.MPOHamiltonian.__mul__), its expectation
value runs through the same environment / matvec kernels with W = odim^2 MPO levels."""
from __future__ import annotations

import numpy as np

from .algorithms import expectation_value
from .environments import environments
from .states import FiniteMPS


def variance(state, H, envs=None):
    from .quasiparticle import LeftGaugedQP, QuasiparticleAnsatz, _QPContext
    if isinstance(state, LeftGaugedQP):
        if not state.finite:
            raise NotImplementedError("variance of an infinite quasiparticle state")
        # toolbox.jl:153-155 converts the state to a FiniteMPS of twice the bond dimension; the same number from the
        # tangent-space machinery: with H' = H - E0 / L,  <phi|H'^2|phi> = <X|H_eff[H' * H'] X> + <gs|H'^2|gs>,
        # <phi|H'|phi> = <X|H_eff[H'] X>  for <X|X> = 1
        be, gs = state.be, state.left_gs
        envs = environments(gs, H) if envs is None else envs
        L = len(gs)
        Hr = H - float(np.sum(expectation_value(gs, H, envs))) / L
        x = be.copy(state.vec)
        be.scal(1.0 / be.norm(x), x)
        vals = []
        for Hx in (Hr, Hr * Hr):
            ctx = _QPContext(Hx, state, environments(gs, Hx), QuasiparticleAnsatz())
            y = ctx.heff(state, x, be.zeros(x.shape))
            vals.append(be.dot(x, y) + ctx.E[0])
        return float(vals[1] - vals[0] ** 2)
    envs = environments(state, H) if envs is None else envs
    if isinstance(state, FiniteMPS):                                              # :140-144
        H2 = H * H
        return float(np.sum(expectation_value(state, H2, environments(state, H2))) - np.sum(expectation_value(state, H, envs)) ** 2)
    Hr = H - expectation_value(state, H, envs)                                    # :135-138
    H2 = Hr * Hr
    return float(np.sum(expectation_value(state, H2, environments(state, H2))))
