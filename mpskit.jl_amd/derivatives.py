"""Effective-Hamiltonian ("derivative") operators, src/algorithms/derivatives.jl:6-71.
Same names and call convention as the reference: `h = ddAC(pos, psi, H, envs); y = h(x)` /
`h * x`; every application is ONE call into libmpsk (mpsk_dAC / mpsk_dC / mpsk_dAC2)."""
from __future__ import annotations

from .backend import DTensor


class MPO_ddC:  # MPO_∂∂C  derivatives.jl:6-9
    def __init__(self, be, leftenv, rightenv):
        self.be, self.leftenv, self.rightenv = be, leftenv, rightenv

    def __call__(self, x: DTensor, out: DTensor = None):
        return self.be.dC(self.leftenv, self.rightenv, x, out=out)

    __mul__ = __call__


class MPO_ddAC:  # MPO_∂∂AC  derivatives.jl:11-15
    """Built once per site visit, applied by every Krylov step: the first application prepares the device-side
    operator (mpsk_hac_create: MPO tensor folded into the right environment where that saves the slab mix), every
    application is then mpsk_hac_apply."""

    def __init__(self, be, o, leftenv, rightenv):
        self.be, self.o, self.leftenv, self.rightenv = be, o, leftenv, rightenv
        self._hac = None

    def __call__(self, x: DTensor, out: DTensor = None):
        if not hasattr(self.be, "hac_create"):                   # host stand-in backend of the CPU tests
            return self.be.dAC(self.o, self.leftenv, self.rightenv, x, out=out)
        if self._hac is None:
            self._hac = self.be.hac_create(self.o, self.leftenv, self.rightenv)
        return self._hac.apply(x, out=out)

    __mul__ = __call__

    def eigsolve_fixed(self, x0: DTensor, m: int, vecs, scal: DTensor, out: DTensor, first_image: DTensor = None):
        """Fixed-budget :SR solve in one library call (mpsk_hac_eigsolve_fixed); None if this operator cannot take it
        (host stand-in backend, complex operator)."""
        if not hasattr(self.be, "hac_create"):
            return None
        if self._hac is None:
            self._hac = self.be.hac_create(self.o, self.leftenv, self.rightenv)
        if self._hac.cplx or self._hac.Dlo != self._hac.Dl:
            return None
        return self._hac.eigsolve_fixed(x0, m, vecs, scal, out, first_image)


class MPO_ddAC2:  # MPO_∂∂AC2  derivatives.jl:17-22
    def __init__(self, be, o1, o2, leftenv, rightenv):
        self.be, self.o1, self.o2, self.leftenv, self.rightenv = be, o1, o2, leftenv, rightenv

    def __call__(self, x: DTensor, out: DTensor = None):
        return self.be.dAC2(self.o1, self.o2, self.leftenv, self.rightenv, x, out=out)

    __mul__ = __call__


class LazyDerivativeSum:  # derivatives.jl:310-323 : (h::LazySum{<:DerivativeOperator})(x) = sum(f_k * h_k(x))
    def __init__(self, be, terms, fs):
        self.be, self.terms, self.fs = be, list(terms), list(fs)

    def __call__(self, x: DTensor, out: DTensor = None):
        y = self.terms[0](x, out=out)
        if self.fs[0] != 1.0:
            self.be.scal(self.fs[0], y)
        for f, h in zip(self.fs[1:], self.terms[1:]):
            self.be.axpby(f, h(x), 1.0, y)
        return y

    __mul__ = __call__


def _lazy(fn, pos, psi, H, envs):
    op = LazyDerivativeSum(psi.be, [fn(pos, psi, h, e) for h, e in zip(H, envs.envs)], H.fs)
    if getattr(psi, "cplx", False) and any(hasattr(e, "vector") for e in envs.envs):
        # a projector term exists in the embedded sector only: iterate on half-embedded vectors (cplx.HalfSpaceOp)
        from .cplx import HalfSpaceOp
        kind = {"ddAC": "AC", "ddC": "C", "ddAC2": "AC2"}[fn.__name__]
        t = psi.AC(pos) if kind != "C" else psi.CR(pos)
        Dr2 = psi.AR(pos + 1).shape[2] if kind == "AC2" else t.shape[-1]
        return HalfSpaceOp(psi.be, op, kind, Dr2 // 2)
    return op


def _is_lazy(H, envs):
    return hasattr(envs, "envs") and hasattr(H, "fs")


NATIVE_CPLX_MIN_D = 384     # embedded bond dimension 2 D below which the embedded matvec wins (launch-bound regime:
                            # profiles/r02_bench_cplx.log -- native 1.81x at D = 1024, 1.57x at 512, 0.76x at 256)


def _native_cplx(psi, *envs_):
    """complex (embedded) state on a backend with the MPSK_C128 kernels: apply through cplx.HalfEmbeddedOp when the
    tensors are large enough for the halved flop count to matter."""
    import os
    if not (getattr(psi, "cplx", False) and hasattr(psi.be, "hac_create")):
        return False
    mode = os.environ.get("MPSK_NATIVE_CPLX", "auto")          # "0": always embedded, "1": always native (A/B switch)
    if mode in ("0", "1"):
        return mode == "1"
    return all(e.shape[1] // 2 >= NATIVE_CPLX_MIN_D for e in envs_)


def _half_space(psi, op, kind, GR):
    """Embedded complex state below the native-kernel threshold: the operator still runs on embedded tensors, but the Krylov
    solvers iterate on HALF-embedded (= interleaved complex) vectors (cplx.HalfSpaceOp: encode . op . decode).  On embedded
    vectors the real space is twice the complex one -- besides the embeddings it holds the anti-structured tensors X_E sigma_z,
    on which an embedded operator acts as its partial complex conjugate, with eigenvalues that can lie BELOW the ground state.
    A solve that restarts from a converged tensor (fixed-budget sweeps: the first residual is rounding noise, normalised to
    O(1)) drifts into that sector: measured on a converged L = 16, D = 64 Heisenberg chain, sweep energies of -7.13 against a
    ground-state energy of -6.9117 (tools/native_vs_embedded.py).  On half vectors the sector does not exist."""
    if not getattr(psi, "cplx", False):
        return op
    from .cplx import HalfSpaceOp
    return HalfSpaceOp(psi.be, op, kind, GR.shape[1] // 2)


def ddC(pos, psi, H, envs):  # ∂∂C  derivatives.jl:34-36
    if _is_lazy(H, envs):
        return _lazy(ddC, pos, psi, H, envs)
    if _native_cplx(psi, envs.leftenv(pos + 1, psi), envs.rightenv(pos, psi)):
        from .cplx import HalfEmbeddedOp
        return HalfEmbeddedOp(psi.be, "C", [], envs.leftenv(pos + 1, psi), envs.rightenv(pos, psi))
    GR = envs.rightenv(pos, psi)
    return _half_space(psi, MPO_ddC(psi.be, envs.leftenv(pos + 1, psi), GR), "C", GR)


def ddAC(pos, psi, H, envs):  # ∂∂AC  derivatives.jl:44-46
    if _is_lazy(H, envs):
        return _lazy(ddAC, pos, psi, H, envs)
    if hasattr(envs, "vector"):          # ProjectionOperator term (excitations.py): rank-one |v><v|
        from .excitations import Proj_ddAC
        return Proj_ddAC(psi.be, envs.vector(pos, psi), bool(getattr(psi, "cplx", False)))
    opp = envs.opp[pos] if hasattr(envs, "opp") else H[pos]
    if _native_cplx(psi, envs.leftenv(pos, psi), envs.rightenv(pos, psi)):
        from .cplx import HalfEmbeddedOp
        return HalfEmbeddedOp(psi.be, "AC", [opp], envs.leftenv(pos, psi), envs.rightenv(pos, psi))
    GR = envs.rightenv(pos, psi)
    return _half_space(psi, MPO_ddAC(psi.be, opp, envs.leftenv(pos, psi), GR), "AC", GR)


def ddAC2(pos, psi, H, envs):  # ∂∂AC2  derivatives.jl:55-58
    if _is_lazy(H, envs):
        return _lazy(ddAC2, pos, psi, H, envs)
    o1 = envs.opp[pos] if hasattr(envs, "opp") else H[pos]
    o2 = envs.opp[pos + 1] if hasattr(envs, "opp") else H[pos + 1]
    if _native_cplx(psi, envs.leftenv(pos, psi), envs.rightenv(pos + 1, psi)):
        from .cplx import HalfEmbeddedOp
        return HalfEmbeddedOp(psi.be, "AC2", [o1, o2], envs.leftenv(pos, psi), envs.rightenv(pos + 1, psi))
    GR = envs.rightenv(pos + 1, psi)
    return _half_space(psi, MPO_ddAC2(psi.be, o1, o2, envs.leftenv(pos, psi), GR), "AC2", GR)
