"""mpskit.jl_amd -- MI355X-native (gfx950) DMRG / VUMPS hot path with the API of MPSKit.jl.

Import as ``import mpskit_jl_amd`` (the directory name contains a dot; the top-level module
``mpskit_jl_amd.py`` registers this package under that name).
"""
from ._lib import MpskError, LIB_PATH  # noqa: F401
from .backend import Backend, DTensor, DeviceMPOSlice, default_backend  # noqa: F401
from .operators import (MPOHamiltonian, LazySum, heisenberg_XXX, transverse_field_ising, hubbard, from_twosite,  # noqa: F401,E402
                        periodic_boundary_conditions)
from .states import FiniteMPS, InfiniteMPS  # noqa: F401,E402
from .environments import FinEnv, MPOHamInfEnv, MultipleEnvironments, environments  # noqa: F401,E402
from .derivatives import ddAC, ddAC2, ddC, MPO_ddAC, MPO_ddAC2, MPO_ddC  # noqa: F401,E402
from .algorithms import (DMRG, DMRG2, VUMPS, IDMRG1, IDMRG2, TDVP, TDVP2, Arnoldi, find_groundstate, calc_galerkin,  # noqa: F401,E402
                         expectation_value, timestep, time_evolve)
from .changebonds import changebonds, OptimalExpand, SvdCut  # noqa: F401,E402
from .excitations import excitations, FiniteExcited, ProjectionOperator  # noqa: F401,E402
from .quasiparticle import QuasiparticleAnsatz, LeftGaugedQP  # noqa: F401,E402
from .toolbox import (variance, entropy, entanglement_spectrum, transfer_spectrum, marek_gap, correlation_length,  # noqa: F401,E402
                      exact_diagonalization)
from . import native_cplx  # noqa: F401,E402   (complex128 states on interleaved storage: NativeFiniteMPS, dmrg, tdvp_step)
