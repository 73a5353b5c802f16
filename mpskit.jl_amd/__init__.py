"""mpskit.jl_amd -- MI355X-native (gfx950) DMRG / VUMPS hot path with the API of MPSKit.jl.

Import as ``import mpskit_jl_amd`` (the directory name contains a dot; the top-level module
``mpskit_jl_amd.py`` registers this package under that name).
"""
from ._lib import MpskError, LIB_PATH  # noqa: F401
from .backend import Backend, DTensor, DeviceMPOSlice, default_backend  # noqa: F401
