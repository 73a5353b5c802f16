"""ctypes binding of libmpsk.so (the C ABI declared in include/mpsk.h).

This is the executable twin of the Julia `ccall` shim described in INTEGRATION.md: every
function below is bound with exactly the argument list a `ccall((:mpsk_xxx, libmpsk), Cint, ...)`
would use.  There is NO CPU fallback: if the shared library is missing or a call fails the
product path raises (`MpskError`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPSK_LIB: developer override for A/B builds of the SAME library (tools/ab_dac.py); never a fallback
LIB_PATH = os.environ.get("MPSK_LIB") or os.path.join(_HERE, "libmpsk.so")


class MpskError(RuntimeError):
    pass


_lib = None

c_void_pp = C.POINTER(C.c_void_p)
c_int32_p = C.POINTER(C.c_int32)
c_double_p = C.POINTER(C.c_double)

# name -> argtypes   (restype is always int except where noted)
SIGNATURES = {
    "mpsk_ctx_create": [C.c_int, c_void_pp],
    "mpsk_ctx_destroy": [C.c_void_p],
    "mpsk_ctx_set_stream": [C.c_void_p, C.c_void_p],
    "mpsk_ctx_synchronize": [C.c_void_p],
    "mpsk_ctx_workspace_reserve": [C.c_void_p, C.c_size_t],
    "mpsk_ctx_force_tile": [C.c_void_p, C.c_int, C.c_int],
    "mpsk_ctx_set_qr_mode": [C.c_void_p, C.c_int],
    "mpsk_ctx_qr_stats": [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)],
    "mpsk_ctx_set_svd_mode": [C.c_void_p, C.c_int],
    "mpsk_ctx_svd_stats": [C.c_void_p, C.POINTER(C.c_int)],
    "mpsk_ctx_split_stats": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)],
    "mpsk_prof_enable": [C.c_void_p, C.c_int],
    "mpsk_prof_summary": [C.c_void_p, C.c_char_p, C.c_size_t],
    "mpsk_malloc": [C.c_void_p, C.c_size_t, c_void_pp],
    "mpsk_free": [C.c_void_p, C.c_void_p],
    "mpsk_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_memcpy_d2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_mposlice_create": [C.c_void_p, C.c_int, C.c_int, c_int32_p, c_int32_p, C.c_int, c_int32_p,
                             c_double_p, c_void_pp, c_void_pp],
    "mpsk_mposlice_destroy": [C.c_void_p],
    "mpsk_mposlice_dims": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mpsk_dAC": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                 C.c_void_p],
    "mpsk_dAC_blocked": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_void_p],
    "mpsk_ctx_set_dtype": [C.c_void_p, C.c_int],
    "mpsk_ctx_get_stream": [C.c_void_p, c_void_pp],
    "mpsk_ctx_get_device": [C.c_void_p, C.POINTER(C.c_int)],
    "mpsk_hac_create": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_void_pp],
    "mpsk_hac_apply": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "mpsk_hac_destroy": [C.c_void_p],
    "mpsk_hac_info": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mpsk_dC": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                C.c_void_p],
    "mpsk_dAC2": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_void_p],
    "mpsk_transfer_left": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_transfer_right": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_regularize": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_qrpos": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "mpsk_qrpos2": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                    C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "mpsk_qrlq_pair": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                       C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "mpsk_lqpos": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int],
    "mpsk_tsvd": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                  C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), c_double_p],
    "mpsk_tsplit": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                    C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), c_double_p],
    "mpsk_gemm": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                  C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64],
    "mpsk_copy2d": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64],
    "mpsk_vdot": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, c_double_p],
    "mpsk_vnrm2": [C.c_void_p, C.c_int64, C.c_void_p, c_double_p],
    "mpsk_vaxpby": [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_double, C.c_void_p],
    "mpsk_vscal": [C.c_void_p, C.c_int64, C.c_double, C.c_void_p],
    "mpsk_vtimes_i": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "mpsk_vcopy": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "mpsk_vzero": [C.c_void_p, C.c_int64, C.c_void_p],
    "mpsk_vmultidot": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_void_p, c_double_p],
    "mpsk_vgs_step": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_void_p, c_double_p],
    "mpsk_vorth_step": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_void_p, c_double_p, c_double_p],
    "mpsk_vorth_step_dev": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_void_p, C.c_void_p],
    "mpsk_vlincomb": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, c_double_p, C.c_void_p],
    "mpsk_cx_embed": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64],
    "mpsk_cx_half": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64],
    "mpsk_vmultilincomb": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_int, c_void_pp, c_double_p],
    "mpsk_vnormalize_dev": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_vnrm2_dev": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "mpsk_ctx_qr_retries": [C.c_void_p, C.POINTER(C.c_long)],
    "mpsk_ctx_qr_defer": [C.c_void_p],
    "mpsk_ctx_side_mark": [C.c_void_p],
    "mpsk_ctx_side_begin": [C.c_void_p],
    "mpsk_ctx_side_end": [C.c_void_p],
    "mpsk_qr_commit": [C.c_void_p, C.POINTER(C.c_int)],
    "mpsk_hac_eigsolve_fixed": [C.c_void_p, C.c_void_p, C.c_int, c_void_pp, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_vritz_dev": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "mpsk_vlincomb_dev": [C.c_void_p, C.c_int64, C.c_int, c_void_pp, C.c_void_p, C.c_void_p],
}
# symbols without the (ctx, ...) -> int shape
EXTRA_SYMBOLS = ["mpsk_version", "mpsk_last_error"]


def load():
    """dlopen libmpsk.so and attach the prototypes.  Raises MpskError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpskError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64.  If
    # libmpsk were loaded first it would bind the system copies and the two ROCr instances would
    # fight over the device ("no ROCm-capable device is detected").  Importing torch first makes
    # libmpsk's DT_NEEDED libamdhip64.so resolve to the already-loaded runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mpsk_version.restype = C.c_int
    lib.mpsk_version.argtypes = []
    lib.mpsk_last_error.restype = C.c_char_p
    lib.mpsk_last_error.argtypes = []
    _lib = lib
    return lib


COMM_PATH = os.path.join(_HERE, "libmpsk_comm.so")
COMM_SIGNATURES = {
    "mpsk_comm_unique_id": [C.c_void_p],
    "mpsk_comm_create": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_void_pp],
    "mpsk_comm_destroy": [C.c_void_p],
    "mpsk_comm_info": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mpsk_comm_allgather": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_comm_allreduce_sum": [C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_comm_reduce_scatter_sum": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpsk_comm_hac_apply": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int],
}
COMM_ID_BYTES = 128
_comm = None


def load_comm():
    """dlopen libmpsk_comm.so (include/mpsk_comm.h: RCCL all-gather / all-reduce on the ctx stream behind a C ABI)."""
    global _comm
    if _comm is not None:
        return _comm
    load()
    if not os.path.exists(COMM_PATH):
        raise MpskError(f"{COMM_PATH} not found: build it with __graft_entry__.build()")
    lib = C.CDLL(COMM_PATH)
    for name, argtypes in COMM_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mpsk_comm_last_error.restype = C.c_char_p
    lib.mpsk_comm_last_error.argtypes = []
    _comm = lib
    return lib


def check_comm(rc, what=""):
    if rc != 0:
        raise MpskError(f"{what} failed (code {rc}): {load_comm().mpsk_comm_last_error().decode('utf-8', 'replace')}")


def check(rc, what=""):
    if rc != 0:
        msg = load().mpsk_last_error().decode("utf-8", "replace")
        raise MpskError(f"{what} failed (code {rc}): {msg}")
