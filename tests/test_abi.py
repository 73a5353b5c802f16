"""The C-ABI library builds, loads and exports every symbol include/mpsk.h declares; the ctypes
binding covers all of them; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mpsk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpsk_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    syms = _declared_symbols()
    for need in ["mpsk_dAC", "mpsk_dC", "mpsk_dAC2", "mpsk_transfer_left", "mpsk_transfer_right", "mpsk_regularize",
                 "mpsk_qrpos", "mpsk_lqpos", "mpsk_tsvd", "mpsk_gemm", "mpsk_vdot", "mpsk_vnrm2", "mpsk_vaxpby",
                 "mpsk_vgs_step", "mpsk_mposlice_create", "mpsk_ctx_create"]:
        assert need in syms


def test_library_exports_every_declared_symbol():
    from mpskit_jl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_binding_covers_the_header():
    from mpskit_jl_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.EXTRA_SYMBOLS)
    assert set(_declared_symbols()) <= bound, sorted(set(_declared_symbols()) - bound)
    lib = _lib.load()
    assert lib.mpsk_version() >= 100


def test_comm_library_exports_its_header():
    """include/mpsk_comm.h (RCCL collectives behind the C ABI): every declared symbol is exported and bound."""
    from mpskit_jl_amd import _lib
    src = open(os.path.join(ROOT, "include", "mpsk_comm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mpsk_comm_[A-Za-z0-9_]+)\s*\(", src)))
    assert "mpsk_comm_allgather" in syms and "mpsk_comm_allreduce_sum" in syms and "mpsk_comm_hac_apply" in syms
    assert os.path.exists(_lib.COMM_PATH), "run __graft_entry__.build() first"
    lib = _lib.load_comm()
    missing = [s_ for s_ in syms if not hasattr(lib, s_)]
    assert not missing, missing
    assert set(syms) <= set(_lib.COMM_SIGNATURES) | {"mpsk_comm_last_error"}


def test_no_cpu_fallback():
    import torch
    import mpskit_jl_amd as mk
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mk.MpskError):
        mk.Backend(0)
    # the C entry point itself also refuses (no device) instead of computing on the host
    from mpskit_jl_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.mpsk_ctx_create(0, ctypes.byref(h)) != 0
    assert b"hip" in lib.mpsk_last_error().lower() or b"device" in lib.mpsk_last_error().lower()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mpskit.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "mpskit_oracle" not in txt and "import oracle" not in txt, f
