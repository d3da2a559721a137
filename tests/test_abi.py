"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/polus_hip.h declares (no compute call is made without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "polus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(polus_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from polus_amd import build
    build.build(verbose=False)
    from polus_amd import _lib
    return _lib.load()


def test_header_and_binding_agree():
    from polus_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.polus_abi_version() == 1


def test_workspace_queries_are_host_only(lib):
    assert lib.polus_gemm_workspace_bytes(768, 768, 1) == 0
    assert lib.polus_gemm_workspace_bytes(768, 768, 4) == 4 * 768 * 768 * 4
    assert lib.polus_attention_bwd_workspace_bytes(2, 128, 12) == 2 * 128 * 12 * 4
    # 512 block partials (POLUS_LN_BWD_BLOCKS default: every workgroup resident at once) + room for
    # ceil(512 / 128) second-stage group partials, [3H] f32 each
    assert lib.polus_layernorm_bwd_workspace_bytes(16384, 768) == (512 + 4) * 3 * 768 * 4
    assert lib.polus_crf_workspace_bytes(4, 16, 3) >= (4 * 16 * 3 + 4 + 4 * 9) * 4
    assert lib.polus_sqnorm_workspace_bytes(10 ** 8) == 1024 * 4


def test_argument_validation_happens_on_the_host(lib):
    # bad shapes are rejected before any launch, with a message
    rc = lib.polus_gemm(0, 0, 0, 0, None, 8, None, 8, None, 8, 8, 8, 8, 1.0, None, None, 0, None, 0, 0, 0, 1, None, 0, None)
    assert rc != 0 and b"null operand" in lib.polus_last_error()
    rc = lib.polus_attention_fwd(0, ctypes.c_void_p(16), None, ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 8, 2, 32, 0.0, 0, None)
    assert rc != 0 and b"head_dim" in lib.polus_last_error()
    rc = lib.polus_gemm_dropout(1, 0, 0, 1, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8, 8, 8, 8, 1.0,
                                None, None, 0, None, 0, 0, 0, 1, None, 0, 1.5, 7, None)
    assert rc != 0 and b"0 <= p < 1" in lib.polus_last_error()


def test_no_cpu_fallback():
    import torch
    from polus_amd import ops
    from polus_amd._lib import PolusHipError
    a = torch.zeros(8, 8)
    with pytest.raises(PolusHipError):
        ops.gemm(a, a, torch.zeros(8, 8))
    if not torch.cuda.is_available():
        from polus_amd.tensor import device
        with pytest.raises(RuntimeError):
            device()


def test_product_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "polus_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
