"""CPU: the C-ABI library loads and exports every symbol include/ttl_hip.h declares; calls
that need a device fail loudly instead of falling back."""
import ctypes as C
import os

import pytest

from ttl_amd import _lib


def test_library_is_built_in_tree():
    for p in _lib.LIB_PATHS.values():
        assert os.path.exists(p), "run `python __graft_entry__.py` (or make -C .../csrc) first"
    with pytest.raises(_lib.TtlError):
        _lib.load("fp8")


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_exports_match_header(precision):
    lib = _lib.load(precision)
    declared = _lib.header_symbols()
    assert len(declared) >= 20
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ttl_version().startswith(b"ttl_hip")
    assert lib.ttl_operand_dtype() == precision.encode()


def test_config_validation_without_gpu():
    lib = _lib.load()
    bad = _lib.ttl_config(224, 16, 700, 12, 3072, 12, 512, 16, 32.0, 9, 11, 1e-5, 64, 1000)   # width not multiple of 128
    assert lib.ttl_workspace_bytes(C.byref(bad)) == 0
    assert b"width" in lib.ttl_last_error()
    ok = _lib.ttl_config(224, 16, 768, 12, 3072, 12, 512, 16, 32.0, 9, 11, 1e-5, 64, 1000)
    assert lib.ttl_workspace_bytes(C.byref(ok)) > 500e6
    # --layer_range need not end at the top layer: layers 6..11 then keep their activations for the backward
    mid = _lib.ttl_config(224, 16, 768, 12, 3072, 12, 512, 16, 32.0, 3, 5, 1e-5, 64, 1000)
    assert lib.ttl_workspace_bytes(C.byref(mid)) > lib.ttl_workspace_bytes(C.byref(ok))
    inv = _lib.ttl_config(224, 16, 768, 12, 3072, 12, 512, 16, 32.0, 7, 5, 1e-5, 64, 1000)    # lo > hi
    assert lib.ttl_workspace_bytes(C.byref(inv)) == 0 and b"layer range" in lib.ttl_last_error()
    inv2 = _lib.ttl_config(224, 16, 768, 12, 3072, 12, 512, 16, 32.0, 9, 12, 1e-5, 64, 1000)   # hi beyond the tower
    assert lib.ttl_workspace_bytes(C.byref(inv2)) == 0


def test_no_cpu_fallback():
    import torch
    from ttl_amd.config import get_config
    from ttl_amd.engine import TTLEngine
    with pytest.raises(_lib.TtlError):
        TTLEngine(get_config("tiny"), 4, 10, "cpu")
    if not torch.cuda.is_available():
        h = C.c_void_p()
        ok = _lib.ttl_config(64, 16, 128, 2, 512, 4, 64, 16, 32.0, 1, 3, 1e-5, 4, 10)
        rc = _lib.load().ttl_ctx_create(C.byref(ok), C.byref(h))
        assert rc != 0 and not h.value          # no device -> error code, no context


def test_product_path_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under ttl_amd/ may import it."""
    pkg = os.path.dirname(_lib.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
