"""CPU: the C-ABI library loads and exports every symbol include/ttl_hip.h declares; calls
that need a device fail loudly instead of falling back."""
import ctypes as C
import os

import pytest

from conftest import ROOT
from ttl_amd import _lib


def test_library_is_built_in_tree():
    for p in _lib.LIB_PATHS.values():
        assert os.path.exists(p), "run `python __graft_entry__.py` (or make -C .../csrc) first"
    with pytest.raises(_lib.TtlError):
        _lib.load("fp8")


@pytest.mark.parametrize("precision", ["bf16", "fp16", "strict"])
def test_exports_match_header(precision):
    lib = _lib.load(precision)
    declared = _lib.header_symbols()
    assert len(declared) >= 20
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ttl_version().startswith(b"ttl_hip")
    assert lib.ttl_operand_dtype() == _lib.OPERAND_DTYPE[precision].encode()


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


def test_struct_layouts_match_the_header_and_the_documented_stub(tmp_path):
    """ttl_config / ttl_episode_args as ctypes sees them (ttl_amd/_lib.py) == as a C compiler lays out include/ttl_hip.h
    (sizeof + every offsetof), and the ctypes stub printed in INTEGRATION.md lists the same ttl_config fields in the same
    order (a field missing there shifts nothing visibly but makes the library read past the caller's struct)."""
    import re
    import shutil
    import subprocess
    import ctypes as C
    from ttl_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ttl_hip.h")).read()

    def c_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [re.search(r"(\w+)\s*$", part.strip()).group(1) for d in body.split(";") if d.strip() for part in d.split(",")]

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    for name, ct in (("ttl_config", _lib.ttl_config), ("ttl_episode_args", _lib.ttl_episode_args), ("ttl_plpd_args", _lib.ttl_plpd_args)):
        fields = c_fields(name)
        assert fields == [f[0] for f in ct._fields_], (name, fields)
        src = tmp_path / (name + ".c")
        prints = "".join('printf("%%zu ", offsetof(%s, %s));' % (name, f) for f in fields)
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ttl_hip.h"\nint main(void){printf("%%zu ", sizeof(%s));%s return 0;}\n'
                       % (name, prints))
        exe = tmp_path / name
        subprocess.check_call([gcc, "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
        nums = [int(v) for v in subprocess.check_output([str(exe)]).split()]
        assert nums[0] == C.sizeof(ct), (name, nums[0], C.sizeof(ct))
        assert nums[1:] == [getattr(ct, f).offset for f in fields], name
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = doc[doc.index("class ttl_config(C.Structure):"):doc.index("cfg = ttl_config(")]
    assert re.findall(r'"(\w+)"', stub) == c_fields("ttl_config")
    n_init = len(re.search(r"cfg = ttl_config\((.*?)\)", doc).group(1).split(","))
    assert n_init == len(c_fields("ttl_config"))



def test_big_gemm_epilogues_issue_the_stores_the_counted_waits_allow_for():
    """gemm_big.hip's first waits of a block's next tile allow for 4*MT unacknowledged epilogue stores per wave
    (wait_tiles1_st); fewer store instructions in some epilogue would let a K-tile be read before it lands.  Checked on the
    ISA hipcc really emits for gfx950 (tools/check_big_epilogue.py)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_big_epilogue.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gemm_big_kernel<5,3,2>: 20 store" in r.stdout


def test_no_load_lands_in_the_result_registers_of_an_mfma_in_flight():
    """tools/scan_mfma_srcc_reuse.py on the ISA of both product builds (round-4 review item 5): no DS read / VMEM load may target
    the vDst block of an MFMA issued <= 4 instructions earlier whose result nothing has consumed yet.  (Loads into a RETIRED SrcC
    block — the pattern round 4 suspected — are what hipcc's allocator emits hundreds of times in the shipped, bitwise-repeatable
    GEMM main loops; the scan counts them and says so.)"""
    import re, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_mfma_srcc_reuse.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "(must be 0): 0" in r.stdout
    for build in ("bf16", "fp16"):
        assert re.search(rf"{build}\s+gemm_big.hip\s+\d{{4}} MFMA instructions", r.stdout), r.stdout[-1500:]    # the scan really saw the kernels


def test_the_three_builds_export_the_same_abi():
    """libttl_hip.so, libttl_hip_fp16.so and the test-only libttl_hip_strict.so: every symbol include/ttl_hip.h declares, and
    each says which operand type it was built for."""
    for prec, dt in _lib.OPERAND_DTYPE.items():
        lib = _lib.load(prec)
        assert lib.ttl_operand_dtype().decode() == dt
        assert not [s for s in _lib.header_symbols() if not hasattr(lib, s)]


def test_degenerate_configs_are_refused_not_crashed():
    """Config validation runs before anything touches a device: an all-zero ttl_config (heads == 0 used to divide by zero —
    found by the ASan host test) and other degenerate fields come back as TTL_EINVAL with a message."""
    lib = _lib.load("bf16")
    h = C.c_void_p()
    cfg = _lib.ttl_config()
    assert lib.ttl_ctx_create(C.byref(cfg), C.byref(h)) != 0 and not h.value
    assert b"positive" in lib.ttl_last_error()
    assert lib.ttl_workspace_bytes(C.byref(cfg)) == 0
    cfg = _lib.ttl_config(224, 0, 768, 12, 3072, 12, 512, 16, 32.0, 9, 11, 1e-5, 64, 200, 0, 0, 0, 0)     # patch_size 0
    assert lib.ttl_ctx_create(C.byref(cfg), C.byref(h)) != 0 and not h.value
    cfg = _lib.ttl_config(8, 16, 768, 12, 3072, 12, 512, 16, 32.0, 9, 11, 1e-5, 64, 200, 0, 0, 0, 0)       # image smaller than a patch
    assert lib.ttl_ctx_create(C.byref(cfg), C.byref(h)) != 0 and not h.value
