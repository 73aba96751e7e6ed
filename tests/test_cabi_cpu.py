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


@pytest.mark.parametrize("precision", ["bf16", "fp16", "strict", "experiments"])
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


def test_the_builds_export_the_same_abi():
    """libttl_hip_fp16.so (the surface's default), libttl_hip.so, and the test-only libttl_hip_strict.so / libttl_hip_fp16_exp.so:
    every symbol include/ttl_hip.h declares, and each says which operand type it was built for."""
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


def test_the_surface_defaults_to_the_fp16_build():
    """No precision named -> the fp16-operand library (the reference's autocast dtype, inside the 1e-3 logit tolerance); bf16 and
    strict are opt-in; no default of the package names bf16 any more."""
    import inspect
    import re
    import subprocess
    import sys
    from ttl_amd import custom_clip, driver, engine, views
    assert _lib.DEFAULT_PRECISION == os.environ.get("TTL_PRECISION", "fp16")
    if not os.environ.get("TTL_PRECISION"):
        assert _lib.load() is _lib.load("fp16") and _lib.load().ttl_operand_dtype() == b"fp16"
    for fn in (custom_clip.ClipTestTimeTuning.__init__, custom_clip.build_text_mode_engine, engine.TTLEngine.__init__,
               engine.TextTowerEngine.__init__, driver.EpisodePipeline.__init__, views.make_views, views.GpuAugMixAugmenter.__init__,
               _lib.load):
        assert inspect.signature(fn).parameters["precision"].default is None, fn
    pkg = os.path.dirname(_lib.__file__)
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert not re.search(r'precision(: str)? *= *"bf16"|default="bf16"', open(os.path.join(pkg, f)).read()), f
    # the process-wide override, and a bad value is refused at import
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(pkg), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", "from ttl_amd import _lib; print(_lib.DEFAULT_PRECISION, _lib.load().ttl_operand_dtype().decode())"],
                       env=dict(env, TTL_PRECISION="bf16"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["bf16", "bf16"], r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", "from ttl_amd import _lib"], env=dict(env, TTL_PRECISION="fp8"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "TTL_PRECISION" in r.stderr


def test_product_builds_read_five_environment_variables_and_say_which():
    """The kernel path of a product build is frozen: the only TTL_* strings in the binaries are the five run-time switches
    ttl_runtime_switches() lists (csrc/common.hpp TtlSwitch); the closed A/B knobs exist in the -DTTL_EXPERIMENTS build only."""
    import re
    import subprocess
    want = {"TTL_GEMM_HUGE": 2, "TTL_GEMM_HUGE_NARROW": -1, "TTL_GEMM_HUGE_MIN_FILL": 85, "TTL_BWD_COMPACT": 1, "TTL_CONCURRENCY": 0}

    def ttl_strings(path):
        out = subprocess.run(["strings", path], capture_output=True, text=True, check=True).stdout
        return sorted(set(l for l in out.splitlines() if re.fullmatch(r"TTL_[A-Z0-9_]+", l)))
    for prec in ("fp16", "bf16", "strict"):
        got = ttl_strings(_lib.LIB_PATHS[prec])
        assert got == sorted(want) and len(got) <= 6, (prec, got)
        sw = _lib.runtime_switches(prec)
        assert {k: v[1] for k, v in sw.items()} == want, sw
        assert all(v[0] == int(os.environ.get(k, v[1])) for k, v in sw.items())
        assert b"EXPERIMENTS" not in _lib.load(prec).ttl_version()
    exp = ttl_strings(_lib.LIB_PATHS["experiments"])
    assert set(want) < set(exp) and {"TTL_QKV_HEAD_MAJOR", "TTL_POOLED_LAST_LAYER", "TTL_GEMM_HUGE_DGRAD", "TTL_ATTN_VARIANT"} <= set(exp), exp
    assert b"EXPERIMENTS" in _lib.load("experiments").ttl_version()


def test_workspace_bytes_is_the_allocation_walk():
    """ttl_workspace_bytes walks ttl_ctx_create's own allocation list (nothing allocated, no GPU needed): it includes the
    packed-backward buffers of top-k selections (round-5 advisor: the old closed formula had not followed them), grows with every
    capacity, and follows TTL_BWD_COMPACT like the context does.  tests/test_gpu_path.py compares it with a live context."""
    import subprocess
    import sys
    lib = _lib.load()
    mk = lambda **o: _lib.ttl_config(*[o.get(k, d) for k, d in (("image_size", 224), ("patch_size", 16), ("width", 768), ("heads", 12),
                                     ("mlp", 3072), ("layers", 12), ("embed", 512), ("rank", 16), ("lora_alpha", 32.0), ("layer_lo", 9),
                                     ("layer_hi", 11), ("ln_eps", 1e-5), ("max_views", 64), ("max_classes", 1000))])
    base = lib.ttl_workspace_bytes(C.byref(mk()))
    assert 1.5e9 < base < 4e9
    assert lib.ttl_workspace_bytes(C.byref(mk(max_views=128))) > 1.6 * base - 4e8
    assert lib.ttl_workspace_bytes(C.byref(mk(rank=32))) > base
    assert lib.ttl_workspace_bytes(C.byref(mk(max_classes=2000))) > base
    code = ("import ctypes as C; from ttl_amd import _lib; "
            "c = _lib.ttl_config(224, 16, 768, 12, 3072, 12, 512, 16, 32.0, 9, 11, 1e-5, 64, 1000); print(_lib.load().ttl_workspace_bytes(C.byref(c)))")
    pkg = os.path.dirname(os.path.dirname(_lib.__file__))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TTL_BWD_COMPACT="0", PYTHONPATH=pkg), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and int(r.stdout) < base - 1e8, (r.stdout, r.stderr, base)      # the SelBuf set is ~30 % of the saved activations
