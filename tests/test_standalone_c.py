"""The C ABI is usable from plain C: examples/standalone_forward.c includes include/ttl_hip.h, is built with gcc
(no hipcc, no torch, no Python on its side) and drives the library through dlopen.  CPU: it compiles as C.
GPU: its logits equal the Python engine's bit for bit (same library, same kernels) and match the reference fixture."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "examples", "standalone_forward.c")


def build(tmp_path):
    exe = str(tmp_path / "standalone_forward")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", SRC, "-L", "/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def write_bundle(tmp_path, cfg, W, tf, scale, flat, x):
    """the example's input file: geometry, every tensor by name, class features, logit scale, adapters, views"""
    N, K = x.shape[0], tf.shape[0]
    bundle = tmp_path / "bundle.bin"
    with open(bundle, "wb") as f:
        f.write(struct.pack("<8i", cfg.image_size, cfg.patch_size, cfg.width, cfg.heads, cfg.mlp, cfg.layers, cfg.embed, cfg.rank))
        f.write(struct.pack("<f2if2i", cfg.lora_alpha, cfg.layer_lo, cfg.layer_hi, cfg.ln_eps, N, K))
        tensors = {k: v for k, v in W.items() if k != "logit_scale"}
        f.write(struct.pack("<3i", N, K, len(tensors)))
        for k, v in tensors.items():
            a = np.ascontiguousarray(v, np.float32)
            f.write(struct.pack("<i", len(k)) + k.encode() + struct.pack("<q", a.size) + a.tobytes())
        f.write(np.ascontiguousarray(tf, np.float32).tobytes())
        f.write(struct.pack("<f", scale))
        f.write(flat.tobytes())
        f.write(np.ascontiguousarray(x, np.float32).tobytes())
    return bundle


def test_header_and_example_compile_as_plain_c(tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_standalone_c_host_matches_python_engine(tmp_path):
    import torch
    from oracle import ttl_oracle as O
    from helpers import load_case, max_rel
    from ttl_amd import _lib
    from ttl_amd.engine import TTLEngine
    g, cfg, W, x, lora0, tf = load_case("tiny_deyo")
    N, K = x.shape[0], tf.shape[0]
    names = O.trainable_names(cfg)
    flat = np.concatenate([lora0[k].reshape(-1) for k in names]).astype(np.float32)
    scale = float(np.exp(W["logit_scale"]))
    bundle = write_bundle(tmp_path, cfg, W, tf, scale, flat, x)
    exe = build(tmp_path)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, _lib.LIB_PATHS[_lib.DEFAULT_PRECISION], str(bundle), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.float32)
    z0, z1 = got[:N * K].reshape(N, K), got[N * K:].reshape(1, K)
    # the same calls through the Python wrapper
    eng = TTLEngine(cfg, N, K, "cuda:0")
    eng.load_weights(W)
    eng.set_text_features(torch.from_numpy(tf), scale)
    fl = torch.from_numpy(flat).cuda()
    eng.bind_lora(fl)
    p0 = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    p1 = eng.episode(torch.from_numpy(x).cuda(), fl.clone(), torch.zeros_like(fl), torch.zeros_like(fl), n_updates=1).cpu().numpy()
    assert np.array_equal(z0, p0) and np.array_equal(z1, p1)
    assert max_rel(z0, g["logits0"]) < 3e-2 and int(np.argmax(z1)) == int(g["top5"][0, 0])
    eng.close()
