"""CPU only: the HOST side of libttl_hip under AddressSanitizer (SURVEY §5 'Sanitizers'; GPU ASan does not exist on the target
pool).  `make asan` compiles api.hip as plain C++ against a host-only stand-in of the HIP runtime and of the launch layer
(csrc/asan/: "device" memory is heap memory, the weight-path kernels run on the host as the kernels index, every other launch
touches the extents of its operands) and builds csrc/asan/asan_host_test.cpp — config validation, weight loading by name,
shared contexts, the launch sequences of several geometries / adapter sets / both towers, graphs, debug copies, error paths —
and the plain-C example.  Any out-of-bounds access, use-after-free or leak on those paths fails the run."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1")


@pytest.fixture(scope="module")
def asan_build():
    subprocess.check_call(["make", "-C", CSRC, "asan"], stdout=subprocess.DEVNULL)
    return os.path.join(CSRC, "build", "asan")


def test_host_glue_under_address_sanitizer(asan_build):
    r = subprocess.run([os.path.join(asan_build, "asan_host_test")], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "ASAN_HOST_OK" in r.stdout and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_plain_c_example_under_address_sanitizer(asan_build, tmp_path):
    """examples/standalone_forward.c (dlopen + the C ABI, its own file parsing and buffer arithmetic) built with ASan and run
    against the ASan build of the library; the stub kernels make its numbers meaningless, its memory accesses are real."""
    from oracle import ttl_oracle as O
    from helpers import load_case
    from test_standalone_c import write_bundle
    g, cfg, W, x, lora0, tf = load_case("tiny_deyo")
    flat = np.concatenate([lora0[k].reshape(-1) for k in O.trainable_names(cfg)]).astype(np.float32)
    bundle = write_bundle(tmp_path, cfg, W, tf, float(np.exp(W["logit_scale"])), flat, x)
    out = tmp_path / "out.bin"
    r = subprocess.run([os.path.join(asan_build, "standalone_forward_asan"), os.path.join(asan_build, "libttl_hip_asan.so"), str(bundle), str(out)],
                       capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, (r.stdout[-1000:], r.stderr[-4000:])
    assert os.path.getsize(out) == (x.shape[0] * tf.shape[0] + tf.shape[0]) * 4
