"""GPU: a C++ host program (examples/cabi_host.cpp; no Python, no torch in the loop) drives forward + backward through
the C ABI; its outputs must match the oracle on the inputs it generated.  The binary is built by
__graft_entry__.build() in the build container and travels to the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from oracle import dau_oracle as orc
from util import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_program_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "build", "cabi_host")
    if not os.path.exists(exe):      # normally built by __graft_entry__.build(); the GPU box has the same toolchain
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", "cabi_host.cpp"),
                               "-L" + os.path.join(ROOT, "dau-convnet_amd", "dau_conv"), "-ldau_conv_hip",
                               "-Wl,-rpath,$ORIGIN/../dau-convnet_amd/dau_conv", "-o", exe])
    out = str(tmp_path / "out.bin")
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    N, S, F, G, H, W, K, abi = np.frombuffer(raw[:32].tobytes(), dtype=np.int32)
    assert abi >= 1
    data = np.frombuffer(raw[32:].tobytes(), dtype=np.float32)
    shapes = [(N, S, H, W), (1, S, G, F), (1, S, G, F), (1, S, G, F), (N, F, H, W),
              (N, F, H, W), (N, S, H, W), (1, S, G, F), (1, S, G, F), (1, S, G, F), (1, S, G, F)]
    arrs, off = [], 0
    for sh in shapes:
        n = int(np.prod(sh)); arrs.append(data[off:off + n].reshape(sh).copy()); off += n
    assert off == data.size
    x, w, mu1, mu2, dy, y, dx, dw, dmu1, dmu2, dsigma = arrs
    assert_parity(y, orc.forward(x, w, mu1, mu2, 0.5), "y")
    want = orc.backward(x, dy, w, mu1, mu2, 0.5)
    for got, key in ((dx, "dx"), (dw, "dw"), (dmu1, "dmu1"), (dmu2, "dmu2"), (dsigma, "dsigma")):
        assert_parity(got, want[key], key)
