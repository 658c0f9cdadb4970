"""The environment-variable surface of libgrnet_hip.so (round-5 review): the product build reads five documented variables (include/grnet_hip.h,
"Environment"); every other GRNET_* name of earlier rounds is a compile-time constant there (csrc/kernels.h: GRNET_AB) and must not change a bit."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import importlib, sys, hashlib
import numpy as np, torch
sys.path.insert(0, {root!r})
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
digest = []
for dtype, n in (("f32", 8), ("bf16", 64)):
    m = pkg.build_synthetic_model(max_frames=n, with_gru=(dtype == "f32"), dtype=dtype)
    frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (n // 8, 1, 1, 1))).cuda()
    out = m(frames, extras=("features",))[-1]
    torch.cuda.synchronize()
    digest += [str(m.num_kernel_launches())] + [hashlib.sha256(out[k].cpu().numpy().tobytes()).hexdigest() for k in ("features", "theta", "verts")]
    if dtype == "f32":
        x, cp = pkg.synth.make_gru_inputs(2, 40)
        y, ph, _ = m.gru_forward(torch.from_numpy(x).cuda(), torch.from_numpy(cp).cuda())
        digest.append(hashlib.sha256(y.cpu().numpy().tobytes() + ph.cpu().numpy().tobytes()).hexdigest())
    m.close()
print("RESULT", *digest)
"""


def _run(env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("GRNET_"):
            e.pop(k)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1].split()[1:]


def test_retired_environment_variables_change_nothing():
    """Variables that used to switch kernels (GRNET_WINO4S, GRNET_BF16_PW_STREAM), numerics (GRNET_GRU_SPLIT: libm against v_exp gate functions,
    GRNET_WINO4) or the plan / schedule (GRNET_BF16_FUSE_UP, GRNET_FUSE_UP, GRNET_LANES, GRNET_BF16_CHAIN_MIN) are set to their non-default
    values: the fp32 and the bf16 forward, the launch counts and the GRU must come out bit for bit as without them."""
    base = _run({})
    retired = {"GRNET_WINO4S": "0", "GRNET_WINO4": "0", "GRNET_GRU_SPLIT": "0", "GRNET_GRU_AGENT": "1", "GRNET_BF16_FUSE_UP": "1", "GRNET_FUSE_UP": "0",
               "GRNET_LANES": "1", "GRNET_BF16_CHAIN_MIN": "8", "GRNET_BF16_PW_STREAM": "0", "GRNET_PW": "0", "GRNET_EDGEPTR": "0", "GRNET_BF16_POOL_WAVES": "6",
               "GRNET_ABL_SKIP": "layer1", "GRNET_STEM": "0", "GRNET_WINO_WIDE": "15"}
    assert _run(retired) == base


def test_documented_variables_still_act():
    """GRNET_BF16_CHAIN and GRNET_WINO are the process-wide defaults of two grnet_set_option values: they DO change the launch plan (and, within the
    documented noise, the bits); GRNET_MULTI_LANE=0 and GRNET_TRACE=1 change neither launches nor bits."""
    base = _run({})
    assert _run({"GRNET_MULTI_LANE": "0", "GRNET_TRACE": "1"}) == base
    nochain = _run({"GRNET_BF16_CHAIN": "0"})
    assert nochain[:5] == base[:5] and int(nochain[5]) > int(base[5])          # the fp32 leg is untouched; the bf16 leg launches every convolution
    nowino = _run({"GRNET_WINO": "0"})
    assert nowino[1] != base[1] and nowino[5:] == base[5:]                    # the fp32 features come from the direct kernels; the bf16 leg is untouched
