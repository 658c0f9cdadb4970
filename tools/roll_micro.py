#!/usr/bin/env python3
"""The row-walking launches of csrc/conv_bf16_roll.hip alone at n frames (grnet_time_conv: 20 back-to-back launches between two HIP events):
    python3 tools/roll_micro.py [n_frames]
With a diagnostic build (GRNET_LIB_PATH=.../libgrnet_hip_abl.so) GRNET_ROLL_DBG drops parts of the work (timing only, results garbage):
bit 0 the 1x1 reduce's / conv1's MFMAs, 1 the 3x3, 2 the expansion, 3 the stores, 4 the input loads."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
    m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for pos, c in enumerate(m.describe_convs()):
        name = C.create_string_buffer(96)
        m._lib.grnet_conv_kernel_info(m._h, pos, n, name, 96, None)
        k = name.value.decode()
        if k in ("conv_bf16_stem_pair", "conv_bf16_bneck"):
            us = C.c_float()
            m._lib.grnet_time_conv(m._h, pos, n, 20, stream, C.byref(us))
            out.append(f"{k} {c['cin']}->{c['cout']}: {us.value:.1f} us")
    print(f"n={n} GRNET_ROLL_DBG={os.environ.get('GRNET_ROLL_DBG', '0')}: " + " | ".join(out[:3]))
    m.close()


if __name__ == "__main__":
    main()
