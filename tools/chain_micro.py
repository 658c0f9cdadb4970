#!/usr/bin/env python3
"""Launch the bf16 chain / band kernels alone at 256 frames (for rocprofv3 --pmc passes and quick timings):
    python3 tools/chain_micro.py [reps]          -> us per launch of every (C, W) chain, 8 convolutions each"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
    m = pkg.build_synthetic_model(max_frames=4, with_gru=False, dtype="bf16")
    g = np.random.Generator(np.random.Philox(key=[88, 1]))
    for c, w in ((32, 56), (64, 28), (128, 14), (256, 7)):
        x = torch.from_numpy(g.standard_normal((256, c, w, w)).astype(np.float32)).cuda()
        ws = [(g.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))).astype(np.float32) for _ in range(8)]
        bs = [(g.standard_normal((c,)) * 0.1).astype(np.float32) for _ in range(8)]
        _, us = m.op_conv_chain(x, ws, bs, reps=reps)
        print(f"chain<{c},{w}> x8 @256 frames: {us:.1f} us per chain, {2.0 * 256 * w * w * c * c * 9 * 8 / us / 1e6:.0f} TFLOP/s")
    m.close()


if __name__ == "__main__":
    main()
