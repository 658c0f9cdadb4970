"""Pin the crop + normalise row (f1) to the REAL OpenCV, the first time `cv2` exists (it is "parity unpinned" until then: opencv-python is not installed
offline, requirements.txt; the oracle restates cv2.getAffineTransform / cv2.warpAffine from their published algorithm and is checked by known-answer cases).

    python tools/check_cv2.py [--n 40] [--seed 0]

If `import cv2` works, for n random 8-bit frames of random sizes and random boxes (square and non-square, inside and hanging over the border, scale 1.0 / 1.1):
  1. cv2.getAffineTransform on the float32 triangles of gen_trans_from_patch_cv (img_utils.py:54-88) against oracle.gen_trans_from_patch   (<= 1e-9 absolute);
  2. cv2.warpAffine(img, M, (w,h), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT) against oracle.warp_affine_u8(img, invert_affine_cv(M))   -- BIT-EXACT;
  3. the whole generate_patch_image_cv (img_utils.py:90-113, both branches) written with cv2 calls here against oracle.patch_image_cv   -- BIT-EXACT;
  4. on a GPU box: GRNet.crop_normalise (grnet_crop_normalise_cv_maps, the HIP kernel) against ToTensor + Normalize of (3)   (<= 1e-6: same uint8 patch, one fp32 expression).
Exit code 1 on any mismatch; skips cleanly (exit code 0, says why) without cv2.  Test infrastructure: imports oracle/."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "video-based-gait-analysis-for-dementia_amd"


def cv2_patch(cv2, img, bbox, scale, patch=224):
    """generate_patch_image_cv(img, c_x, c_y, bb_w, bb_h, patch, patch, do_flip=False, scale, rot=0) spelled with cv2 calls (img_utils.py:90-113)."""
    c_x, c_y, bw, bh = [float(v) for v in bbox]

    def trans(dst_w, dst_h):
        src_w, src_h = bw * scale, bh * scale
        centre = np.array([c_x, c_y], np.float64)
        src = np.zeros((3, 2), np.float32)
        src[0] = centre
        src[1] = centre + np.array([0, src_h * 0.5], np.float32)
        src[2] = centre + np.array([src_w * 0.5, 0], np.float32)
        dc = np.array([dst_w * 0.5, dst_h * 0.5], np.float32)
        dst = np.zeros((3, 2), np.float32)
        dst[0] = dc
        dst[1] = dc + np.array([0, dst_h * 0.5], np.float32)
        dst[2] = dc + np.array([dst_w * 0.5, 0], np.float32)
        return cv2.getAffineTransform(np.float32(src), np.float32(dst))

    if bw != bh:
        s = patch / max(bh, bw)
        iw, ih = int(s * bw), int(s * bh)
        first = cv2.warpAffine(img, trans(iw, ih), (iw, ih), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT)
        dx, dy = patch / 2 - first.shape[1] / 2, patch / 2 - first.shape[0] / 2
        return cv2.warpAffine(first, np.array([[1, 0, dx], [0, 1, dy]]).astype(np.float64), (patch, patch), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT), trans(iw, ih), (iw, ih)
    m = trans(patch, patch)
    return cv2.warpAffine(img, m, (patch, patch), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT), m, (patch, patch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("check_cv2: SKIPPED -- the cv2 package is not installed (pip install opencv-python where a network exists)")
        return 0
    from oracle import grnet_oracle as oracle
    rng = np.random.default_rng(a.seed)
    bad = 0
    cases = []
    for k in range(a.n):
        h, w = int(rng.integers(120, 700)), int(rng.integers(160, 900))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        bw = float(rng.uniform(40, 1.2 * w))
        bh = bw if k % 2 == 0 else float(rng.uniform(40, 1.2 * h))
        bbox = np.array([rng.uniform(-0.1 * w, 1.1 * w), rng.uniform(-0.1 * h, 1.1 * h), bw, bh], np.float32)
        scale = 1.1 if k % 3 else 1.0
        want, m_cv, (ow, oh) = cv2_patch(cv2, img, bbox, scale)
        m_or = oracle.gen_trans_from_patch(float(bbox[0]), float(bbox[1]), float(bbox[2]), float(bbox[3]), ow, oh, scale)
        dm = float(np.abs(np.asarray(m_cv, np.float64).reshape(-1) - m_or).max())
        one = oracle.warp_affine_u8(img, oracle.invert_affine_cv(np.asarray(m_cv, np.float64).reshape(-1)), (ow, oh))
        one_cv = cv2.warpAffine(img, m_cv, (ow, oh), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT)
        got = oracle.patch_image_cv(img, bbox, scale)
        ok = dm <= 1e-9 and np.array_equal(one, one_cv) and np.array_equal(got, want)
        bad += not ok
        print(f"case {k:3d}: {w}x{h} box {bbox.round(1).tolist()} scale {scale}: |M - M_cv| {dm:.2e}, warp mismatches {int((one != one_cv).sum())}, "
              f"patch mismatches {int((got != want).sum())}  {'ok' if ok else 'MISMATCH'}")
        cases.append((img, bbox, scale, want))
    try:
        import torch
        gpu = torch.cuda.is_available()
    except ImportError:
        gpu = False
    if gpu:
        pkg = importlib.import_module(PKG)
        model = pkg.build_synthetic_model(max_frames=1, device_id=0, with_gru=False)
        worst = 0.0
        for img, bbox, scale, want in cases:
            crop = model.crop_normalise(torch.from_numpy(img).cuda().unsqueeze(0), torch.from_numpy(bbox).unsqueeze(0), scale=scale)[0].cpu().numpy()
            ref = ((want.astype(np.float32) / np.float32(255.0) - oracle.IMAGENET_MEAN) / oracle.IMAGENET_STD).transpose(2, 0, 1)
            worst = max(worst, float(np.abs(crop - ref).max()))
        print(f"HIP crop kernel vs cv2 patches: max |difference| {worst:.2e} (normalised units)")
        bad += worst > 1e-6
    else:
        print("no GPU here: the HIP kernel leg is skipped (tests/test_gpu_round4.py holds it bit-exact against the oracle)")
    print("check_cv2:", "FAILED" if bad else "ok -- row f1 is pinned to this cv2 " + cv2.__version__)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
