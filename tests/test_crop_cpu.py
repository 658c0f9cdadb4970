"""CPU tests of the crop's geometry (row f1): the oracle's restatement of generate_patch_image_cv (img_utils.py:90-113) -- both the
single warp of a square box and the TWO warps of a non-square one (:97-106) -- on known answers, and the product's host-side map
records (pipeline.cv_crop_maps, what the HIP kernel evaluates) against the oracle's own derivation."""
import numpy as np


def _img(h, w, seed):
    return np.random.Generator(np.random.Philox(key=[41, seed])).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_two_warp_crop_2to1_box_is_decimation_plus_letterbox(oracle):
    """A 448 x 224 box centred on a 448-wide image, scale 1: s = 224/448 = 1/2, the first warp is the exact 2:1 decimation (every
    sampled position is an integer pixel), iw x ih = 224 x 112, the second warp moves it down by 224/2 - 112/2 = 56 rows; rows
    outside stay 0 -- the reference's aspect-preserving crop."""
    img = _img(448, 448, 1)
    p = oracle.patch_image_cv(img, [224, 224, 448, 224], 1.0)
    exp = np.zeros((224, 224, 3), np.uint8)
    exp[56:168] = img[112:336:2, 0:448:2]
    assert np.array_equal(p, exp)
    tall = oracle.patch_image_cv(img, [224, 224, 224, 448], 1.0)               # the transposed case: letterbox left and right
    exp_t = np.zeros((224, 224, 3), np.uint8)
    exp_t[:, 56:168] = img[0:448:2, 112:336:2]
    assert np.array_equal(tall, exp_t)


def test_two_warp_crop_half_pixel_offset_blends_neighbours(oracle):
    """A 111 x 224 box: s = 1, iw = 111 is odd, so the second warp's offset is 112 - 55.5 = 56.5 pixels and every patch pixel is
    the rounded mean of two neighbours of the (exactly copied) first image: (a + b + 1) >> 1 by the 15-bit fixed-point blend."""
    img = _img(224, 300, 2)
    p = oracle.patch_image_cv(img, [100.5, 112, 111, 224], 1.0).astype(np.int64)
    first = np.zeros((224, 113, 3), np.int64)                                  # one zero column on either side
    first[:, 1:112] = img[:, 45:156]                                           # src x = u - 55.5 + 100.5 = u + 45
    exp = np.zeros((224, 224, 3), np.int64)
    for u in range(224):
        a, b = u - 57, u - 56                                                  # the two taps of position u - 56.5
        va = first[:, a + 1] if -1 <= a < 112 else 0
        vb = first[:, b + 1] if -1 <= b < 112 else 0
        exp[:, u] = (va + vb + 1) >> 1
    assert np.array_equal(p, exp)


def test_square_box_is_one_warp(oracle):
    """w == h: one warp straight into the patch (the `else` branch, img_utils.py:107-108); an integer-aligned unit-scale box copies."""
    img = _img(300, 400, 3)
    p = oracle.patch_image_cv(img, [200, 150, 224, 224], 1.0)
    assert np.array_equal(p, img[38:262, 88:312])


def test_product_crop_maps_reproduce_the_oracle_patch(pkg, oracle):
    """pipeline.cv_crop_maps (the records grnet_crop_normalise_cv_maps evaluates on the GPU) applied with the oracle's warpAffine, in
    two steps where iw > 0, gives the oracle's patch_image_cv -- which derives its own maps -- for square and non-square boxes, boxes
    hanging over the border, float32 and float64 boxes, scale 1.0 / 1.1 / 1.21."""
    img = _img(260, 340, 4)
    g = np.random.Generator(np.random.Philox(key=[41, 5]))
    boxes = []
    for _ in range(24):
        w, h = g.uniform(40, 400, 2)
        if g.random() < 0.4:
            h = w
        boxes.append([g.uniform(-20, 360), g.uniform(-20, 280), w, h])
    boxes = np.array(boxes + [[170.0, 130.0, 300.0, 150.0], [30.25, 240.5, 101.0, 224.0]], np.float64)
    for dtype in (np.float64, np.float32):
        bb = boxes.astype(dtype)
        for scale in (1.0, 1.1, 1.21):
            maps = pkg.pipeline.cv_crop_maps(bb, scale)
            assert maps.shape == (len(bb), 10) and maps.dtype == np.float64
            for b, m in zip(bb, maps):
                ref = oracle.patch_image_cv(img, b, scale)
                if m[6] == 0:
                    assert float(b[2]) == float(b[3])
                    got = oracle.warp_affine_u8(img, m[:6], 224)
                else:
                    first = oracle.warp_affine_u8(img, m[:6], (int(m[6]), int(m[7])))
                    got = oracle.warp_affine_u8(first, [1.0, 0.0, m[8], 0.0, 1.0, m[9]], 224)
                assert np.array_equal(got, ref)
