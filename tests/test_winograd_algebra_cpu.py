"""The transform matrices the Winograd kernels are built on (csrc/conv_wino.hip: F(2x2,3x3); csrc/conv_wino4.hip: F(4x4,3x3)),
checked on the CPU: A^T [ (G g G^T) . (B^T d B) ] A equals the 3x3 correlation of the patch exactly (fp64), summed over channels
like the kernels do, and in fp32 stays within the error the per-layer GPU tests allow.  The same constants appear in the kernels'
1-D transform helpers (bt_lo / bt_hi, the epilogue's A^T rows) and in pack_wino*_weights (G)."""
import numpy as np
import pytest

F23 = dict(
    BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
    G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
    AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64), m=2)
F43 = dict(
    BT=np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                 [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
    G=np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
    AT=np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64), m=4)


def direct(d, g):
    """(C, m+2, m+2) patch, (C, 3, 3) filter -> (m, m) correlation summed over channels (what nn.Conv2d computes)."""
    m = d.shape[1] - 2
    return np.array([[np.sum(d[:, i:i + 3, j:j + 3] * g) for j in range(m)] for i in range(m)])


def winograd(d, g, t, dtype):
    BT, AT = t["BT"].astype(dtype), t["AT"].astype(dtype)
    U = np.einsum("ia,cab,jb->cij", t["G"], g.astype(np.float64), t["G"]).astype(dtype)      # filter transform in fp64, stored in dtype
    V = np.einsum("ia,cab,jb->cij", BT, d.astype(dtype), BT).astype(dtype)
    M = np.zeros(U.shape[1:], dtype)
    for c in range(U.shape[0]):                                                              # channel sum in the kernels' precision
        M = (M + U[c] * V[c]).astype(dtype)
    return (AT @ M @ AT.T).astype(dtype)


@pytest.mark.parametrize("t", [F23, F43], ids=["F(2x2,3x3)", "F(4x4,3x3)"])
def test_winograd_identity_is_exact_in_fp64(t):
    rng = np.random.default_rng(3)
    for C in (1, 8, 40):
        d = rng.standard_normal((C, t["m"] + 2, t["m"] + 2))
        g = rng.standard_normal((C, 3, 3))
        assert np.allclose(winograd(d, g, t, np.float64), direct(d, g), rtol=0, atol=1e-11 * C)


@pytest.mark.parametrize("t,bound", [(F23, 5e-6), (F43, 5e-5)], ids=["F(2x2,3x3)", "F(4x4,3x3)"])
def test_winograd_fp32_rounding_stays_inside_the_layer_bound(t, bound):
    """fp32 transforms and accumulation on a 256-channel layer with He-scaled weights: the distance to the exact sum, relative to
    the output rms, is ~1e-6 (F(2x2,3x3)) / ~1e-5 (F(4x4,3x3)); the per-layer GPU tests allow 2e-5 / 1e-4 of the output maximum."""
    rng = np.random.default_rng(4)
    C, worst = 256, 0.0
    for _ in range(6):
        d = rng.standard_normal((C, t["m"] + 2, t["m"] + 2)).astype(np.float32)
        g = (rng.standard_normal((C, 3, 3)) * np.sqrt(2 / (C * 9))).astype(np.float32)
        ref = direct(d.astype(np.float64), g.astype(np.float64))
        worst = max(worst, float(np.abs(winograd(d, g, t, np.float32) - ref).max() / np.sqrt(np.mean(ref ** 2))))
    assert worst < bound, worst
