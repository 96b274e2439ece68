"""Independent cross-checks of the OpenCV restatements the oracle leans on (OpenCV itself is absent offline, so none of them can be
pinned bit for bit): each restated routine against the SAME operation computed another way with what is installed -- exact-arithmetic
bilinear sampling (torch.nn.functional.interpolate / plain NumPy) for cv2.resize and cv2.warpAffine.  The error bounds asserted are the
ones the fixed-point schemes of those OpenCV routines imply, so a wrong pixel-centre convention, border rule or tap order fails here."""
import numpy as np
import pytest

from oracle.crops import crop_obb, cv2_resize_linear_u8, cv2_warp_affine_inverse_linear_u8, obb_crop_geometry


@pytest.mark.parametrize("src_hw,dst_hw", [((300, 140), (256, 128)), ((97, 41), (256, 128)), ((640, 480), (256, 128)), ((31, 17), (256, 128)),
                                           ((500, 333), (128, 64)), ((20, 60), (256, 128)), ((256, 128), (256, 128))])
def test_resize_restatement_is_half_pixel_bilinear_with_11_bit_coefficients(src_hw, dst_hw):
    """cv2.resize(INTER_LINEAR) on uint8: half-pixel centres, clamped borders, coefficients quantised to 1/2048 and one final rounding
    => within 0.5 (final rounding) + about 0.25 (coefficient quantisation and the intermediate shift, two axes) of the exact bilinear
    value; measured 0.7506 at worst."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    rng = np.random.default_rng(src_hw[0] * 1000 + src_hw[1])
    src = rng.integers(0, 256, (*src_hw, 3), dtype=np.uint8)
    got = cv2_resize_linear_u8(src, (dst_hw[1], dst_hw[0])).astype(np.float64)
    t = torch.from_numpy(src.astype(np.float64)).permute(2, 0, 1)[None]
    ref = F.interpolate(t, size=dst_hw, mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 0.76
    if src_hw == dst_hw:
        assert np.array_equal(got, src)


def _exact_affine_bilinear(img, iM, out_wh):
    """dst(x, y) = bilinear sample of img at (iM[0] x + iM[1] y + iM[2], iM[3] x + iM[4] y + iM[5]), zero outside (BORDER_CONSTANT)."""
    h, w = img.shape[:2]
    ow, oh = out_wh
    xs, ys = np.meshgrid(np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64))
    sx, sy = iM[0] * xs + iM[1] * ys + iM[2], iM[3] * xs + iM[4] * ys + iM[5]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok[..., None], img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64), 0.0)
    return (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)


@pytest.mark.parametrize("box", [[320, 240, 120, 60, 0.3], [100, 90, 40, 80, -1.1], [600, 440, 90, 90, 2.5], [50, 50, 200, 30, 0.0]])
def test_warp_affine_restatement_samples_where_the_exact_map_does(box):
    """cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT): source coordinates in 1/32-pixel fixed point, 15-bit tap weights.  On a smooth
    image (gradient <= 1.5 levels per pixel) the result is within 1.2 levels of exact bilinear sampling at the unquantised coordinates;
    on noise it stays within what a 1/32-pixel coordinate error can do (<= 255 / 32 * 2 + 1)."""
    yy, xx = np.meshgrid(np.arange(480), np.arange(640), indexing="ij")
    smooth = np.stack([(0.2 * xx + 0.1 * yy) % 256, (0.15 * yy + 40) % 256, (0.1 * xx + 0.2 * yy + 80) % 256], axis=-1)
    smooth = np.where(np.abs(np.diff(smooth, axis=1, append=smooth[:, -1:])).max(axis=-1, keepdims=True) > 5, 128, smooth).astype(np.uint8)   # no wrap edges
    noise = np.random.default_rng(3).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    ow, oh, iM = obb_crop_geometry(box)
    for img, bound in ((smooth, None), (noise, 255 / 32 * 2 + 1)):
        got = cv2_warp_affine_inverse_linear_u8(img, iM, (ow, oh)).astype(np.float64)
        ref = _exact_affine_bilinear(img, iM, (ow, oh))
        assert got.shape == ref.shape == (oh, ow, 3)
        d = np.abs(got - ref)
        if bound is None:       # smooth: bounded by the local gradient; the flattened wrap seams are excluded by comparing robustly
            assert np.percentile(d, 99) <= 1.2 and np.median(d) <= 0.5
        else:
            assert d.max() <= bound
    assert np.array_equal(crop_obb(box, noise), cv2_warp_affine_inverse_linear_u8(noise, iM, (ow, oh)))
