"""Reference unit tests of the ReID pre-processing (tests/unit/test_reid_preprocessing.py:8-18, tests/unit/test_base_backend.py:36-72)
restated on the oracle's crop functions -- same inputs, same assertions; where /root/reference is mounted, the reference's own
`resize_pad` runs beside the oracle's under the cv2 stand-in and must return the same bytes."""
import numpy as np
import pytest

from oracle import ref_harness
from oracle.crops import get_crops, is_obb, resize_pad_u8

IMAGENET_MEAN_BGR = (104, 116, 124)         # preprocessing.py IMAGENET_MEAN_BGR (int(round(255 * mean)) of the RGB mean, reversed)


def test_resize_pad_uses_bgr_imagenet_mean_padding_for_opencv_crops():
    crop_color_bgr = (7, 13, 19)
    crop = np.full((10, 4, 3), crop_color_bgr, dtype=np.uint8)
    padded = resize_pad_u8(crop, (10, 10))
    mean_bgr = np.asarray(IMAGENET_MEAN_BGR, dtype=np.uint8)
    assert padded.shape == (10, 10, 3)
    assert np.all(padded[:, :3] == mean_bgr)
    assert np.all(padded[:, 7:] == mean_bgr)
    assert np.all(padded[:, 3:7] == np.asarray(crop_color_bgr, dtype=np.uint8))
    if ref_harness.reference_available():
        ref_harness.install_standins()
        # the module alone (the reid package's __init__ pulls in torchvision, absent here)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_reid_preprocessing", ref_harness.REFERENCE_ROOT / "boxmot" / "reid" / "core" / "preprocessing.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        REF_MEAN, resize_pad = mod.IMAGENET_MEAN_BGR, mod.resize_pad
        assert tuple(int(v) for v in REF_MEAN) == IMAGENET_MEAN_BGR
        assert np.array_equal(resize_pad(crop, (10, 10)), padded)
        rng = np.random.default_rng(0)
        for hw in ((37, 91), (120, 33), (256, 128), (5, 300)):
            c = rng.integers(0, 255, hw + (3,), dtype=np.uint8)
            assert np.array_equal(resize_pad(c, (256, 128)), resize_pad_u8(c, (256, 128))), hw


def test_boxes_layouts_and_obb_crops_like_the_base_backend_tests():
    """test_base_backend.py:36-72: AABB rows pass through; a 5-column (cx, cy, w, h, angle) table is oriented; an axis-aligned OBB
    box crops the rectangle it covers."""
    assert not is_obb(np.array([[10, 20, 30, 40]], dtype=np.float32))
    assert is_obb(np.array([[32, 24, 20, 10, 0.0]], dtype=np.float32))
    img = np.zeros((64, 64, 3), dtype=np.uint8)
    img[19:29, 22:42] = 255
    crops = get_crops(np.array([[32, 24, 20, 10, 0.0]], dtype=np.float32), img, input_shape=(16, 8))
    assert crops.shape == (1, 3, 16, 8)
    assert np.count_nonzero(crops) > 0
    # the rectangle is white: after (x / 255 - mean) / std every interior pixel is the normalised 1.0 of its channel
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    want = (np.float32(1.0) - mean) / std
    assert np.allclose(crops[0, :, 4:12, 2:6], want[:, None, None], atol=1e-5)
