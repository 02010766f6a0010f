"""CPU checks of the frame-resize restatement: oracle/resize_ref.py against the golden bytes produced by Pillow
(tests/golden/make_resize_golden.py), against the Pillow installed here when importable, and the product's host-side
weight set-up (vidil_amd/preprocess.py) against the oracle's.  All bit-exact (integer arithmetic)."""
import os

import numpy as np
import pytest

from common import ROOT
from oracle import resize_ref as R


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "resize_golden.npz"))


def _cases(g):
    return sorted(k[:-3] for k in g.files if k.endswith("_in"))


def test_oracle_matches_golden_pillow_bytes():
    g = _golden()
    assert len(_cases(g)) == 10
    for key in _cases(g):
        img, S = g[key + "_in"], int(g[key + "_S"])
        assert np.array_equal(R.blip_process_frame_u8(img, S), g[key + "_blip"]), key
        assert np.array_equal(R.clip_process_frame_u8(img, S), g[key + "_clip"]), key


def test_oracle_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for h, w, oh, ow in [(360, 640, 224, 224), (240, 320, 224, 224), (100, 130, 224, 224), (480, 270, 224, 224),
                         (224, 300, 224, 224), (37, 53, 224, 224), (300, 224, 224, 224), (224, 224, 224, 224)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(R.resize_bicubic_u8(img, ow, oh), ref), (h, w)


def test_identity_geometry_is_a_copy():
    img = np.random.default_rng(1).integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(R.blip_process_frame_u8(img, 224), img)
    assert np.array_equal(R.clip_process_frame_u8(img, 224), img)


@pytest.mark.parametrize("in_size,out_size", [(640, 224), (360, 224), (53, 224), (224, 224), (398, 224), (1280, 384)])
def test_product_weight_tables_equal_oracle(in_size, out_size):
    from vidil_amd.preprocess import axis_weights

    ksize, bounds, kk = R.precompute_coeffs(in_size, 0.0, float(in_size), out_size)
    fixed = R.normalize_coeffs_8bpc(kk)
    k2, b2, c2 = axis_weights(in_size, out_size)
    assert k2 == ksize
    assert np.array_equal(np.array(b2), bounds)
    assert np.array_equal(np.array(c2), fixed)
    # a cropped window is the same rows of the same table
    k3, b3, c3 = axis_weights(in_size, out_size, 5, 17)
    assert np.array_equal(np.array(b3), bounds[5:22]) and np.array_equal(np.array(c3), fixed[5:22])
    # fixed-point weights of every output sum to ~1.0 (2^22 within a few ulps of rounding)
    assert np.all(np.abs(fixed.sum(1) - (1 << R.PRECISION_BITS)) <= ksize)


def test_clip_output_size_rule():
    from vidil_amd.preprocess import clip_resized_hw

    for h, w in [(360, 640), (640, 360), (224, 224), (225, 224), (100, 333), (719, 1279)]:
        assert clip_resized_hw(h, w, 224) == R.clip_output_size(h, w, 224)
    assert R.clip_output_size(360, 640, 224) == (224, 398)
