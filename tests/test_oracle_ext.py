"""Decoder-only features (predictor, cross-colour, colour cache, meta-Huffman): the oracle's
test-vector encoder must round-trip through the oracle decoder.  The inverse transforms the
decoder applies are pinned separately against the compiled reference (test_oracle_golden.py);
the container grammar around them follows webp/vp8l/vp8l.h:101-328."""
import itertools

import numpy as np
import pytest

import _oracle as O


def _img(mode, w, h, seed=7):
    if mode == "rand":
        return np.random.default_rng(seed).integers(0, 1 << 32, size=w * h, dtype=np.uint32)
    return O.gen_image(mode, w, h, seed, 200)


CASES = [
    dict(use_subtract_green=1),
    dict(predictor_bits=2),
    dict(predictor_bits=4, use_subtract_green=1, transform_order=0),
    dict(color_bits=3),
    dict(predictor_bits=3, color_bits=2, use_subtract_green=1, transform_order=1),
    dict(predictor_bits=2, color_bits=5, use_subtract_green=1, transform_order=2),
    dict(predictor_bits=2, color_bits=2, transform_order=3),
    dict(cache_bits=4),
    dict(cache_bits=11, use_subtract_green=1),
    dict(meta_bits=2, meta_groups=3),
    dict(meta_bits=3, meta_groups=7, cache_bits=6, predictor_bits=3, use_subtract_green=1),
    dict(use_palette=1, cache_bits=3),
    dict(use_palette=1, meta_bits=2, meta_groups=2),
]


@pytest.mark.parametrize("prm", CASES)
@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (33, 18), (64, 64)])
@pytest.mark.parametrize("mode", [3, 4, 5, "rand"])
def test_ext_roundtrip(prm, shape, mode):
    w, h = shape
    img = _img(mode, w, h)
    rc, data = O.encode_ext(img, w, h, **prm)
    assert rc == 0
    rc, dec, dw, dh = O.decode(data)
    assert rc == 0 and (dw, dh) == (w, h)
    assert np.array_equal(dec, img)


def test_ext_palette_small_counts():
    rng = np.random.default_rng(3)
    for ncol, (w, h) in itertools.product((1, 2, 3, 4, 5, 16, 17, 256), ((9, 4), (32, 3), (1, 7))):
        pal = rng.integers(0, 1 << 32, size=ncol, dtype=np.uint32)
        img = pal[rng.integers(0, ncol, size=w * h)]
        for prm in (dict(use_palette=1), dict(use_palette=1, cache_bits=5)):
            rc, data = O.encode_ext(img, w, h, **prm)
            assert rc == 0
            rc, dec, _, _ = O.decode(data)
            assert rc == 0 and np.array_equal(dec, img)
        # the reference-faithful encoder takes the same path for <=256 colours
        rc, data = O.encode(img, w, h)
        assert rc == 0
        rc, dec, _, _ = O.decode(data)
        assert rc == 0 and np.array_equal(dec, img)
