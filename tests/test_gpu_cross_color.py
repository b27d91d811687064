"""The optional forward cross-colour transform.  The reference ENCODER has none; its DECODER reads it
(ReadTransform vp8l.h:111-121, inverse transform.h:192-218).  Checked here:
  * the stream equals what the oracle's test-vector encoder writes for the SAME per-tile elements (and predictor
    modes, when both transforms are on), so residuals, tile images and the coded image are as the dialect prescribes,
  * the oracle's decoder (pinned against the reference's inverse transforms) and the GPU decoder give back the input,
  * an image whose channels are scaled copies of each other gets smaller."""
import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def encode(img, w, h, cross_bits, pred_bits=0):
    import torch
    enc = webp_amd.Encoder()
    enc.set_cross_color_bits(cross_bits)
    enc.set_predictor_bits(pred_bits)
    d_in = torch.from_numpy(np.ascontiguousarray(img).view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0
    stream = bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy())
    ce = enc.fetch_stage(0, webp_amd.STAGE_CROSS_ELEMS)
    pm = enc.fetch_stage(0, webp_amd.STAGE_PRED_MODES)
    info = enc.fetch_stage(0, webp_amd.STAGE_INFO)
    enc.close()
    return stream, ce, pm, info


def correlated_image(w, h, seed):
    """red = 0.6 green + noise, blue = 0.3 green + 0.4 red + noise: what subtract green alone leaves a lot of.
    (Green stays below 128: the transform multiplies SIGNED bytes, transform.h:192-195.)"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 128, size=w * h)
    r = (g * 6 // 10 + rng.integers(0, 4, size=w * h)) & 0xFF
    b = (g * 3 // 10 + r * 4 // 10 + rng.integers(0, 4, size=w * h)) & 0xFF
    return (np.uint32(0xFF000000) | (r.astype(np.uint32) << 16) | (g.astype(np.uint32) << 8) | b.astype(np.uint32))


CASES = [(3, 64, 64, 2, 0), (3, 200, 50, 3, 0), (4, 333, 77, 4, 0), (3, 61, 19, 2, 2), (3, 1, 40, 2, 0), (3, 40, 1, 5, 3),
         ("corr", 512, 512, 5, 0), ("corr", 700, 300, 2, 4), (3, 1030, 9, 9, 0), ("corr", 100, 100, 3, 3)]


@pytest.mark.parametrize("mode,w,h,bits,pred", CASES)
def test_stream_equals_oracle_for_the_same_elements(mode, w, h, bits, pred):
    img = correlated_image(w, h, 5 + w) if mode == "corr" else O.gen_image(mode, w, h, 31 + w, 200)
    stream, ce, pm, info = encode(img, w, h, bits, pred)
    if info["ncolors"]:                      # few colours: colour indexing, which excludes the other transforms
        rc, ref = O.encode(img, w, h)
        assert ce is None and rc == 0 and stream == ref
        return
    assert ce is not None and ce["bits"] >= bits and ce["tx"] * ce["ty"] <= 1024
    assert ce["tx"] == (w + (1 << ce["bits"]) - 1) >> ce["bits"] and ce["ty"] == (h + (1 << ce["bits"]) - 1) >> ce["bits"]
    assert ((ce["elems"] >> 24) == 0xFF).all()
    prm = dict(use_subtract_green=1, color_bits=ce["bits"], color_elems=ce["elems"], transform_order=0)
    if pred:
        assert pm is not None
        prm.update(predictor_bits=pm["bits"], predictor_modes=pm["modes"])
    rc, ref = O.encode_ext(img, w, h, **prm)
    assert rc == 0
    assert stream == ref
    rc, back, bw, bh = O.decode(stream)
    assert rc == 0 and (bw, bh) == (w, h) and np.array_equal(back, img)
    got, gw, gh = webp_amd.decode(stream)
    assert (gw, gh) == (w, h) and np.array_equal(got, img)


def test_correlated_channels_get_smaller():
    w, h = 640, 360
    img = correlated_image(w, h, 3)
    plain, _, _, _ = encode(img, w, h, 0)
    cross, ce, _, _ = encode(img, w, h, 5)
    assert len(cross) < 0.85 * len(plain), (len(cross), len(plain))
    e = ce["elems"]
    # the multipliers found sit near the factors the image was made with, relative to what subtract green removed:
    # red - green = -0.4 green  -> g2r near -0.4 * 32
    g2r = (e & 0xFF).astype(np.int8)
    assert abs(int(np.median(g2r)) + 13) <= 3, np.median(g2r)
    got, gw, gh = webp_amd.decode(cross)
    assert np.array_equal(got, img)


def test_host_entry_point_flags():
    w, h = 200, 120
    img = correlated_image(w, h, 9)
    both = webp_amd.encode(img, w, h, predictor_bits=3, cross_color_bits=4)
    cross = webp_amd.encode(img, w, h, cross_color_bits=4)
    plain = webp_amd.encode(img, w, h)
    rc, ref = O.encode(img, w, h)
    assert rc == 0 and plain == ref
    assert both != cross and cross != plain
    for s in (both, cross):
        rc, back, bw, bh = O.decode(s)
        assert rc == 0 and np.array_equal(back, img)
