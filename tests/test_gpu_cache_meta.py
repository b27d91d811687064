"""Colour cache and meta-Huffman EMISSION (the reference encoder writes neither, vp8l.h:513-514; its decoder reads both,
vp8l.h:180-328).  Pinned like the other decoder-only features: the GPU encoder's stream equals the oracle's test-vector encoder
for the same parameters (cache bits, and the tile-to-group image the GPU chose), and decodes back to the input on the oracle
(whose reader of such streams is pinned against the reference's components) and on the GPU.  In the standard dialect libwebp
is the checker."""
import numpy as np
import pytest
import torch

import _oracle as O
import _webpstd as S
import webp_amd

pytestmark = pytest.mark.gpu


def encode_with(img, w, h, cache_bits=0, meta_bits=0, groups=4, pred=0, cross=0, standard=False):
    enc = webp_amd.Encoder()
    enc.set_cache_bits(cache_bits)
    enc.set_meta(meta_bits, groups)
    enc.set_predictor_bits(pred)
    enc.set_cross_color_bits(cross)
    enc.set_standard(standard)
    d_in = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint32).view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0, st
    stream = bytes(d_out[int(off[0]):int(off[0]) + int(lens[0])].cpu().numpy())
    info = enc.fetch_stage(0, webp_amd.STAGE_INFO)
    meta = enc.fetch_stage(0, webp_amd.STAGE_META_GROUPS) if meta_bits else None
    modes = enc.fetch_stage(0, webp_amd.STAGE_PRED_MODES) if pred else None
    elems = enc.fetch_stage(0, webp_amd.STAGE_CROSS_ELEMS) if cross else None
    enc.close()
    return stream, info, meta, modes, elems


def oracle_stream(img, w, h, info, cache_bits, meta, modes=None, elems=None):
    prm = dict(cache_bits=cache_bits)
    if info["ncolors"]:
        prm["use_palette"] = 1
    else:
        prm["use_subtract_green"] = 1
        if modes is not None:
            prm.update(predictor_bits=modes["bits"], predictor_modes=modes["modes"])
        if elems is not None:
            prm.update(color_bits=elems["bits"], color_elems=elems["elems"])
    if meta is not None and int(meta["groups"].max()) > 0:   # one group: no entropy image is written
        prm.update(meta_bits=meta["bits"], meta_groups=int(meta["groups"].max()) + 1, meta_image=meta["groups"])
    rc, s = O.encode_ext(img, w, h, **prm)
    assert rc == 0
    return s


CASES = [(3, 64, 64, 0), (4, 130, 40, 0), (5, 61, 19, 13), (5, 97, 61, 200), (3, 300, 200, 0), (4, 512, 512, 0), (5, 512, 300, 2), (3, 1, 1, 0),
         (4, 1, 77, 0), (3, 1031, 17, 0)]


@pytest.mark.parametrize("cache_bits,meta_bits,groups", [(4, 0, 0), (11, 0, 0), (1, 0, 0), (0, 2, 2), (0, 4, 8), (6, 3, 4), (10, 5, 3)])
def test_stream_equals_the_oracles_test_vector_encoder(cache_bits, meta_bits, groups):
    for mode, w, h, ncol in CASES:
        img = O.gen_image(mode, w, h, 21 + w, ncol or 200)
        s, info, meta, _, _ = encode_with(img, w, h, cache_bits, meta_bits, groups or 4)
        if meta is not None:
            assert meta["tx"] * meta["ty"] <= 1024 and int(meta["groups"].max()) + 1 == min(groups, meta["tx"] * meta["ty"])
        ref = oracle_stream(img, w, h, info, cache_bits, meta)
        assert len(s) == len(ref) and s == ref, (mode, w, h)
        rc, back, w1, h1 = O.decode(s)
        assert rc == 0 and (w1, h1) == (w, h) and np.array_equal(back, img)
        got, w2, h2 = webp_amd.decode(s)
        assert (w2, h2) == (w, h) and np.array_equal(got, img)


def test_with_predictor_and_cross_colour_in_front():
    for mode, w, h in ((3, 300, 200), (4, 257, 130)):
        img = O.gen_image(mode, w, h, 5, 200)
        s, info, meta, modes, elems = encode_with(img, w, h, cache_bits=7, meta_bits=4, groups=5, pred=3, cross=4)
        ref = oracle_stream(img, w, h, info, 7, meta, modes, elems)
        assert s == ref
        assert np.array_equal(O.decode(s)[1], img) and np.array_equal(webp_amd.decode(s)[0], img)


def test_cache_helps_where_pixels_come_back_beyond_the_match_window():
    """A picture of a few thousand colours scattered as noise cannot use the palette; a colour that returns after more than 1024
    pixels cannot be matched, but it often still sits in the cache."""
    rng = np.random.default_rng(9)
    w, h = 640, 400
    pal = rng.integers(0, 1 << 24, size=3000, dtype=np.uint32) | np.uint32(0xFF000000)
    img = pal[rng.integers(0, 3000, size=w * h)].astype(np.uint32)
    plain, _, _, _, _ = encode_with(img, w, h)
    cached, info, _, _, _ = encode_with(img, w, h, cache_bits=10)
    assert info["ncolors"] == 0
    assert len(cached) < 0.96 * len(plain)
    assert np.array_equal(webp_amd.decode(cached)[0], img) and np.array_equal(O.decode(cached)[1], img)


def test_groups_help_on_a_picture_with_flat_and_busy_halves():
    w, h = 512, 512
    a = O.gen_image(3, w, h, 3, 0).reshape(h, w)
    b = O.gen_image(4, w, h, 4, 0).reshape(h, w)
    img = np.where(np.arange(w)[None, :] < w // 2, a, b).astype(np.uint32).reshape(-1)
    plain, _, _, _, _ = encode_with(img, w, h)
    grouped, _, meta, _, _ = encode_with(img, w, h, meta_bits=5, groups=4)
    assert len(grouped) < len(plain)
    assert np.array_equal(webp_amd.decode(grouped)[0], img)


@pytest.mark.skipif(S.pillow() is None, reason="Pillow with WebP support is not importable")
def test_standard_dialect_streams_with_cache_and_groups_decode_in_libwebp():
    for mode, w, h, ncol in CASES[:7]:
        img = O.gen_image(mode, w, h, 77, ncol or 200)
        for cache_bits, meta_bits in ((6, 0), (0, 3), (11, 4)):
            s, _, _, _, _ = encode_with(img, w, h, cache_bits, meta_bits, 4, standard=True)
            got, w1, h1 = S.libwebp_decode(webp_amd.riff_container(s))
            assert (w1, h1) == (w, h) and np.array_equal(got, img), (mode, w, h, cache_bits, meta_bits)
            assert np.array_equal(webp_amd.decode(s)[0], img)


def test_one_shot_flags_and_a_batch_of_4k_images():
    img = O.gen_image(3, 200, 100, 1, 0)
    s = webp_amd.encode(img, 200, 100, cache_bits=5, meta_bits=3)
    assert np.array_equal(webp_amd.decode(s)[0], img) and np.array_equal(O.decode(s)[1], img)
    w, h = 3840, 2160
    d_in = torch.empty((3, w * h), dtype=torch.int32, device="cuda")
    for k, mode in enumerate((3, 4, 5)):
        webp_amd.generate_device(mode, w, h, 12345 + k, 200, d_in[k].data_ptr())
    enc = webp_amd.Encoder()
    enc.set_cache_bits(9)
    enc.set_meta(6, 8)
    cap = enc.bound(w, h)
    d_out = torch.empty(3 * cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in[k].data_ptr() for k in range(3)], [w] * 3, [h] * 3, d_out.data_ptr(), d_out.numel())
    assert (st == 0).all()
    dec = webp_amd.Decoder()
    d_dec = torch.zeros((3, w * h), dtype=torch.int32, device="cuda")
    st = dec.decode_device([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], [w] * 3, [h] * 3, [d_dec[k].data_ptr() for k in range(3)])
    assert (st == 0).all() and torch.equal(d_dec, d_in)
