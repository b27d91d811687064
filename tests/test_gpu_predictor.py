"""The optional forward predictor transform.  The reference ENCODER has none; its DECODER reads it
(ReadTransform vp8l.h:111-121, inverse transform.h:110-191).  Checked here:
  * the stream equals what the oracle's test-vector encoder writes for the SAME per-tile modes (so the residuals,
    the mode sub-image and the coded image are all as the dialect prescribes),
  * the oracle's decoder (pinned against the reference's inverse transforms) and the GPU decoder give back the input,
  * natural images get smaller."""
import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def encode(img, w, h, bits):
    import torch
    enc = webp_amd.Encoder()
    enc.set_predictor_bits(bits)
    d_in = torch.from_numpy(np.ascontiguousarray(img).view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0
    stream = bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy())
    pm = enc.fetch_stage(0, webp_amd.STAGE_PRED_MODES)
    info = enc.fetch_stage(0, webp_amd.STAGE_INFO)
    enc.close()
    return stream, pm, info


@pytest.mark.parametrize("mode,w,h,bits", [(3, 64, 64, 2), (3, 200, 50, 3), (4, 333, 77, 4), (3, 61, 19, 2), (3, 1, 40, 2), (3, 40, 1, 5),
                                            (3, 512, 512, 5), (4, 700, 300, 2), (3, 1030, 9, 9)])
def test_stream_equals_oracle_for_the_same_modes(mode, w, h, bits):
    img = O.gen_image(mode, w, h, 31 + w, 200)
    stream, pm, info = encode(img, w, h, bits)
    if info["ncolors"]:                      # few colours: colour indexing, which excludes the predictor
        rc, ref = O.encode(img, w, h)
        assert pm is None and rc == 0 and stream == ref
        return
    assert pm is not None
    assert pm["bits"] >= bits and pm["tx"] * pm["ty"] <= 1024
    assert pm["tx"] == (w + (1 << pm["bits"]) - 1) >> pm["bits"] and pm["ty"] == (h + (1 << pm["bits"]) - 1) >> pm["bits"]
    assert ((pm["modes"] >> 8) & 0xff).max() <= 13
    rc, ref = O.encode_ext(img, w, h, use_subtract_green=1, predictor_bits=pm["bits"], predictor_modes=pm["modes"], transform_order=0)
    assert rc == 0
    assert stream == ref
    rc, back, bw, bh = O.decode(stream)
    assert rc == 0 and (bw, bh) == (w, h) and np.array_equal(back, img)
    got, gw, gh = webp_amd.decode(stream)
    assert (gw, gh) == (w, h) and np.array_equal(got, img)


def test_palette_images_are_left_alone():
    img = O.gen_image(5, 100, 80, 3, 40)
    stream, pm, info = encode(img, 100, 80, 4)
    assert pm is None and info["ncolors"] == 40
    rc, ref = O.encode(img, 100, 80)
    assert rc == 0 and stream == ref


def test_natural_image_gets_smaller():
    # a smooth image with a little noise: the predictor removes the gradients
    rng = np.random.default_rng(8)
    w, h = 640, 360
    y, x = np.mgrid[0:h, 0:w]
    r = (x * 255 // w + rng.integers(0, 3, size=(h, w))).clip(0, 255)
    g = (y * 255 // h + rng.integers(0, 3, size=(h, w))).clip(0, 255)
    b = ((x + y) * 255 // (w + h) + rng.integers(0, 3, size=(h, w))).clip(0, 255)
    img = (np.uint32(0xFF000000) | (r.astype(np.uint32) << 16) | (g.astype(np.uint32) << 8) | b.astype(np.uint32)).reshape(-1)
    plain, _, _ = encode(img, w, h, 0)
    pred, pm, _ = encode(img, w, h, 4)
    assert len(pred) < 0.8 * len(plain), (len(pred), len(plain))
    got, gw, gh = webp_amd.decode(pred)
    assert np.array_equal(got, img)


def test_full_size_round_trip_and_size():
    import torch
    w, h = 3840, 2160
    d_in = torch.empty((2, w * h), dtype=torch.int32, device="cuda")
    webp_amd.generate_device(3, w, h, 12345, 0, d_in[0].data_ptr())
    webp_amd.generate_device(4, w, h, 12348, 0, d_in[1].data_ptr())
    enc = webp_amd.Encoder()
    cap = enc.bound(w, h)
    d_out = torch.empty(2 * cap, dtype=torch.uint8, device="cuda")
    sizes = {}
    for bits in (0, 5):
        enc.set_predictor_bits(bits)
        off, lens, st = enc.encode_device([d_in[k].data_ptr() for k in range(2)], [w] * 2, [h] * 2, d_out.data_ptr(), d_out.numel())
        assert (st == 0).all()
        sizes[bits] = [int(x) for x in lens]
        dec = webp_amd.Decoder()
        d_dec = torch.zeros((2, w * h), dtype=torch.int32, device="cuda")
        stt = dec.decode_device([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], [w] * 2, [h] * 2,
                                [d_dec[k].data_ptr() for k in range(2)])
        assert (stt == 0).all() and torch.equal(d_dec, d_in)
        dec.close()
    pm = enc.fetch_stage(0, webp_amd.STAGE_PRED_MODES)
    assert pm["bits"] == 7 and pm["tx"] * pm["ty"] == 30 * 17      # 5 asked, 7 is the first with <= 1024 tiles
    enc.close()
    print("bytes without / with predictor:", sizes)


def test_one_shot_entry_point_with_flags():
    img = O.gen_image(3, 96, 64, 5, 200)
    plain = webp_amd.encode(img, 96, 64)
    rc, ref = O.encode(img, 96, 64)
    assert rc == 0 and plain == ref
    small = webp_amd.encode(img, 96, 64, predictor_bits=3)
    assert small != plain
    got, w, h = webp_amd.decode(small)
    assert (w, h) == (96, 64) and np.array_equal(got, img)
    assert webp_amd.encode(img, 96, 64) == ref           # the plain context is untouched by the flagged call
    for bad in (1, 10, 16):
        with pytest.raises(webp_amd.Vp8lError):
            webp_amd.encode(img, 96, 64, predictor_bits=bad)


def test_streams_the_reference_decoder_read():
    """tests/golden/gpu_ext_streams.json (tools/pin_gpu_streams.py): sha256 of the streams this encoder wrote for a fixed
    set of images and options (predictor, cross-colour, colour cache, meta codes, alone and together) which the REFERENCE's
    own compiled decoder (vp8l.h:330-345) turned back into the images.  The encoder must still write exactly those."""
    import hashlib
    import json
    import os
    cases = json.load(open(os.path.join(O.ROOT, "tests", "golden", "gpu_ext_streams.json")))
    assert len(cases) >= 50
    for ent in cases:
        img, w, h = O.case_image(ent["image"])
        data = webp_amd.encode(img, w, h, **ent["options"])
        assert len(data) == ent["bytes"] and hashlib.sha256(data).hexdigest() == ent["sha256"], (ent["image"], ent["options"])
