"""The standard VP8L dialect on the GPU, through the C ABI.  Decoding: libwebp's own lossless files (committed fixtures, and
larger ones made here when Pillow is importable) must give libwebp's pixels in every form of the decoder.  Encoding
(VP8L_FLAG_STANDARD / VP8L_OPT_STANDARD): libwebp must decode the streams to the input, also with the predictor and cross-colour
options (whose last-column rule differs between the dialects), and this decoder must read them back."""
import os

import numpy as np
import pytest
import torch

import _oracle as O
import _webpstd as S
import webp_amd
from test_gpu_decode_forms import decode_with

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libwebp_streams.npz")
needs_pillow = pytest.mark.skipif(S.pillow() is None, reason="Pillow with WebP support is not importable")


def test_committed_libwebp_streams_on_the_gpu():
    z = np.load(GOLDEN)
    names = sorted(set(k.rsplit("/", 1)[0] for k in z.files))
    streams = [z[n + "/stream"].tobytes() for n in names]
    res = webp_amd.decode_batch(streams)
    for n, (st, arr, w, h) in zip(names, res):
        assert st == 0 and [w, h] == list(z[n + "/size"]), n
        assert np.array_equal(arr, z[n + "/argb"]), n
    for threads, whole in ((64, 1), (1024, 1), (1024, 2)):
        for n, (st, arr) in zip(names, decode_with(threads, streams, whole_device=whole)):
            assert st == 0 and np.array_equal(arr, z[n + "/argb"]), (n, threads, whole)


def big_images(rng):
    w, h = 1024, 700
    y, x = np.mgrid[0:h, 0:w]
    photo = np.stack([(x * 160 // w + y * 80 // h + rng.integers(-3, 4, size=x.shape)) & 255,
                      (x * 150 // w + y * 90 // h + rng.integers(-3, 4, size=x.shape)) & 255,
                      (x * 100 // w + y * 60 // h + rng.integers(-3, 4, size=x.shape)) & 255, np.full_like(x, 255)], axis=-1).astype(np.uint8)
    yield "photo_like", S.rgba_to_argb(photo), w, h
    yield "graphics", O.gen_image(4, w, h, 9, 0), w, h
    yield "palette", O.gen_image(5, w, h, 9, 40), w, h
    yield "noise", S.rgba_to_argb(rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)), w, h


@needs_pillow
def test_large_libwebp_files_in_every_decoder_form():
    rng = np.random.default_rng(3)
    streams, want = [], []
    for name, argb, w, h in big_images(rng):
        for opts in (dict(quality=20, method=1), dict(quality=90, method=4)):
            f = S.libwebp_encode(argb, w, h, **opts)
            px, w0, h0 = S.libwebp_decode(f)
            assert np.array_equal(px, argb)
            streams.append(S.vp8l_payload(f))
            want.append(px)
    for threads, whole in ((1024, 1), (1024, 2), (64, 1)):
        for k, (st, arr) in enumerate(decode_with(threads, streams, whole_device=whole)):
            assert st == 0 and np.array_equal(arr, want[k]), (k, threads, whole)


@needs_pillow
def test_standard_streams_from_the_gpu_encoder_decode_in_libwebp():
    rng = np.random.default_rng(4)
    cases = [(argb, w, h) for _, argb, w, h in S.test_images(rng)]
    cases += [(O.gen_image(m, w, h, 77, nc), w, h) for m, w, h, nc in ((3, 640, 360, 0), (4, 333, 500, 0), (5, 256, 256, 200), (5, 211, 97, 13),
                                                                         (5, 640, 100, 2), (3, 1, 300, 0), (4, 2000, 3, 0))]
    for argb, w, h in cases:
        ref = webp_amd.encode(argb, w, h)
        s = webp_amd.encode(argb, w, h, standard=True)
        assert int.from_bytes(s[:4], "little") == len(s) - 4 and s[4] == 0x2F
        assert int.from_bytes(ref[:4], "little") == 0                 # the default stays the reference's dialect
        got, w1, h1 = S.libwebp_decode(webp_amd.riff_container(s))
        assert (w1, h1) == (w, h) and np.array_equal(got, argb)
        back, w2, h2 = webp_amd.decode(s)
        assert (w2, h2) == (w, h) and np.array_equal(back, argb)


@needs_pillow
@pytest.mark.parametrize("pred,cross", [(4, 0), (0, 3), (3, 5), (9, 2)])
def test_standard_streams_with_predictor_and_cross_colour(pred, cross):
    rng = np.random.default_rng(5)
    for name, argb, w, h in list(big_images(rng))[:2] + [("small", O.gen_image(3, 97, 61, 3, 0), 97, 61), ("column", O.gen_image(3, 2, 200, 3, 0), 2, 200)]:
        s = webp_amd.encode(argb, w, h, predictor_bits=pred, cross_color_bits=cross, standard=True)
        got, w1, h1 = S.libwebp_decode(webp_amd.riff_container(s))
        assert (w1, h1) == (w, h) and np.array_equal(got, argb), name
        back, _, _ = webp_amd.decode(s)
        assert np.array_equal(back, argb), name
        # the same options in the reference's dialect still round-trip here (last column: top-right := left)
        d = webp_amd.encode(argb, w, h, predictor_bits=pred, cross_color_bits=cross)
        assert np.array_equal(webp_amd.decode(d)[0], argb), name


@needs_pillow
def test_standard_batch_on_the_device_and_4k():
    w, h = 3840, 2160
    d_in = torch.empty((2, w * h), dtype=torch.int32, device="cuda")
    webp_amd.generate_device(3, w, h, 12345, 0, d_in[0].data_ptr())
    webp_amd.generate_device(4, w, h, 12348, 0, d_in[1].data_ptr())
    enc = webp_amd.Encoder()
    enc.set_standard(True)
    cap = enc.bound(w, h)
    d_out = torch.empty(2 * cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in[0].data_ptr(), d_in[1].data_ptr()], [w, w], [h, h], d_out.data_ptr(), d_out.numel())
    assert (st == 0).all()
    for k in range(2):
        s = bytes(d_out[int(off[k]):int(off[k]) + int(lens[k])].cpu().numpy())
        got, w1, h1 = S.libwebp_decode(webp_amd.riff_container(s))
        assert (w1, h1) == (w, h)
        assert np.array_equal(got, d_in[k].cpu().numpy().view(np.uint32))   # 8 Mpixel images: codes would exceed 15 bits unlimited
