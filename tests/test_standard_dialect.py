"""The standard VP8L dialect on the CPU build of the sequential cores, with libwebp (through Pillow) as the checker:
libwebp's own lossless files must decode to libwebp's pixels, and streams written in the standard dialect must decode in libwebp
to the input.  Committed fixtures (tests/golden/libwebp_streams.npz, made by tests/golden/make_libwebp_golden.py) repeat the
first half where Pillow is missing."""
import os

import numpy as np
import pytest

import _oracle as O
import _webpstd as S
from test_host_cores import K, core_decode  # noqa: F401  (K is a fixture)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libwebp_streams.npz")


def test_committed_libwebp_streams_decode_to_libwebps_pixels(K):
    z = np.load(GOLDEN)
    names = sorted(set(k.rsplit("/", 1)[0] for k in z.files))
    assert len(names) >= 20
    for name in names:
        stream, want, (w, h) = z[name + "/stream"].tobytes(), z[name + "/argb"], z[name + "/size"]
        rc, got, w1, h1 = core_decode(K, stream, int(w), int(h))
        assert rc == 0 and (w1, h1) == (w, h), name
        assert np.array_equal(got, want), name


@pytest.mark.skipif(S.pillow() is None, reason="Pillow with WebP support is not importable")
def test_libwebp_files_made_here_decode_to_libwebps_pixels(K):
    rng = np.random.default_rng(11)
    for name, argb, w, h in S.test_images(rng):
        for opts in (dict(quality=0, method=0), dict(quality=50, method=3), dict(quality=100, method=6)):
            f = S.libwebp_encode(argb, w, h, **opts)
            want, w0, h0 = S.libwebp_decode(f)
            assert (w0, h0) == (w, h) and np.array_equal(want, argb)
            rc, got, w1, h1 = core_decode(K, S.vp8l_payload(f), w, h)
            assert rc == 0 and (w1, h1) == (w, h), (name, opts)
            assert np.array_equal(got, want), (name, opts)


def core_encode_std(K, img, w, h):
    import ctypes as C
    K.core_encode_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_int]
    K.core_encode_ex.restype = C.c_long
    cap = w * h * 5 + 65536
    buf = np.zeros(cap, dtype=np.uint8)
    n = K.core_encode_ex(np.ascontiguousarray(img, dtype=np.uint32).ctypes.data, w, h, buf.ctypes.data, cap, 1)
    assert n > 0
    return buf[:n].tobytes()


@pytest.mark.skipif(S.pillow() is None, reason="Pillow with WebP support is not importable")
def test_standard_streams_of_the_encoder_core_decode_in_libwebp(K):
    rng = np.random.default_rng(12)
    cases = [(argb, w, h) for _, argb, w, h in S.test_images(rng)]
    cases += [(O.gen_image(m, w, h, 5, 40), w, h) for m, w, h in ((3, 64, 64), (4, 130, 40), (5, 61, 19), (3, 400, 300))]
    for argb, w, h in cases:
        s = core_encode_std(K, argb, w, h)
        assert int.from_bytes(s[:4], "little") == len(s) - 4 and s[4] == 0x2F
        got, w1, h1 = S.libwebp_decode(S.riff(s))
        assert (w1, h1) == (w, h) and np.array_equal(got, argb)
        rc, mine, _, _ = core_decode(K, s, w, h)       # and the same stream through this decoder (dialect told by the size field)
        assert rc == 0 and np.array_equal(mine, argb)


def test_skewed_histograms_stay_within_15_bits(K):
    """An image whose symbol counts follow a Fibonacci-like ladder would get codes far longer than 15 bits."""
    counts = [1, 1]
    while len(counts) < 40:
        counts.append(counts[-1] + counts[-2])
    vals = np.repeat(np.arange(len(counts), dtype=np.uint32), np.minimum(counts, 20000))
    rng = np.random.default_rng(2)
    rng.shuffle(vals)
    w = 251
    h = len(vals) // w
    argb = (0xFF000000 | (vals[:w * h] << 8) | ((vals[:w * h] * 3) & 0xFF)).astype(np.uint32)
    s = core_encode_std(K, argb, w, h)
    rc, mine, _, _ = core_decode(K, s, w, h)
    assert rc == 0 and np.array_equal(mine, argb)
    if S.pillow() is not None:
        got, _, _ = S.libwebp_decode(S.riff(s))
        assert np.array_equal(got, argb)
