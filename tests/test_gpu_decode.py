"""GPU parity of the decoder through the C ABI: pixels must equal the oracle's for the encoder's
own streams and for streams that use the decoder-only features (predictor, cross-colour, colour
cache, meta-Huffman); damaged streams must get the oracle's verdict."""
import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def gpu_decode_all(streams):
    return webp_amd.decode_batch(streams)


def test_roundtrip_own_streams():
    cases = [(3, 64, 64), (4, 64, 64), (5, 64, 64), (5, 61, 19), (3, 1, 1), (4, 2, 1), (3, 300, 200), (4, 512, 512), (5, 512, 512),
             (3, 1031, 17), (4, 16384, 1), (3, 1, 500)]
    imgs = [O.gen_image(m, w, h, 31 + w, 200) for m, w, h in cases]
    streams = [O.encode(img, w, h)[1] for img, (m, w, h) in zip(imgs, cases)]
    res = gpu_decode_all(streams)
    for (st, arr, w, h), img, (m, cw, ch) in zip(res, imgs, cases):
        assert st == 0 and (w, h) == (cw, ch)
        assert np.array_equal(arr, img), (m, cw, ch)


def test_palette_bundling_variants():
    rng = np.random.default_rng(8)
    streams, imgs = [], []
    for ncol in (1, 2, 3, 4, 5, 16, 17, 256):
        pal = rng.choice(1 << 32, size=ncol, replace=False).astype(np.uint32)
        for (w, h) in ((9, 4), (63, 17), (64, 64), (1, 7)):
            img = pal[rng.integers(0, ncol, size=w * h)]
            rc, s = O.encode(img, w, h)
            assert rc == 0
            streams.append(s)
            imgs.append(img)
    for (st, arr, w, h), img in zip(gpu_decode_all(streams), imgs):
        assert st == 0 and np.array_equal(arr, img)


EXT = [
    dict(use_subtract_green=1), dict(predictor_bits=2), dict(predictor_bits=4, use_subtract_green=1),
    dict(color_bits=3), dict(predictor_bits=3, color_bits=2, use_subtract_green=1, transform_order=1),
    dict(predictor_bits=2, color_bits=5, use_subtract_green=1, transform_order=2),
    dict(predictor_bits=2, color_bits=2, transform_order=3), dict(cache_bits=4),
    dict(cache_bits=11, use_subtract_green=1), dict(meta_bits=2, meta_groups=3),
    dict(meta_bits=3, meta_groups=7, cache_bits=6, predictor_bits=3, use_subtract_green=1),
    dict(use_palette=1, cache_bits=3), dict(use_palette=1, meta_bits=2, meta_groups=2),
    dict(meta_bits=2, meta_groups=40, cache_bits=1),
]


def test_decoder_only_features():
    streams, imgs = [], []
    for prm in EXT:
        for (w, h) in ((1, 1), (7, 5), (33, 18), (64, 64), (130, 9), (300, 1100)):
            for mode in (3, 4, 5, "rand"):
                if w * h > 100000 and mode != 4:
                    continue
                img = (np.random.default_rng(1).integers(0, 1 << 32, size=w * h, dtype=np.uint32) if mode == "rand"
                       else O.gen_image(mode, w, h, 5, 200))
                rc, s = O.encode_ext(img, w, h, **prm)
                assert rc == 0
                streams.append(s)
                imgs.append(img)
    res = gpu_decode_all(streams)
    for k, ((st, arr, w, h), img) in enumerate(zip(res, imgs)):
        assert st == 0, (k, st)
        assert np.array_equal(arr, img), k


def test_every_predictor_mode_and_invalid_mode():
    w, h = 37, 29
    img = np.random.default_rng(3).integers(0, 1 << 32, size=w * h, dtype=np.uint32)
    streams = []
    for mode in range(15):
        tiles = np.full(((w + 31) // 32) * ((h + 31) // 32), 0xFF000000 | (mode << 8), dtype=np.uint32)
        rc, s = O.encode_ext(img, w, h, predictor_bits=5, predictor_modes=tiles)
        assert rc == 0
        streams.append(s)
    res = gpu_decode_all(streams)
    for mode, (st, arr, _, _) in enumerate(res):
        rc0, ref, _, _ = O.decode(streams[mode])
        assert st == rc0
        if mode <= 13:
            assert st == 0 and np.array_equal(arr, ref) and np.array_equal(arr, img)
        else:
            assert st == -5


def test_corrupt_streams_get_the_oracle_verdict():
    rng = np.random.default_rng(77)
    img = O.gen_image(4, 40, 30, 3, 200)
    bases = [O.encode(img, 40, 30)[1],
             O.encode_ext(img, 40, 30, predictor_bits=2, cache_bits=5, meta_bits=2, meta_groups=3)[1],
             O.encode(O.gen_image(5, 40, 30, 3, 10), 40, 30)[1]]
    streams = [b"", b"\0\0\0\0\x2f", b"\0\0\0\0\x2e\0\0\0\0\0"]
    for base in bases:
        for it in range(100):
            b = bytearray(base)
            if it % 3 == 0:
                b = b[: int(rng.integers(9, len(b)))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    pos = int(rng.integers(9, len(b)))
                    b[pos] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(b))
    res = gpu_decode_all(streams)
    accepted = 0
    for s, (st, arr, w, h) in zip(streams, res):
        rc0, ref, w0, h0 = O.decode(s)
        assert st == rc0, (st, rc0, len(s))
        if rc0 == 0:
            assert np.array_equal(arr, ref)
            accepted += 1
    assert accepted > 20


def test_corrupt_colour_cache_streams_get_the_oracle_verdict():
    # The workgroup form resolves cache tokens from keys alone; a damaged stream may ask for a slot that was never written
    # (the one case where a pixel's key is not its token's index), chain cache tokens through copies, or cut them short.
    rng = np.random.default_rng(4242)
    bases = []
    for (w, h, mode, prm) in ((130, 90, 4, dict(cache_bits=4)), (130, 90, 4, dict(cache_bits=11, use_subtract_green=1)),
                              (97, 61, 5, dict(cache_bits=2)), (64, 64, 3, dict(cache_bits=7, meta_bits=3, meta_groups=4)),
                              (200, 40, 4, dict(cache_bits=1, predictor_bits=3))):
        img = O.gen_image(mode, w, h, 9, 40)
        rc, s = O.encode_ext(img, w, h, **prm)
        assert rc == 0
        bases.append(s)
    streams = list(bases)
    for base in bases:
        for it in range(120):
            b = bytearray(base)
            if it % 4 == 0:
                b = b[: int(rng.integers(9, len(b)))]
            else:
                for _ in range(int(rng.integers(1, 3))):
                    pos = int(rng.integers(len(b) // 4, len(b)))   # mostly inside the pixel data
                    b[pos] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(b))
    res = gpu_decode_all(streams)
    accepted = 0
    for s, (st, arr, w, h) in zip(streams, res):
        rc0, ref, w0, h0 = O.decode(s)
        assert st == rc0, (st, rc0, len(s))
        if rc0 == 0:
            assert np.array_equal(arr, ref)
            accepted += 1
    assert accepted > 50


def test_cache_tokens_with_unrelated_indices():
    # Test vectors in which cache tokens ask for arbitrary slots, written or not: what they decode to is defined (an unwritten
    # slot holds 0, and that pixel is then stored under the key of 0, not under the index asked for); the oracle says what.
    streams, want = [], []
    for (w, h, mode) in ((130, 90, 4), (64, 64, 3), (300, 200, 4), (97, 61, 5)):
        img = O.gen_image(mode, w, h, 21, 60)
        for prm in (dict(cache_bits=1), dict(cache_bits=4), dict(cache_bits=11), dict(cache_bits=6, meta_bits=3, meta_groups=3),
                    dict(cache_bits=5, predictor_bits=3, use_subtract_green=1)):
            for every in (1, 2, 5, 17):
                rc, s = O.encode_ext(img, w, h, stray_cache=every, **prm)
                assert rc == 0
                rc0, ref, _, _ = O.decode(s)
                assert rc0 == 0
                streams.append(s)
                want.append(ref)
    res = gpu_decode_all(streams)
    for k, ((st, arr, w, h), ref) in enumerate(zip(res, want)):
        assert st == 0, (k, st)
        assert np.array_equal(arr, ref), k


def test_encode_then_decode_on_device():
    import torch
    w, h = 640, 360
    d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
    webp_amd.generate_device(3, w, h, 77, 0, d_in.data_ptr())
    enc, dec = webp_amd.Encoder(), webp_amd.Decoder()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, status = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert status[0] == 0
    d_img = torch.empty(w * h, dtype=torch.int32, device="cuda")
    st = dec.decode_device([d_out.data_ptr() + int(off[0])], [int(lens[0])], [w], [h], [d_img.data_ptr()])
    assert st[0] == 0
    assert torch.equal(d_img, d_in)


def test_hostile_group_count_inside_a_batch_of_valid_streams():
    """A 40-byte stream whose meta-Huffman image asks for 65536 Huffman groups (or a few hundred) must get its own verdict
    (the oracle's: the groups lie past the end of the stream) without costing the valid streams of the call theirs, and
    without sizing any workspace by the count.  Streams that really hold many groups still decode."""
    from test_host_cores import hostile_group_count_stream
    imgs = [O.gen_image(m, 40, 30, 5 + m, 200) for m in (3, 4, 5)]
    good = [O.encode(img, 40, 30)[1] for img in imgs]
    many = O.gen_image(3, 64, 64, 9, 200)
    rc, many_s = O.encode_ext(many, 64, 64, meta_bits=2, meta_groups=40, cache_bits=1)
    assert rc == 0
    bad = [hostile_group_count_stream(ix) for ix in (0xFFFF, 300, 5)]
    streams = [good[0], bad[0], good[1], many_s, bad[1], good[2], bad[2]]
    res = gpu_decode_all(streams)
    for k, ref in ((0, imgs[0]), (2, imgs[1]), (5, imgs[2]), (3, many)):
        assert res[k][0] == 0 and np.array_equal(res[k][1], ref)
    for k in (1, 4, 6):
        assert res[k][0] == O.decode(streams[k])[0] == -6
