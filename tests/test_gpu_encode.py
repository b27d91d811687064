"""GPU parity of the encoder, called through the C ABI: stream bytes and every intermediate
plane (residual pixels, per-position matches on the parse, histograms, code lengths) must equal
the oracle's bit for bit."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def check_stream(img, w, h):
    rc, ref = O.encode(img, w, h)
    if rc != 0:
        with pytest.raises(webp_amd.Vp8lError) as ei:
            webp_amd.encode(img, w, h)
        assert ei.value.status == rc
        return None
    got = webp_amd.encode(img, w, h)
    assert len(got) == len(ref)
    assert got == ref
    return got


def test_known_answer_streams(golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "kat_streams.json")))
    for k in kats:
        img = O.gen_image(k["mode"], k["w"], k["h"], k["seed"], 200)
        got = webp_amd.encode(img, k["w"], k["h"])
        assert len(got) == k["bytes"], k
        assert hashlib.sha256(got).hexdigest().startswith(k["sha256_16"]), k


@pytest.mark.parametrize("mode,w,h", [(3, 64, 64), (4, 64, 64), (5, 64, 64), (5, 61, 19), (3, 200, 50), (4, 333, 77)])
def test_stage_parity(mode, w, h):
    torch = torch_cuda()
    img = O.gen_image(mode, w, h, 4242, 200)
    rc, ref, tr = O.encode(img, w, h, trace=True)
    assert rc == 0
    enc = webp_amd.Encoder()
    d_in = torch.from_numpy(img.view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, status = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert status[0] == 0
    info = enc.fetch_stage(0, webp_amd.STAGE_INFO)
    assert info["ncolors"] == tr["ncolors"] and info["xsize"] == tr["xsize"] and info["M"] == tr["xsize"] * tr["ysize"]
    assert np.array_equal(enc.fetch_stage(0, webp_amd.STAGE_RESIDUAL), tr["residual"])
    m = enc.fetch_stage(0, webp_amd.STAGE_MATCH)
    # per-position best match against the oracle at EVERY position
    ln, ds = O.lz77_best_all(tr["residual"])
    assert np.array_equal(m & 0xFF, ln)
    assert np.array_equal(np.where(ln > 0, (m >> 8) & 0x7FF, 0), ds)
    # parse flags = token start positions
    starts = np.zeros(info["M"], dtype=bool)
    p = 0
    for d, l, s in tr["tokens"]:
        starts[p] = True
        p += l if l else 1
    assert np.array_equal((m >> 31).astype(bool), starts)
    h5 = enc.fetch_stage(0, webp_amd.STAGE_HISTO).reshape(5, 280)
    cl = enc.fetch_stage(0, webp_amd.STAGE_CODELEN).reshape(5, 280)
    for c in range(5):
        a = O.ALPHABET[c]
        assert np.array_equal(h5[c, :a], tr["histo"][c]), c
        assert np.array_equal(cl[c, :a], tr["lengths"][c]), c
    got = bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy())
    assert got == ref
    enc.close()


SHAPES = [(1, 1), (2, 1), (1, 2), (3, 3), (16, 1), (17, 1), (1, 300), (127, 1), (128, 1), (129, 1), (130, 2),
          (1023, 1), (1024, 1), (1025, 1), (2047, 1), (2048, 1), (2049, 1), (4095, 1), (4097, 1), (16383, 1), (16384, 1),
          (127, 129), (129, 127), (640, 26), (1031, 17)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_edge_shapes_bytes_equal(w, h):
    for mode in (3, 4):
        check_stream(O.gen_image(mode, w, h, 7 + w, 200), w, h)


def test_random_content_small():
    rng = np.random.default_rng(99)
    for it in range(60):
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 60))
        alpha = int(rng.choice([1, 2, 3, 40, 300, 1 << 20]))
        img = (rng.integers(0, alpha, size=w * h).astype(np.uint32) * np.uint32(0x01020301)) ^ np.uint32(0xFF000000)
        check_stream(img, w, h)


@pytest.mark.parametrize("ncol", [1, 2, 3, 4, 5, 16, 17, 200, 256, 257])
def test_palette_paths(ncol):
    rng = np.random.default_rng(ncol)
    pal = rng.choice(1 << 32, size=ncol, replace=False).astype(np.uint32)
    for (w, h) in ((9, 4), (63, 17), (64, 64), (1, 7), (2050, 3)):
        img = pal[rng.integers(0, ncol, size=w * h)]
        check_stream(img, w, h)
        # generator-style palette image (long matches)
        check_stream(O.gen_image(5, w, h, 1, min(ncol, 256)), w, h)


def test_long_matches_and_window_edges():
    rng = np.random.default_rng(5)
    n = 6000
    a = rng.integers(0, 1 << 24, size=n, dtype=np.uint32) | np.uint32(0xFF000000)
    a[700:1000] = a[100:400]
    a[1500:1900] = a[1499]
    a[3000:3300] = a[1976:2276]   # distance exactly 1024
    a[4000:4300] = a[2975:3275]   # distance 1025: out of window
    a[5000:5800] = np.tile(a[4990:5000], 80)
    for (w, h) in ((n, 1), (100, 60), (60, 100)):
        check_stream(a, w, h)
    check_stream(np.full(5000, 0xFF123456, dtype=np.uint32), 5000, 1)
    check_stream(np.full(4096 * 3, 0x00000000, dtype=np.uint32), 4096, 3)


def test_batch_mixed_sizes_and_errors():
    imgs = [(O.gen_image(3, 64, 64, 1), 64, 64), (O.gen_image(5, 61, 19, 2, 4), 61, 19), (np.zeros(0, np.uint32), 0, 5),
            (O.gen_image(4, 300, 200, 3), 300, 200), (O.gen_image(5, 50, 50, 4, 200), 50, 50), (O.gen_image(3, 1, 1, 5), 1, 1)]
    res = webp_amd.encode_batch(imgs)
    for (img, w, h), (st, data) in zip(imgs, res):
        rc, ref = O.encode(img, w, h)
        assert st == rc
        if rc == 0:
            assert data == ref


def test_argument_errors():
    with pytest.raises(webp_amd.Vp8lError) as e:
        webp_amd.encode(np.zeros(0, np.uint32), 0, 0)
    assert e.value.name == "InvalidARGBImage"
    with pytest.raises(webp_amd.Vp8lError) as e:
        webp_amd.encode(np.zeros(16385, np.uint32), 16385, 1)
    assert e.value.name == "TooBigARGBImage"


def test_device_generator_matches_oracle():
    torch = torch_cuda()
    for mode, w, h, nc in ((3, 257, 31, 0), (4, 300, 100, 0), (5, 123, 45, 200), (5, 64, 64, 4), (3, 1, 1, 0), (3, 1, 9, 0)):
        d = torch.empty(w * h, dtype=torch.int32, device="cuda")
        webp_amd.generate_device(mode, w, h, 12345 + mode, nc, d.data_ptr())
        got = d.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, O.gen_image(mode, w, h, 12345 + mode, nc if nc else 200)), (mode, w, h)


def test_medium_images_bytes_equal():
    for mode, w, h in ((3, 512, 512), (4, 512, 512), (5, 512, 512), (3, 1920, 120), (4, 777, 333)):
        check_stream(O.gen_image(mode, w, h, 12345, 200), w, h)


def test_unit_longer_than_the_staging_window():
    # 6144 consecutive tokens whose codes are long (symbols that are rare in every channel): more stream
    # bits than the emit kernel's 131072-bit staging window, so a unit is written in several windows
    rng = np.random.default_rng(77)
    w, h = 4096, 64
    c = rng.integers(0, 16, size=(w * h, 3)).astype(np.uint32)
    base = np.uint32(0xFF000000) | (c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]
    img = base.copy()
    rare = rng.integers(16, 256, size=(6144, 4)).astype(np.uint32)
    img[20 * w: 20 * w + 6144] = (rare[:, 0] << 24) | (rare[:, 1] << 16) | (rare[:, 2] << 8) | rare[:, 3]
    plain = check_stream(base, w, h)
    got = check_stream(img, w, h)
    assert 8 * (len(got) - len(plain)) > 6144 * 36   # the stretch costs well over 32 bits a pixel
    dec, dw, dh = webp_amd.decode(got)
    assert (dw, dh) == (w, h) and np.array_equal(dec, img)


def test_pixels_not_16_byte_aligned():
    # the match kernel reads the caller's pixels 16 bytes at a time when it can; a pointer that is only
    # 4-byte aligned takes the separate transform kernel.  Same bytes either way.
    torch = torch_cuda()
    w, h = 333, 41
    img = O.gen_image(3, w, h, 77, 200)
    rc, ref = O.encode(img, w, h)
    assert rc == 0
    enc = webp_amd.Encoder()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for shift in (0, 1, 2, 3):
        buf = torch.zeros(w * h + 8, dtype=torch.int32, device="cuda")
        buf[shift: shift + w * h] = torch.from_numpy(img.view(np.int32)).cuda()
        off, lens, status = enc.encode_device([buf.data_ptr() + 4 * shift], [w], [h], d_out.data_ptr(), cap)
        assert status[0] == 0
        assert bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy()) == ref, shift
        assert np.array_equal(enc.fetch_stage(0, webp_amd.STAGE_RESIDUAL), O.encode(img, w, h, trace=True)[2]["residual"])
    enc.close()


def test_palette_scan_edges():
    # The palette scan reads 16 bytes a thread where the caller's pixels are aligned so and falls back to single words
    # at the end of an image and for unaligned pointers; colours that show only in the last pixels, in the last partial
    # group of four, or first beyond 256 must be seen all the same.  Images of several 16384-pixel superblocks.
    torch = torch_cuda()
    rng = np.random.default_rng(2026)
    enc = webp_amd.Encoder()
    cases = []
    for (w, h, ncol) in ((201, 199, 16), (130, 127, 256), (257, 130, 3), (181, 183, 200)):
        pal = rng.choice(1 << 32, size=ncol + 2, replace=False).astype(np.uint32)
        base = pal[rng.integers(0, max(1, ncol - 1), size=w * h)]
        a = base.copy(); a[-1] = pal[ncol - 1]                     # the last colour of the palette only in the last pixel
        b = a.copy(); b[-2] = pal[ncol]; b[-3] = pal[ncol + 1]     # two more at the very end (beyond 256 for ncol = 256)
        c = base.copy(); c[16384 + 5] = pal[ncol - 1]              # ... or once in the second superblock
        cases += [(a, w, h), (b, w, h), (c, w, h)]
    for img, w, h in cases:
        rc, ref = O.encode(img, w, h)
        assert rc == 0
        cap = enc.bound(w, h)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for shift in (0, 1, 3):
            buf = torch.zeros(w * h + 8, dtype=torch.int32, device="cuda")
            buf[shift: shift + w * h] = torch.from_numpy(img.view(np.int32)).cuda()
            off, lens, status = enc.encode_device([buf.data_ptr() + 4 * shift], [w], [h], d_out.data_ptr(), cap)
            assert status[0] == 0
            assert bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy()) == ref, (w, h, shift)
    enc.close()
