"""The two forms of the entropy decoder (a workgroup per stream, a wave per stream) against each other and
against the oracle, on the encoder's streams and on hand-written streams with tokens the encoders never
produce: references longer than 255 and 1024 pixels, overlapping runs, distances beyond the LDS ring."""
import numpy as np
import pytest
import torch

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def decode_with(threads, streams, whole_device=1, used=None, scout_form=0, shift=0):
    """Decodes through the device-memory entry point with the given form; returns [(status, pixels)].  `shift`: the
    streams start this many bytes (a multiple of 4) into their allocations, which end right behind them."""
    dec = webp_amd.Decoder()
    dec.set_threads_per_stream(threads)
    dec.set_whole_device(whole_device)
    dec.set_scout_form(scout_form)
    infos = [webp_amd.get_info(s) for s in streams]
    d_streams = []
    for s in streams:
        buf = np.frombuffer(b"\xA5" * shift + s + b"\0" * ((-len(s)) % 4), dtype=np.uint8)
        d_streams.append(torch.from_numpy(buf.copy()).cuda())
    d_out = [torch.full((w * h,), 0x5A5A5A5A, dtype=torch.int32, device="cuda") for w, h in infos]
    st = dec.decode_device([d.data_ptr() + shift for d in d_streams], [len(s) for s in streams], [w for w, _ in infos],
                           [h for _, h in infos], [d.data_ptr() for d in d_out])
    torch.cuda.synchronize()
    if used is not None:
        used.append(dec.last_whole_device_streams())
        used.append(dec.last_scout_streams())
    return [(int(st[i]), d_out[i].cpu().numpy().view(np.uint32)) for i in range(len(streams))]


class Bits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):  # LSB first (bit_writer.h:60-64)
        for i in range(n):
            self.bits.append((v >> i) & 1)

    def code(self, c, n):  # a canonical code: its most significant bit comes first in the stream
        for i in range(n - 1, -1, -1):
            self.bits.append((c >> i) & 1)

    def bytes(self):
        b = self.bits + [0] * ((-len(self.bits)) % 8)
        return bytes(sum(b[i + k] << k for k in range(8)) for i in range(0, len(b), 8))


ORDER = [66, 67, 0, 1, 2, 3, 4, 5, 65] + list(range(6, 65))  # huffman_io.h:49-65


def write_flat_code(bw, alphabet, used, length):
    """A normal code (huffman_io.h:181-193) in which the symbols in `used` all have `length` bits and nothing is run-length coded."""
    assert len(used) == 1 << length
    ncl = ORDER.index(length) + 1
    bw.put(0, 1)
    bw.put(ncl - 4, 6)
    for i in range(ncl):
        bw.put(1 if ORDER[i] in (0, length) else 0, 6)  # code-length code: symbols 0 and `length`, one bit each
    bw.put(0, 1)
    uset = set(used)
    for s in range(alphabet):
        bw.code(1 if s in uset else 0, 1)


def write_single(bw, sym):  # simple code with one symbol: it takes no bits in the pixel data (huffman_io.h:152-178)
    bw.put(1, 1)
    bw.put(0, 1)
    bw.put(1, 1)
    bw.put(sym, 8)


def prefix_symbol(v):  # inverse of lz77.cpp:50-57
    for sym in range(40):
        if sym < 4:
            base, nb = sym + 1, 0
        else:
            nb = (sym - 2) >> 1
            base = ((2 + (sym & 1)) << nb) + 1
        if base <= v < base + (1 << nb):
            return sym, nb, v - base
    raise ValueError(v)


def handmade_stream(w, h, tokens):
    """tokens: ('lit', g) or ('ref', length, distance).  Returns (stream, expected pixels or None)."""
    green_used = list(range(232)) + list(range(256, 280))  # 256 symbols of 8 bits
    bw = Bits()
    bw.put(0, 32)
    bw.put(0x2F, 8)
    bw.put(w - 1, 14)
    bw.put(h - 1, 14)
    bw.put(0, 1)
    bw.put(0, 3)
    bw.put(0, 1)  # no transform
    bw.put(0, 1)  # no colour cache
    bw.put(0, 1)  # no meta-Huffman image
    write_flat_code(bw, 280, green_used, 8)
    write_single(bw, 0x10)  # red
    write_single(bw, 0x20)  # blue
    write_single(bw, 0xFF)  # alpha
    write_flat_code(bw, 40, list(range(32)), 5)
    rank = {s: i for i, s in enumerate(green_used)}
    out = []
    ok = True
    for t in tokens:
        if t[0] == "lit":
            bw.code(rank[t[1]], 8)
            out.append(0xFF100020 | (t[1] << 8))
        else:
            _, length, dist = t
            sym, nb, extra = prefix_symbol(length)
            bw.code(rank[256 + sym], 8)
            bw.put(extra, nb)
            sym, nb, extra = prefix_symbol(dist + 120)
            bw.code(sym, 5)
            bw.put(extra, nb)
            if dist > len(out) or len(out) + length > w * h:
                ok = False
                break
            for _ in range(length):
                out.append(out[-dist])
    return bw.bytes(), (np.array(out, dtype=np.uint32) if ok and len(out) == w * h else None)


def handmade_cases():
    rng = np.random.default_rng(5)
    cases = []

    def lits(n):
        return [("lit", int(g)) for g in rng.integers(0, 232, size=n)]

    w, h = 4096, 16  # large enough for the whole-device form
    total = w * h
    toks = lits(3000) + [("ref", 1000, 2900), ("ref", 4096, 1), ("ref", 200, 3000), ("ref", 255, 5)] + lits(70)
    toks += [("ref", 256, 8000), ("ref", 3, 2), ("ref", 2500, 2049), ("ref", 1, 1)] + lits(5) + [("ref", 128, 2048), ("ref", 128, 2047)]
    used = 3000 + 1000 + 4096 + 200 + 255 + 70 + 256 + 3 + 2500 + 1 + 5 + 256
    # a long stretch of short references reading each other, then the rest as literals
    k = 0
    while used + 40 < total - 3000:
        toks.append(("ref", 1 + k % 9, 1 + (k * 7) % 13))
        used += 1 + k % 9
        k += 1
    toks += lits(total - used)
    cases.append((w, h, toks))
    # the reference encoder's doubling runs, across many rounds
    toks = lits(1)
    n = 1
    while n < total:
        step = min(n, 128, total - n)
        toks.append(("ref", step, step))
        n += step
    cases.append((w, h, toks))
    # failures: a reference before the first pixel, one running past the last, both after some valid data
    cases.append((64, 64, lits(10) + [("ref", 4, 11)]))
    cases.append((64, 64, lits(4000) + [("ref", 128, 7)]))
    cases.append((64, 64, lits(4090) + [("ref", 3, 1), ("ref", 2, 3), ("ref", 3, 1)]))
    cases.append((64, 64, lits(4090) + [("ref", 3, 1), ("ref", 3, 3)]))  # ends exactly with a reference
    return cases


def test_handmade_tokens_on_both_forms():
    streams, expect = [], []
    for w, h, toks in handmade_cases():
        s, e = handmade_stream(w, h, toks)
        rc0, ref, w0, h0 = O.decode(s)
        if e is not None:
            assert rc0 == 0 and (w0, h0) == (w, h) and np.array_equal(ref, e)  # the oracle reads the hand-written stream as intended
        else:
            assert rc0 != 0
        streams.append(s)
        expect.append((rc0, ref))
    for threads in (1024, 64):
        for (st, px), (rc0, ref) in zip(decode_with(threads, streams), expect):
            assert st == rc0, (threads, st, rc0)
            if rc0 == 0:
                assert np.array_equal(px, ref), threads


def test_both_forms_agree_on_encoder_streams():
    cases = [(3, 640, 360), (4, 640, 360), (5, 640, 360), (4, 4096, 3), (3, 17, 4001), (4, 1, 1), (3, 2, 2), (3, 1100, 700)]
    imgs = [O.gen_image(m, w, h, 900 + w, 200) for m, w, h in cases]
    flat = np.full(300 * 300, 0xFF336699, dtype=np.uint32)  # one colour: every stream symbol is a zero-bit code
    flat2 = flat.copy()
    flat2[::7919] ^= 0x010101
    imgs += [flat, flat2]
    cases += [(0, 300, 300), (0, 300, 300)]
    streams = []
    for img, (m, w, h) in zip(imgs, cases):
        rc, s = O.encode(img, w, h)
        assert rc == 0
        streams.append(s)
    # decoder-only features: these take the one-lane path in both forms, inside the 1024-thread kernel too
    ext = [dict(cache_bits=5), dict(meta_bits=2, meta_groups=3), dict(predictor_bits=3, use_subtract_green=1)]
    for prm in ext:
        rc, s = O.encode_ext(imgs[0], 640, 360, **prm)
        assert rc == 0
        streams.append(s)
        imgs.append(imgs[0])
    a = decode_with(1024, streams)
    b = decode_with(64, streams)
    for k, ((sa, pa), (sb, pb), img) in enumerate(zip(a, b, imgs)):
        assert sa == 0 and sb == 0, (k, sa, sb)
        assert np.array_equal(pa, img), k
        assert np.array_equal(pb, img), k


@pytest.mark.parametrize("scout_form,shift", [(0, 0), (0, 4), (1, 0)])
def test_whole_device_form(scout_form, shift):
    """Large streams cut into segments and decoded by many workgroups: same pixels and verdicts, with either form of
    the scouts, and with streams that start 4 bytes off an 8-byte boundary (the lane form reads 8-byte words)."""
    cases = [(3, 1100, 700), (4, 1100, 700), (5, 1024, 600), (4, 4096, 40), (3, 300, 300), (3, 40, 40), (4, 2000, 1100)]
    imgs = [O.gen_image(m, w, h, 700 + w, 200) for m, w, h in cases]
    flat = np.full(700 * 500, 0xFF336699, dtype=np.uint32)
    flat[::7919] ^= 0x010101
    imgs.append(flat)
    cases.append((0, 700, 500))
    streams = []
    for img, (m, w, h) in zip(imgs, cases):
        rc, s = O.encode(img, w, h)
        assert rc == 0
        streams.append(s)
    # the main image behind a predictor transform, and two kinds the segment kernels leave to one workgroup
    for prm in (dict(predictor_bits=4, use_subtract_green=1), dict(cache_bits=5), dict(meta_bits=3, meta_groups=3)):
        rc, s = O.encode_ext(imgs[0], 1100, 700, **prm)
        assert rc == 0
        streams.append(s)
        imgs.append(imgs[0])
    used = []
    res = decode_with(1024, streams, whole_device=2, used=used, scout_form=scout_form, shift=shift)
    for k, ((st, px), img) in enumerate(zip(res, imgs)):
        assert st == 0, (k, st)
        assert np.array_equal(px, img), k
    assert used[0] >= 7, used  # the large one-group streams went through the segment kernels ...
    assert used[1] >= 6, used  # ... and the scouts of the generated images were accepted (the nearly flat one repeats one token)
    # damaged streams: the verdict and the pixels of the one-workgroup form
    rng = np.random.default_rng(12)
    bad = []
    for base in (streams[0], streams[1], streams[6]):
        for it in range(12):
            b = bytearray(base)
            if it % 3 == 0:
                b = b[: int(rng.integers(len(b) // 2, len(b)))]
            else:
                pos = int(rng.integers(len(b) // 4, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
            bad.append(bytes(b))
    a = decode_with(1024, bad, whole_device=2, scout_form=scout_form, shift=shift)
    b = decode_with(1024, bad, whole_device=1)
    for k, ((sa, pa), (sb, pb)) in enumerate(zip(a, b)):
        assert sa == sb, (k, sa, sb)
        if sa == 0:
            assert np.array_equal(pa, pb), k


@pytest.mark.parametrize("scout_form", [0, 1])
def test_whole_device_handmade_tokens(scout_form):
    streams, expect = [], []
    for w, h, toks in handmade_cases():
        s, e = handmade_stream(w, h, toks)
        rc0, ref, w0, h0 = O.decode(s)
        streams.append(s)
        expect.append((rc0, ref))
    for (st, px), (rc0, ref) in zip(decode_with(1024, streams, whole_device=2, scout_form=scout_form), expect):
        assert st == rc0, (st, rc0)
        if rc0 == 0:
            assert np.array_equal(px, ref)


@pytest.mark.parametrize("scout_form", [0, 1])
def test_whole_device_when_scouts_cannot_fall_in_step(scout_form):
    """Uniform random 7-bit values give codes of one length: a decoder started at a wrong bit never meets the token chain,
    so the scouts are rejected and the entry maps decode the stream; same pixels.  (All 256 byte values would give the
    256 x 8-bit code that the reference itself cannot read back.)"""
    rng = np.random.default_rng(2)
    w, h = 768, 512
    img = rng.integers(0, 1 << 32, size=w * h, dtype=np.uint32) & 0x7F7F7F7F
    img2 = (img & 0x003F1F7F) | 0xFF000000
    streams = []
    for im in (img, img2):
        rc, s = O.encode(im, w, h)
        assert rc == 0
        streams.append(s)
    used = []
    res = decode_with(1024, streams, whole_device=2, used=used, scout_form=scout_form)
    for (st, px), im in zip(res, (img, img2)):
        assert st == 0 and np.array_equal(px, im)
    assert used[0] == 2, used  # green and alpha have codes of one length, red and blue not quite: some scouts may fall in step
    # all four codes of one length (red and blue are green plus uniform noise): tokens of 28 bits, no scout of either
    # form can fall in step
    g = rng.integers(0, 128, size=w * h, dtype=np.uint32)
    img3 = (rng.integers(0, 128, size=w * h, dtype=np.uint32) << 24) | (((g + rng.integers(0, 128, size=w * h, dtype=np.uint32)) & 255) << 16) \
        | (g << 8) | ((g + rng.integers(0, 128, size=w * h, dtype=np.uint32)) & 255)
    rc, s3 = O.encode(img3, w, h)
    assert rc == 0
    used = []
    (st, px), = decode_with(1024, [s3], whole_device=2, used=used, scout_form=scout_form)
    assert st == 0 and np.array_equal(px, img3)
    assert used == [1, 0], used


def test_lane_scouts_built_with_512_threads():
    """The lane scouts (k_par_scout_lanes) are written for any whole number of waves per workgroup; the shipped library
    uses 256 threads, libvp8l_amd_lanes512.so is the same source built with 512.  Every case of this file must give the
    same pixels, verdicts and scout counts with it (the cases run in a child process that loads that library)."""
    import os
    import subprocess
    import sys
    if os.environ.get("VP8L_AMD_LIB"):
        pytest.skip("already running on another build of the library")
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), "webp_amd", "csrc", "libvp8l_amd_lanes512.so")
    assert os.path.exists(lib), "build the test variant first (__graft_entry__.build())"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "--timeout", "120", "--timeout-method", "thread"],
                       env=dict(os.environ, VP8L_AMD_LIB=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


EXT_KINDS = [
    dict(use_subtract_green=1), dict(predictor_bits=2), dict(color_bits=3), dict(predictor_bits=3, color_bits=2, use_subtract_green=1, transform_order=1),
    dict(cache_bits=4), dict(cache_bits=11, use_subtract_green=1), dict(meta_bits=2, meta_groups=3),
    dict(meta_bits=3, meta_groups=7, cache_bits=6, predictor_bits=3, use_subtract_green=1), dict(use_palette=1, cache_bits=3),
    dict(use_palette=1, meta_bits=2, meta_groups=2),
]


def test_thousands_of_small_streams_in_one_call():
    # a call with at least 1024 small streams is decoded by a wave per stream instead of a workgroup per stream: every kind
    # of stream, and damaged ones, in one such call
    rng = np.random.default_rng(99)
    streams = []
    for prm in EXT_KINDS:
        for (w, h) in ((7, 5), (33, 18), (64, 64)):
            for mode in (3, 4, 5):
                rc, s = O.encode_ext(O.gen_image(mode, w, h, 11, 200), w, h, **prm)
                assert rc == 0
                streams.append(s)
    for (w, h) in ((1, 1), (40, 30), (255, 200)):
        for mode in (3, 4, 5):
            streams.append(O.encode(O.gen_image(mode, w, h, 3, 60), w, h)[1])
    base = list(streams)
    while len(streams) < 1100:
        b = bytearray(base[int(rng.integers(0, len(base)))])
        if len(b) > 12 and rng.random() < 0.5:
            b[int(rng.integers(9, len(b)))] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(b))
    res = decode_with(0, streams, whole_device=0)   # one call through the device-memory entry point
    for k, (s, (st, arr)) in enumerate(zip(streams, res)):
        rc0, ref, w0, h0 = O.decode(s)
        assert st == rc0, (k, st, rc0)
        if rc0 == 0:
            assert np.array_equal(arr, ref), k
