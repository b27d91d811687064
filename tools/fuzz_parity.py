"""Randomised parity sweep (GPU box): many small images of many kinds through vp8l_encode / vp8l_encode_batch / vp8l_decode, each
stream compared with the oracle's byte for byte (default options) or decoded by the oracle back to the input (optional
coders), and decoded on the GPU back to the input.  usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import _oracle as O
import _webpstd as S
import webp_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
webp_amd.init()


def image(kind, w, h):
    n = w * h
    A = np.uint32(0xFF000000)
    if kind == 0:   # generators of the benchmark
        return O.gen_image(int(rng.integers(3, 6)), w, h, int(rng.integers(1, 1 << 20)), int(rng.choice([2, 3, 4, 5, 16, 17, 200, 256])))
    if kind == 1:   # noise over an alphabet of some size
        alpha = int(rng.choice([1, 2, 3, 4, 7, 16, 40, 256, 300, 5000, 1 << 24]))
        return (rng.integers(0, alpha, size=n).astype(np.uint32) * np.uint32(0x00010307) & np.uint32(0xFFFFFF)) | A
    if kind == 2:   # runs
        k = max(1, n // int(rng.integers(1, 60)))
        vals = rng.integers(0, int(rng.choice([2, 5, 300, 1 << 24])), size=k).astype(np.uint32) * np.uint32(0x00030201) & np.uint32(0xFFFFFF)
        return np.resize(np.repeat(vals | A, rng.integers(1, 150, size=k)), n).astype(np.uint32)
    if kind == 3:   # periodic with defects
        per = int(rng.choice([1, 2, 3, 5, 64, 127, 128, 129, 1000, 1024, 1025, max(1, w), max(1, w) + 1]))
        base = rng.integers(0, 1 << 24, size=per).astype(np.uint32) | A
        a = np.resize(base, n).astype(np.uint32)
        for _ in range(int(rng.integers(0, 6))):
            a[int(rng.integers(0, n))] ^= np.uint32(0x00010000)
        return a
    if kind == 4:   # one frequent value with specks (what predictor residuals look like), varying alpha
        a = np.full(n, 0xFF000000 | int(rng.integers(0, 1 << 24)), dtype=np.uint32)
        m = rng.random(n) < rng.choice([0.001, 0.02, 0.2])
        a[m] = rng.integers(0, 1 << 32, size=int(m.sum()), dtype=np.uint64).astype(np.uint32)
        return a
    # smooth gradient with noise, non-opaque alpha now and then
    y, x = np.divmod(np.arange(n, dtype=np.int64), w)
    nz = rng.integers(0, int(rng.choice([1, 2, 8])), size=n)
    g = (x * 3 + y * 2 + nz) & 255
    r = (x + nz) & 255
    b = (y + 2 * nz) & 255
    al = np.where(rng.random(n) < 0.01, rng.integers(0, 256, size=n), 255) if rng.random() < 0.3 else np.full(n, 255)
    return ((al.astype(np.uint32) << 24) | (r.astype(np.uint32) << 16) | (g.astype(np.uint32) << 8) | b.astype(np.uint32)).astype(np.uint32)


t_end = time.time() + budget
cases = fails = undecodable = 0
while time.time() < t_end:
    w = int(rng.choice([1, 2, 7, 8, 9, 64, 130, 256, 300, 512, 700, 1024, 2049]))
    h = int(rng.integers(1, max(2, min(200, 200000 // w))))
    img = image(int(rng.integers(0, 6)), w, h)
    opts = {}
    if rng.random() < 0.4:
        opts = dict(predictor_bits=int(rng.choice([0, 2, 4, 5])), cross_color_bits=int(rng.choice([0, 0, 3, 5])),
                    cache_bits=int(rng.choice([0, 0, 1, 6, 11])), meta_bits=int(rng.choice([0, 0, 2, 5])), standard=bool(rng.random() < 0.3))
    cases += 1
    try:
        s = webp_amd.encode(img, w, h, **opts)
        plain = not any(opts.values())
        if plain:
            rc, ref = O.encode(img, w, h)
            assert rc == 0 and s == ref, "stream differs from the oracle's"
        if opts.get("standard"):   # the standard dialect's checker is libwebp (the oracle restates the reference's dialect only)
            if S.pillow() is not None:
                back, ow, oh = S.libwebp_decode(S.riff(s))
                assert (ow, oh) == (w, h) and np.array_equal(back, img), "libwebp does not decode the stream to the input"
            px, dw, dh = webp_amd.decode(s)
            assert dw == w and dh == h and np.array_equal(np.asarray(px, dtype=np.uint32).ravel(), img), "GPU decode differs"
            continue
        rc, back, ow, oh = O.decode(s)
        if rc == -6:
            # The reference writes streams it cannot read: a code whose lengths are all 8 is described with one bit per item that
            # no reader takes (huffman_io.h:363 against huffman_coding.h:207-214).  Reproduced byte for byte; the GPU decoder must
            # refuse it like the oracle does.
            undecodable += 1
            try:
                webp_amd.decode(s)
                raise AssertionError("the GPU decoded a stream the oracle calls InvalidHuffman")
            except webp_amd.Vp8lError as e:
                assert e.status == -6, "verdicts differ: oracle -6, GPU %d" % e.status
        else:
            assert rc == 0 and ow == w and oh == h and np.array_equal(back, img), "the oracle does not decode the stream to the input (rc %d)" % rc
            px, dw, dh = webp_amd.decode(s)
            assert dw == w and dh == h and np.array_equal(np.asarray(px, dtype=np.uint32).ravel(), img), "GPU decode differs"
    except Exception as exc:  # keep going: report every failing case
        fails += 1
        print("FAIL", w, h, opts, repr(exc)[:200], flush=True)
    if cases % 50 == 0:
        print(cases, "cases,", fails, "failures,", undecodable, "streams that no reader of the dialect takes (both decoders agree)", flush=True)
# a batch through the batch entry points, mixed sizes
imgs = []
for _ in range(40):
    w = int(rng.choice([1, 9, 130, 512, 1024])); h = int(rng.integers(1, 120))
    imgs.append((image(int(rng.integers(0, 6)), w, h), w, h))
streams = webp_amd.encode_batch(imgs)
for (im, w, h), (st, s) in zip(imgs, streams):
    rc, ref = O.encode(im, w, h)
    if not (rc == 0 and st == 0 and s == ref):
        fails += 1
        print("FAIL batch", w, h, flush=True)
outs = webp_amd.decode_batch([s for _, s in streams])
for (im, w, h), o, (_, s) in zip(imgs, outs, streams):
    if o[0] == -6 and O.decode(s)[0] == -6:
        continue  # (see above)
    if not (o[0] == 0 and np.array_equal(o[1], im)):
        fails += 1
        print("FAIL batch decode", w, h, flush=True)
print("done:", cases, "cases + a batch of 40,", fails, "failures")
sys.exit(1 if fails else 0)
