#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the COMPILED REFERENCE COMPONENTS
(oracle/_ref/libref_parts.so, built by `make -C oracle ref` from /root/reference in place).

Run in the build container only (the reference checkout does not exist on the GPU box):
    python tests/golden/make_golden.py

Outputs (data only: inputs and the reference's outputs for them):
  ref_parts.npz      component vectors: LZ77 token streams, prefix/distance codes, Huffman
                     lengths+codes, serialised code descriptions, inverse transforms, bundling,
                     colour-cache tables, bit-writer bytes.
  kat_streams.json   whole-stream known answers (size + sha256) recorded from the compiled
                     reference encoder for seven synthetic images, plus the oracle's streams for
                     the small ones (hex) once the oracle reproduces those hashes.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

import ctypes as C  # noqa: E402

# size/sha256-prefix pairs measured on the compiled reference (VP8_LOSSLESS_ENCODER stream bytes,
# generators mode 3/4/5, seed 12345): (w, h, mode, bytes, sha256[:16])
REFERENCE_KATS = [
    (64, 64, 3, 8502, "fbb85bbe23ed39c2"),
    (64, 64, 4, 641, "f3231143c078aaf9"),
    (64, 64, 5, 1716, "5e623affc190cb34"),
    (61, 19, 5, 617, "e18482afedfc500c"),
    (512, 512, 3, 88515, "883c59786b1dd8be"),
    (512, 512, 4, 35614, "e5f537ed74fb94cc"),
    (512, 512, 5, 59802, "25cb6e63c6372294"),
]


def lz_inputs(rng):
    cases = []
    for n, alpha in [(1, 2), (2, 1), (3, 2), (17, 2), (130, 1), (300, 3), (1500, 4), (2600, 50), (3000, 100000)]:
        cases.append(rng.integers(0, alpha, size=n, dtype=np.uint32))
    # injected repeats (long matches, window edge, >128 runs)
    a = rng.integers(0, 1 << 24, size=2500, dtype=np.uint32)
    a[700:1000] = a[100:400]
    a[1500:1700] = a[1499]
    a[2000:2300] = a[976:1276]  # distance exactly 1024
    cases.append(a)
    b = np.tile(rng.integers(0, 7, size=37, dtype=np.uint32), 60)
    cases.append(b)
    return cases


def histo_inputs(rng):
    out = []
    for n in (40, 68, 256, 280):
        out.append(np.zeros(n, dtype=np.uint32))
        h = np.zeros(n, dtype=np.uint32); h[n // 3] = 5; out.append(h)
        h = np.zeros(n, dtype=np.uint32); h[0] = 1; h[n - 1] = 1; out.append(h)
        h = np.zeros(n, dtype=np.uint32); h[3] = 9; h[n - 2] = 2; out.append(h)
        out.append(np.ones(n, dtype=np.uint32))
        out.append(rng.integers(0, 4, size=n, dtype=np.uint32))          # tie heavy
        out.append(rng.integers(0, 3, size=n, dtype=np.uint32) * 7)
        out.append((rng.integers(0, 1 << 20, size=n) * (rng.random(n) < 0.4)).astype(np.uint32))
        out.append((2 ** rng.integers(0, 18, size=n)).astype(np.uint32))
        g = np.zeros(n, dtype=np.uint32); k = min(n, 30); a, b = 1, 1
        for i in range(k):
            g[i * (n // k)] = a; a, b = b, a + b
        out.append(g)                                                    # Fibonacci: deep tree
    out.append(np.array([1, 1, 1, 1, 2, 2, 4, 0], dtype=np.uint32))
    return out


def main():
    R = O.ref()
    if R is None:
        sys.exit("oracle/_ref/libref_parts.so missing: run `make -C oracle ref` in the build container")
    rng = np.random.default_rng(20251111)
    z = {}

    # --- LZ77 token streams
    for i, data in enumerate(lz_inputs(rng)):
        out = np.zeros((data.size, 3), dtype=np.uint32)
        k = R.ref_lz77(data.ctypes.data, data.size, out.ctypes.data, data.size)
        z[f"lz_in_{i}"] = data
        z[f"lz_tok_{i}"] = out[:k].copy()

    # --- prefix codes for v = 1..1300 and a few large
    vs = np.concatenate([np.arange(1, 1301), [4096, 65535, 65536, 1 << 20]]).astype(np.uint32)
    pe = np.zeros((vs.size, 3), dtype=np.uint32)
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    for i, v in enumerate(vs):
        R.ref_prefix_encode(int(v), C.byref(a), C.byref(b), C.byref(c))
        pe[i] = (a.value, b.value, c.value)
    z["prefix_v"] = vs
    z["prefix_enc"] = pe

    # --- distance <-> plane code
    xs = np.array([1, 2, 3, 5, 7, 8, 9, 10, 15, 16, 17, 61, 100, 200, 256, 512, 1920, 3840, 16383], dtype=np.uint32)
    d2c = np.zeros((xs.size, 1024), dtype=np.uint32)
    c2d = np.zeros((xs.size, 1144), dtype=np.uint32)
    for i, x in enumerate(xs):
        for d in range(1, 1025):
            d2c[i, d - 1] = R.ref_dist_to_code(int(x), d)
        for code in range(1, 1145):
            c2d[i, code - 1] = R.ref_code_to_dist(int(x), code)
    z["plane_xsize"] = xs
    z["plane_d2c"] = d2c
    z["plane_c2d"] = c2d

    # --- Huffman lengths/codes + serialised descriptions
    for i, h in enumerate(histo_inputs(rng)):
        n = h.size
        ln = np.zeros(n, dtype=np.uint8)
        cd = np.zeros(n, dtype=np.uint32)
        maxlen = 32 if n == 68 else 64
        used = R.ref_huffman_build(h.ctypes.data, n, maxlen, ln.ctypes.data, cd.ctypes.data)
        buf = np.zeros(4096, dtype=np.uint8)
        nb = R.ref_write_code(h.ctypes.data, n, maxlen, buf.ctypes.data, buf.size)
        z[f"huf_histo_{i}"] = h
        z[f"huf_used_{i}"] = np.array([used], dtype=np.int64)
        z[f"huf_len_{i}"] = ln
        z[f"huf_code_{i}"] = cd
        z[f"huf_desc_{i}"] = buf[: max(nb, 0)].copy()  # includes the terminating 1 bit

    # --- inverse transforms
    t = 0
    for (w, h, bits) in [(1, 1, 2), (5, 1, 2), (1, 6, 2), (13, 9, 2), (37, 21, 3), (64, 33, 4)]:
        img = rng.integers(0, 1 << 32, size=w * h, dtype=np.uint32)
        tx, ty = (w + (1 << bits) - 1) >> bits, (h + (1 << bits) - 1) >> bits
        modes = (rng.integers(0, 14, size=tx * ty, dtype=np.uint32) << 8) | 0xFF000000 | rng.integers(0, 255, size=tx * ty, dtype=np.uint32)
        for typ, td in ((0, modes), (1, rng.integers(0, 1 << 32, size=tx * ty, dtype=np.uint32))):
            out = img.copy()
            rc = R.ref_inverse_transform(typ, out.ctypes.data, w, h, td.ctypes.data, tx, ty, bits)
            z[f"tr_{t}_meta"] = np.array([typ, w, h, tx, ty, bits, rc], dtype=np.int64)
            z[f"tr_{t}_in"] = img; z[f"tr_{t}_td"] = td; z[f"tr_{t}_out"] = out; t += 1
        out = img.copy()
        rc = R.ref_inverse_transform(2, out.ctypes.data, w, h, None, 0, 0, 0)
        z[f"tr_{t}_meta"] = np.array([2, w, h, 0, 0, 0, rc], dtype=np.int64)
        z[f"tr_{t}_in"] = img; z[f"tr_{t}_td"] = np.zeros(0, dtype=np.uint32); z[f"tr_{t}_out"] = out; t += 1
    # one predictor case per mode over a whole image (tile covers everything), plus an invalid mode
    for mode in list(range(14)) + [14]:
        w, h, bits = 19, 11, 5
        img = rng.integers(0, 1 << 32, size=w * h, dtype=np.uint32)
        td = np.array([0xFF000000 | (mode << 8)], dtype=np.uint32)
        out = img.copy()
        rc = R.ref_inverse_transform(0, out.ctypes.data, w, h, td.ctypes.data, 1, 1, bits)
        z[f"tr_{t}_meta"] = np.array([0, w, h, 1, 1, bits, rc], dtype=np.int64)
        z[f"tr_{t}_in"] = img; z[f"tr_{t}_td"] = td; z[f"tr_{t}_out"] = out; t += 1
    # colour indexing, bits 0..3, widths that do and do not fill the last group
    for ncol in (200, 16, 5, 4, 3, 2, 1):
        bits = 0 if ncol > 16 else 1 if ncol > 4 else 2 if ncol > 2 else 3
        for w, h in ((16, 3), (13, 4), (1, 2)):
            cix = (w + (1 << bits) - 1) >> bits
            pal = rng.integers(0, 1 << 32, size=ncol, dtype=np.uint32)
            idx = rng.integers(0, ncol, size=(h, w), dtype=np.uint8)
            packed = np.zeros((h, cix), dtype=np.uint32)
            for y in range(h):
                R.ref_bundle(np.ascontiguousarray(idx[y]).ctypes.data, w, bits, packed[y].ctypes.data)
            img = np.zeros(w * h, dtype=np.uint32)
            img[: cix * h] = packed.reshape(-1)
            out = img.copy()
            rc = R.ref_inverse_transform(3, out.ctypes.data, w, h, pal.ctypes.data, ncol, 1, bits)
            z[f"tr_{t}_meta"] = np.array([3, w, h, ncol, 1, bits, rc], dtype=np.int64)
            z[f"tr_{t}_in"] = img; z[f"tr_{t}_td"] = pal; z[f"tr_{t}_out"] = out
            z[f"tr_{t}_idx"] = idx; z[f"tr_{t}_packed"] = packed; t += 1
    z["tr_count"] = np.array([t], dtype=np.int64)

    # --- pixel add/sub
    pa = rng.integers(0, 1 << 32, size=64, dtype=np.uint32)
    pb = rng.integers(0, 1 << 32, size=64, dtype=np.uint32)
    z["px_a"] = pa; z["px_b"] = pb
    z["px_sub"] = np.array([R.ref_pixels_sub(int(x), int(y)) for x, y in zip(pa, pb)], dtype=np.uint32)
    z["px_sum"] = np.array([R.ref_pixels_sum(int(x), int(y)) for x, y in zip(pa, pb)], dtype=np.uint32)

    # --- colour cache tables
    cols = rng.integers(0, 1 << 32, size=300, dtype=np.uint32)
    z["cc_colors"] = cols
    for bits in (1, 4, 11):
        tab = np.zeros(1 << bits, dtype=np.uint32)
        R.ref_color_cache(bits, cols.ctypes.data, cols.size, tab.ctypes.data)
        z[f"cc_table_{bits}"] = tab

    # --- bit writer
    vals = rng.integers(0, 1 << 32, size=200, dtype=np.uint32)
    cnts = rng.integers(0, 33, size=200, dtype=np.uint32)
    buf = np.zeros(2048, dtype=np.uint8)
    nb = R.ref_bit_writer(vals.ctypes.data, cnts.ctypes.data, vals.size, buf.ctypes.data, buf.size)
    z["bw_vals"] = vals; z["bw_cnts"] = cnts; z["bw_bytes"] = buf[:nb].copy()

    np.savez_compressed(os.path.join(HERE, "ref_parts.npz"), **z)

    # --- whole-stream known answers
    kats = []
    for w, h, mode, size, sha in REFERENCE_KATS:
        img = O.gen_image(mode, w, h, 12345, 200)
        rc, data = O.encode(img, w, h)
        ok = rc == 0 and len(data) == size and hashlib.sha256(data).hexdigest().startswith(sha)
        ent = dict(w=w, h=h, mode=mode, seed=12345, bytes=size, sha256_16=sha, source="compiled reference encoder")
        if ok and size < 10000:
            ent["stream_hex"] = data.hex()  # oracle bytes, identical to the reference's by the hash above
        if not ok:
            print("WARNING: oracle does not reproduce", ent)
        kats.append(ent)
    with open(os.path.join(HERE, "kat_streams.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote ref_parts.npz (%d arrays) and kat_streams.json" % len(z))


if __name__ == "__main__":
    main()
