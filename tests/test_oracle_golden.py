"""The CPU oracle against the golden vectors recorded from the compiled reference components
(tests/golden/ref_parts.npz, made by tests/golden/make_golden.py) and against the whole-stream
known answers of the compiled reference encoder (tests/golden/kat_streams.json)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_parts.npz"), allow_pickle=False)


def test_lz77_token_streams(G):
    i = 0
    while f"lz_in_{i}" in G:
        toks = O.lz77(G[f"lz_in_{i}"])
        assert np.array_equal(toks, G[f"lz_tok_{i}"]), f"case {i}"
        i += 1
    assert i >= 10


def test_prefix_encode(G):
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    for v, exp in zip(G["prefix_v"], G["prefix_enc"]):
        O.lib().ora_prefix_encode(int(v), C.byref(a), C.byref(b), C.byref(c))
        assert (a.value, b.value, c.value) == tuple(int(t) for t in exp), int(v)


def test_plane_codes(G):
    L = O.lib()
    for i, xs in enumerate(G["plane_xsize"]):
        got = np.array([L.ora_dist_to_code(int(xs), d) for d in range(1, 1025)], dtype=np.uint32)
        assert np.array_equal(got, G["plane_d2c"][i]), f"d2c xsize={xs}"
        got = np.array([L.ora_code_to_dist(int(xs), c) for c in range(1, 1145)], dtype=np.uint32)
        assert np.array_equal(got, G["plane_c2d"][i]), f"c2d xsize={xs}"


def _bits_of_desc(desc):
    """The reference bytes carry a terminating 1 bit after the description."""
    last = len(desc) - 1
    while desc[last] == 0:
        last -= 1
    top = int(desc[last]).bit_length() - 1
    return last * 8 + top


def test_huffman_lengths_codes_and_description(G):
    i = 0
    while f"huf_histo_{i}" in G:
        h = G[f"huf_histo_{i}"]
        maxlen = 32 if h.size == 68 else 64
        used, ln, cd = O.huffman_build(h, maxlen)
        assert used == int(G[f"huf_used_{i}"][0]), f"case {i}"
        if used >= 0:
            assert np.array_equal(ln, G[f"huf_len_{i}"]), f"lengths case {i}"
            assert np.array_equal(cd, G[f"huf_code_{i}"]), f"codes case {i}"
            desc = G[f"huf_desc_{i}"]
            nbits = _bits_of_desc(desc)
            bits, data = O.write_code(ln)
            assert bits == nbits, f"desc bits case {i}"
            ref = bytearray(desc.tobytes())
            ref[nbits // 8] &= ~(1 << (nbits % 8)) & 0xFF
            assert bytes(ref[: len(data)]) == data, f"desc bytes case {i}"
            assert not any(ref[len(data):])
        i += 1
    assert i >= 40


def test_huffman_kat():
    # counts {1,1,1,1,2,2,4,0} -> lengths {4,4,4,4,3,3,1,0}, written codes {3,11,7,15,1,5,0,-}
    used, ln, cd = O.huffman_build(np.array([1, 1, 1, 1, 2, 2, 4, 0], dtype=np.uint32))
    assert used == 7
    assert ln.tolist() == [4, 4, 4, 4, 3, 3, 1, 0]
    assert cd.tolist()[:7] == [3, 11, 7, 15, 1, 5, 0]


def test_inverse_transforms(G):
    n = int(G["tr_count"][0])
    assert n > 50
    for t in range(n):
        typ, w, h, tx, ty, bits, rc = (int(v) for v in G[f"tr_{t}_meta"])
        td = G[f"tr_{t}_td"]
        got_rc, out = O.inverse_transform(typ, G[f"tr_{t}_in"], w, h, td if td.size else None, tx, bits)
        assert got_rc == rc, f"case {t}"
        if rc == 0:
            assert np.array_equal(out, G[f"tr_{t}_out"]), f"case {t} type {typ} {w}x{h}"
        if f"tr_{t}_idx" in G:  # bundling
            idx, packed = G[f"tr_{t}_idx"], G[f"tr_{t}_packed"]
            for y in range(h):
                dst = np.zeros(packed.shape[1], dtype=np.uint32)
                O.lib().ora_bundle(np.ascontiguousarray(idx[y]).ctypes.data, w, bits, dst.ctypes.data)
                assert np.array_equal(dst, packed[y])


def test_colour_cache_hash(G):
    cols = G["cc_colors"]
    for bits in (1, 4, 11):
        tab = np.zeros(1 << bits, dtype=np.uint32)
        for c in cols:
            tab[((0x1E35A7BD * int(c)) & 0xFFFFFFFF) >> (32 - bits)] = c
        assert np.array_equal(tab, G[f"cc_table_{bits}"])


def test_whole_stream_known_answers(golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "kat_streams.json")))
    assert len(kats) == 7
    for k in kats:
        img = O.gen_image(k["mode"], k["w"], k["h"], k["seed"], 200)
        rc, data = O.encode(img, k["w"], k["h"])
        assert rc == 0
        assert len(data) == k["bytes"], k
        assert hashlib.sha256(data).hexdigest().startswith(k["sha256_16"]), k
        if "stream_hex" in k:
            assert data.hex() == k["stream_hex"]
        rc, dec, w, h = O.decode(data)
        assert rc == 0 and (w, h) == (k["w"], k["h"])
        assert np.array_equal(dec, img)
