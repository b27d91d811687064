"""Randomised comparison of the oracle with the reference's own compiled components.  Runs only
where oracle/_ref/libref_parts.so exists (the build container); the same checks on a fixed set
of cases run everywhere from tests/golden/ref_parts.npz."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no reference checkout here)")


def test_lz77_random():
    R = O.ref()
    rng = np.random.default_rng(1)
    for it in range(60):
        n = int(rng.integers(1, 4000))
        alpha = int(rng.choice([1, 2, 3, 5, 17, 300, 100000]))
        data = rng.integers(0, alpha, size=n, dtype=np.uint32)
        if it % 3 == 0 and n > 600:
            s = int(rng.integers(0, n - 300)); d = int(rng.integers(1, min(1100, s + 1) + 1)) if s > 0 else 0
            if s - d >= 0:
                L = int(rng.integers(1, 300))
                data[s:s + L] = data[s - d:s - d + L][: len(data[s:s + L])]
        out = np.zeros((n, 3), dtype=np.uint32)
        k = R.ref_lz77(data.ctypes.data, n, out.ctypes.data, n)
        assert np.array_equal(O.lz77(data), out[:k]), f"iter {it} n={n} alpha={alpha}"


def test_huffman_random():
    R = O.ref()
    rng = np.random.default_rng(2)
    for it in range(300):
        n = int(rng.choice([40, 68, 256, 280]))
        kind = it % 5
        if kind == 0:
            h = rng.integers(0, 3, size=n)
        elif kind == 1:
            h = rng.integers(0, 1 << int(rng.integers(1, 24)), size=n) * (rng.random(n) < rng.random())
        elif kind == 2:
            h = rng.geometric(0.01, size=n) * (rng.random(n) < 0.7)
        elif kind == 3:
            h = np.round(1e6 * rng.dirichlet(np.full(n, 0.05)))
        else:
            h = (rng.integers(1, 5, size=n) * 100) * (rng.random(n) < 0.5)
        h = np.ascontiguousarray(h, dtype=np.uint32)
        maxlen = 32 if n == 68 else 64
        ln = np.zeros(n, dtype=np.uint8); cd = np.zeros(n, dtype=np.uint32)
        used = R.ref_huffman_build(h.ctypes.data, n, maxlen, ln.ctypes.data, cd.ctypes.data)
        u2, l2, c2 = O.huffman_build(h, maxlen)
        assert u2 == used, it
        if used < 0:
            continue
        assert np.array_equal(l2, ln) and np.array_equal(c2, cd), it
        buf = np.zeros(4096, dtype=np.uint8)
        nb = R.ref_write_code(h.ctypes.data, n, maxlen, buf.ctypes.data, buf.size)
        bits, data = O.write_code(l2)
        ref = bytearray(buf[:nb].tobytes())
        last = len(ref) - 1
        while ref[last] == 0:
            last -= 1
        nbits = last * 8 + ref[last].bit_length() - 1
        assert bits == nbits, it
        ref[nbits // 8] &= ~(1 << (nbits % 8)) & 0xFF
        assert bytes(ref[: len(data)]) == data, it


class _BW:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for i in range(n):
            self.bits.append((v >> i) & 1)

    def put_bytes_bits(self, data, nbits):
        for i in range(nbits):
            self.bits.append((data[i >> 3] >> (i & 7)) & 1)

    def tobytes(self):
        out = bytearray((len(self.bits) + 7) // 8)
        for i, b in enumerate(self.bits):
            out[i >> 3] |= b << (i & 7)
        return bytes(out)


def test_reference_reader_decodes_oracle_written_groups():
    """huffman_io.h:200-214: the reference reads 5 code descriptions written by the oracle and
    decodes symbols coded with the oracle's codes."""
    R = O.ref()
    rng = np.random.default_rng(5)
    for it in range(40):
        cache_bits = int(rng.choice([0, 0, 3, 11]))
        cache_size = (1 << cache_bits) if cache_bits else 0
        bw = _BW()
        tables = []
        for c in range(5):
            n = O.ALPHABET[c] + (cache_size if c == 0 else 0)
            nused = int(rng.choice([1, 2, 3, 10, n]))
            h = np.zeros(n, dtype=np.uint32)
            idx = rng.choice(n, size=min(nused, n), replace=False)
            h[idx] = rng.integers(1, 1000, size=idx.size)
            used, ln, cd = O.huffman_build(h)
            bits, data = O.write_code(ln)
            assert bits > 0
            bw.put_bytes_bits(data, bits)
            tables.append((np.flatnonzero(h), ln, cd, used))
        which = rng.integers(0, 5, size=200).astype(np.uint8)
        syms = np.zeros(200, dtype=np.int32)
        for i, c in enumerate(which):
            used_syms, ln, cd, used = tables[c]
            s = int(rng.choice(used_syms))
            syms[i] = s
            if used > 1:
                bw.put(int(cd[s]), int(ln[s]))
        data = bw.tobytes()
        got = np.zeros(200, dtype=np.int32)
        rc = R.ref_read_group(data, len(data), cache_size, which.ctypes.data, 200, got.ctypes.data)
        if rc == -6:
            # only the all-one-length code-length tree (reference defect: it writes 1 bit per item
            # and reads none) may be rejected; the oracle's reader must reject it too
            pos = C.c_uint64(0)
            bad = False
            for c in range(5):
                n = O.ALPHABET[c] + (cache_size if c == 0 else 0)
                ln = np.zeros(n, dtype=np.uint8); simple = C.c_int(); ns = C.c_int(); ss = (C.c_uint16 * 2)()
                if O.lib().ora_read_code(data, len(data), C.byref(pos), n, ln.ctypes.data, C.byref(simple), ss, C.byref(ns)) != 0:
                    bad = True
                    break
                if not simple.value and not np.array_equal(ln, tables[c][1]):
                    bad = True
                    break
            assert bad, it
            continue
        assert rc == 0
        assert np.array_equal(got, syms), it


def test_bit_writer_bytes():
    R = O.ref()
    rng = np.random.default_rng(9)
    vals = rng.integers(0, 1 << 32, size=500, dtype=np.uint32)
    cnts = rng.integers(0, 33, size=500, dtype=np.uint32)
    buf = np.zeros(4096, dtype=np.uint8)
    nb = R.ref_bit_writer(vals.ctypes.data, cnts.ctypes.data, 500, buf.ctypes.data, buf.size)
    bw = _BW()
    for v, n in zip(vals, cnts):
        bw.put(int(v), int(n))
    assert bw.tobytes() == buf[:nb].tobytes()
