"""PNG input/output of the command line tool (include/webp_amd/png_io.hpp) against Pillow and against PNG files
built here by hand (every colour type / bit depth / interlacing / filter the format has).  Host only.

Expected pixels follow the reference's read_png (webp.cpp:60-110): 16 -> 8 bits by dropping the low byte, grey of
less than 8 bits scaled to 8, palette and grey to RGB, tRNS to alpha; without alpha 0xFF,R,G,B; WITH alpha the
reference takes the four bytes R,G,B,A of a pixel as A,R,G,B (kept)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "tools", "png_dump")

pytestmark = pytest.mark.skipif(not os.path.exists(DUMP), reason="tools/png_dump not built")


def dump(path, tmp_path, piece=None):
    out = str(tmp_path / "out.argb")
    r = subprocess.run([DUMP, path, out] + ([str(piece)] if piece else []), capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr
    raw = open(out, "rb").read()
    w, h = struct.unpack("<II", raw[:8])
    return np.frombuffer(raw[8:], dtype=np.uint32).reshape(h, w), ""


def expected_argb(rgba, has_alpha):
    r, g, b, a = [rgba[..., k].astype(np.uint32) for k in range(4)]
    if has_alpha:
        return (r << 24) | (g << 16) | (b << 8) | a      # the reference's byte order for alpha images
    return np.uint32(0xFF000000) | (r << 16) | (g << 8) | b


# ---- a PNG writer for the tests: any colour type, depth, interlacing; filters cycled per row -----------------
def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def _pack_rows(samples, depth):           # samples: (h, w*channels) ints -> list of byte rows
    rows = []
    for row in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in row))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in row))
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in row)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def _filter(rows, bpp, start):
    out = b""
    prev = bytes(len(rows[0])) if rows else b""
    for k, cur in enumerate(rows):
        ft = (start + k) % 5
        f = bytearray()
        for i, v in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
            f.append((v - p) & 0xff)
        out += bytes([ft]) + bytes(f)
        prev = cur
    return out


def make_png(samples, ctype, depth, interlace=False, plte=None, trns=None):
    """samples: (h, w, channels) integer array at file depth."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    if interlace:
        for k, (x0, y0, dx, dy) in enumerate([(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, k)
    else:
        raw = _filter(_pack_rows(samples.reshape(h, -1), depth), bpp, 0)
    comp = zlib.compress(raw)
    idat = _chunk(b"IDAT", comp[:len(comp) // 2]) + _chunk(b"IDAT", comp[len(comp) // 2:])   # split on purpose
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    png += _chunk(b"gAMA", struct.pack(">I", 45455))      # ancillary chunks are skipped
    if plte is not None:
        png += _chunk(b"PLTE", bytes(plte))
    if trns is not None:
        png += _chunk(b"tRNS", bytes(trns))
    return png + idat + _chunk(b"IEND", b"")


def to8(v, depth):
    v = v.astype(np.uint32)
    if depth == 16:
        return v >> 8
    if depth == 8:
        return v
    return v * 255 // ((1 << depth) - 1)


CASES = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("ctype,depth", CASES)
@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("with_trns", [False, True])
def test_every_png_kind(tmp_path, ctype, depth, interlace, with_trns):
    if with_trns and ctype in (4, 6):
        pytest.skip("tRNS is not allowed with an alpha channel")
    rng = np.random.default_rng(ctype * 100 + depth + 7 * interlace)
    w, h = int(rng.integers(1, 23)), int(rng.integers(1, 19))
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    plte = trns = None
    if ctype == 3:
        ncol = 1 << depth
        pal = rng.integers(0, 256, size=(ncol, 3))
        plte = pal.reshape(-1).tolist()
        smp = rng.integers(0, ncol, size=(h, w, 1))
        rgba = np.concatenate([pal[smp[..., 0]], np.full((h, w, 1), 255)], axis=-1)
        if with_trns:
            al = rng.integers(0, 256, size=max(1, ncol // 2))
            trns = al.tolist()
            alpha = np.full(ncol, 255)
            alpha[:len(al)] = al
            rgba[..., 3] = alpha[smp[..., 0]]
    else:
        smp = rng.integers(0, 1 << depth, size=(h, w, ch))
        if ctype == 0:
            g = to8(smp[..., 0], depth)
            rgba = np.stack([g, g, g, np.full((h, w), 255)], axis=-1)
            if with_trns:
                key = int(smp[0, 0, 0])
                trns = struct.pack(">H", key)
                rgba[..., 3] = np.where(smp[..., 0] == key, 0, 255)
        elif ctype == 2:
            c8 = to8(smp, depth)
            rgba = np.concatenate([c8, np.full((h, w, 1), 255)], axis=-1)
            if with_trns:
                key = smp[h // 2, w // 2]
                trns = struct.pack(">HHH", *[int(v) for v in key])
                rgba[..., 3] = np.where((smp == key).all(axis=-1), 0, 255)
        elif ctype == 4:
            g, a = to8(smp[..., 0], depth), to8(smp[..., 1], depth)
            rgba = np.stack([g, g, g, a], axis=-1)
        else:
            rgba = to8(smp, depth)
    path = str(tmp_path / "in.png")
    open(path, "wb").write(make_png(smp, ctype, depth, interlace, plte, trns))
    got, err = dump(path, tmp_path)
    assert got is not None, err
    has_alpha = with_trns or ctype in (4, 6)
    assert np.array_equal(got, expected_argb(rgba, has_alpha))


REF_CLI = os.path.join(ROOT, "oracle", "_ref", "webp_ref_cli")


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref/webp_ref_cli not built (no reference checkout here)")
def test_reference_tool_reads_the_same_pixels(tmp_path):
    """The reference's own command line tool (webp.cpp with libpng, compiled in place by oracle/Makefile) encodes every
    PNG kind; the pixels inside its .webp (read back with the oracle's decoder) must be the pixels png_io gives, and its
    -d must write the RGB that png_io's writer writes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from PIL import Image
    rng = np.random.default_rng(77)
    n = 0
    for ctype, depth in CASES:
        for interlace in (False, True):
            for with_trns in (False, True):
                if with_trns and ctype in (4, 6):
                    continue
                # widths that fill whole groups of bundled palette indices: the reference's own decoder overruns its
                # rows otherwise (transform.h:70-86) and its -d then crashes on few-colour pictures
                w, h = 8 * int(rng.integers(1, 4)), int(rng.integers(2, 19))
                ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
                plte = trns = None
                if ctype == 3:
                    ncol = 1 << depth
                    plte = rng.integers(0, 256, size=ncol * 3).tolist()
                    smp = rng.integers(0, ncol, size=(h, w, 1))
                    if with_trns:
                        trns = rng.integers(0, 256, size=max(1, ncol // 2)).tolist()
                else:
                    # few levels, so that no channel uses all 256 values evenly (the stream the reference cannot read back)
                    smp = rng.integers(0, min(1 << depth, 23), size=(h, w, ch)) * (1 if depth < 16 else 2311)
                    if with_trns:
                        v = smp[0, 0]
                        trns = b"".join(struct.pack(">H", int(x)) for x in v)
                path = str(tmp_path / "k.png")
                open(path, "wb").write(make_png(smp, ctype, depth, interlace, plte, trns))
                mine, err = dump(path, tmp_path)
                assert mine is not None, err
                out = str(tmp_path / "k.webp")
                if os.path.exists(out):
                    os.remove(out)
                r = subprocess.run([REF_CLI, "-e", path, out], capture_output=True, text=True)
                assert r.returncode == 0
                blob = open(out, "rb").read()
                assert blob[:4] == b"RIFF" and blob[8:16] == b"WEBPVP8L"
                rc, dec, dw, dh = O.decode(blob[16:])
                assert rc == 0 and (dw, dh) == (w, h), (ctype, depth, interlace, with_trns)
                assert np.array_equal(dec.reshape(h, w), mine), (ctype, depth, interlace, with_trns)
                # and back: the reference's save2png against this repository's writer
                ref_png, my_png, raw = str(tmp_path / "r.png"), str(tmp_path / "m.png"), str(tmp_path / "m.argb")
                assert subprocess.run([REF_CLI, "-d", out, ref_png], capture_output=True).returncode == 0
                open(raw, "wb").write(struct.pack("<II", w, h) + mine.astype(np.uint32).tobytes())
                assert subprocess.run([DUMP, "-w", raw, my_png]).returncode == 0
                a, b = Image.open(ref_png), Image.open(my_png)
                assert a.mode == b.mode == "RGB" and np.array_equal(np.asarray(a), np.asarray(b))
                n += 1
    assert n > 40


def test_against_pillow(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    for mode in ("RGB", "RGBA", "L", "LA", "P", "1"):
        w, h = 37, 21
        if mode == "P":
            im = Image.fromarray(rng.integers(0, 256, size=(h, w), dtype=np.uint8), "P")
            im.putpalette(rng.integers(0, 256, size=768, dtype=np.uint8).tobytes())
        elif mode == "1":
            im = Image.fromarray((rng.integers(0, 2, size=(h, w), dtype=np.uint8) * 255), "L").convert("1")
        else:
            ch = len(mode)
            arr = rng.integers(0, 256, size=(h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
            im = Image.fromarray(arr, mode)
        path = str(tmp_path / ("pil_%s.png" % mode))
        im.save(path)
        got, err = dump(path, tmp_path)
        assert got is not None, err
        rgba = np.asarray(Image.open(path).convert("RGBA")).astype(np.uint32)
        assert np.array_equal(got, expected_argb(rgba, mode in ("RGBA", "LA"))), mode


def test_writer_read_back_by_pillow(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(6)
    w, h = 45, 33
    argb = rng.integers(0, 1 << 32, size=(h, w), dtype=np.uint64).astype(np.uint32)
    src = str(tmp_path / "a.argb")
    open(src, "wb").write(struct.pack("<II", w, h) + argb.tobytes())
    out = str(tmp_path / "a.png")
    assert subprocess.run([DUMP, "-w", src, out]).returncode == 0
    im = Image.open(out)
    assert im.mode == "RGB" and im.size == (w, h)       # save2png drops alpha (webp.h:90-95,122-125)
    rgb = np.asarray(im).astype(np.uint32)
    assert np.array_equal((rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2], argb & 0xFFFFFF)


def test_image_data_inflated_in_many_pieces(tmp_path):
    """zlib counts in 32 bits, so the reader offers input and output room in pieces (1 GiB); a piece of a few bytes
    forces the many-pieces path on a small file: same pixels, and damaged or short data is still refused."""
    rng = np.random.default_rng(8)
    w, h = 61, 47
    smp = rng.integers(0, 256, size=(h, w, 4))
    for interlace in (False, True):
        path = str(tmp_path / ("pieces%d.png" % interlace))
        png = make_png(smp, 6, 8, interlace)
        open(path, "wb").write(png)
        whole, err = dump(path, tmp_path)
        assert whole is not None, err
        for piece in (1, 7, 100, 4096):
            got, err = dump(path, tmp_path, piece)
            assert got is not None, (piece, err)
            assert np.array_equal(got, whole), piece
    # incompressible data: output room runs out many times with input still pending
    path = str(tmp_path / "stored.png")
    raw = b"".join(b"\0" + rng.integers(0, 256, size=w * 4, dtype=np.uint8).tobytes() for _ in range(h))
    png = (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
           + _chunk(b"IDAT", zlib.compress(raw, 0)) + _chunk(b"IEND", b""))
    open(path, "wb").write(png)
    whole, err = dump(path, tmp_path)
    assert whole is not None, err
    for piece in (3, 1000):
        got, err = dump(path, tmp_path, piece)
        assert got is not None and np.array_equal(got, whole), (piece, err)
    short = (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h + 1, 8, 6, 0, 0, 0))
             + _chunk(b"IDAT", zlib.compress(raw, 0)) + _chunk(b"IEND", b""))
    open(path, "wb").write(short)
    for piece in (None, 5):
        got, err = dump(path, tmp_path, piece)
        assert got is None


def test_broken_files_are_refused(tmp_path):
    good = make_png(np.zeros((4, 4, 3), dtype=np.int64), 2, 8)
    for bad in (b"", b"notapng", good[:40], good[:-30], good.replace(b"IHDR", b"IHDx")):
        path = str(tmp_path / "bad.png")
        open(path, "wb").write(bad)
        got, err = dump(path, tmp_path)
        assert got is None


def test_header_sizes_beyond_the_codec_limit_allocate_nothing(tmp_path):
    """A 60-byte file whose header claims 2^31 x 2^31 pixels is refused right after IHDR (the encoder's limit is 16384 a
    side), quickly and with an error exit rather than an out-of-memory kill."""
    import time
    good = make_png(np.zeros((4, 4, 3), dtype=np.int64), 2, 8)
    i = good.index(b"IHDR")
    for w, h in ((1 << 31, 1 << 31), (16385, 4), (4, 16385)):
        hdr = struct.pack(">II", w, h) + good[i + 12:i + 17]
        bad = good[:i - 4] + _chunk(b"IHDR", hdr) + good[i + 21:]
        path = str(tmp_path / "huge.png")
        open(path, "wb").write(bad)
        t0 = time.time()
        got, err = dump(path, tmp_path)
        assert got is None and time.time() - t0 < 5
