"""The oracle against the reference's own compiled encoder and decoder DRIVERS
(webp/vp8l/vp8l.h:585-615 and :330-351, oracle/_ref/libref_driver.so).  Runs only where that
library was built (the build container: it needs the reference checkout and a libpng header);
the recorded form of the same checks, tests/golden/ref_driver.json, runs everywhere
(test_oracle_golden_driver.py)."""
import numpy as np
import pytest

import _oracle as O
from test_oracle_ext import CASES as EXT_CASES

pytestmark = pytest.mark.skipif(not O.ref_driver_available(),
                                reason="oracle/_ref/libref_driver.so not built (no reference checkout here)")


def _palette_image(rng, ncol, w, h):
    pal = rng.integers(0, 1 << 32, size=ncol, dtype=np.uint32)
    return pal[rng.integers(0, ncol, size=w * h)]


def test_whole_streams_equal_the_reference_encoder():
    """vp8l.h:585-612: generators, noise, palettes of every bundling width, edge shapes."""
    rng = np.random.default_rng(11)
    cases = []
    for mode in (3, 4, 5):
        for (w, h) in ((1, 1), (2, 1), (1, 2), (7, 5), (8, 8), (61, 19), (130, 3), (3, 130), (100, 64)):
            cases.append((O.gen_image(mode, w, h, 12345 + w, 200), w, h))
    for ncol in (1, 2, 3, 4, 5, 16, 17, 200, 256, 257):
        for (w, h) in ((1, 1), (9, 4), (16, 16), (63, 17), (130, 5)):
            cases.append((_palette_image(rng, ncol, w, h), w, h))
    for (w, h) in ((1, 1), (5, 3), (40, 40)):
        cases.append((rng.integers(0, 1 << 32, size=w * h, dtype=np.uint32), w, h))
    # few values above 256 colours: long matches at every distance
    cases.append((rng.integers(0, 3, size=90 * 70, dtype=np.uint32) * 0x010203 + (np.arange(6300) % 300).astype(np.uint32) * 0x01000000, 90, 70))
    assert len(cases) > 80
    for i, (img, w, h) in enumerate(cases):
        rc_ref, ref = O.ref_encode(img, w, h)
        rc, data = O.encode(img, w, h)
        assert rc == rc_ref, f"case {i} {w}x{h}"
        if rc == 0:
            assert data == ref, f"case {i} {w}x{h}"


def test_encoder_argument_errors():
    # vp8l.h:587-590
    img = np.zeros(4, dtype=np.uint32)
    assert O.ref_encode(img, 0, 4)[0] == O.encode(img, 0, 4)[0] == -1
    assert O.ref_encode(img, 4, 0)[0] == O.encode(img, 4, 0)[0] == -1
    assert O.ref_encode(img, 16385, 1)[0] == O.encode(img, 16385, 1)[0] == -2
    assert O.ref_encode(img, 1, 16385)[0] == O.encode(img, 1, 16385)[0] == -2


def _defect2(prm, w):
    """transform.h:70-86 writes whole pixel groups: a bundled palette row that the width does
    not fill overruns the image (the reference's own heap damage, not a property of the stream)."""
    return bool(prm.get("use_palette")) and w % 8 != 0


@pytest.mark.parametrize("prm", EXT_CASES)
def test_reference_decoder_reads_the_test_vector_encoder(prm):
    """vp8l.h:101-154,180-328: transform list, sub-images, colour cache, meta codes — none of which
    the reference encoder can write — decoded by the reference's own decoder."""
    for mode in (3, 4, 5, "rand"):
        for (w, h) in ((1, 1), (7, 5), (33, 18), (64, 64), (40, 24)):
            if _defect2(prm, w):
                continue
            img = (np.random.default_rng(7).integers(0, 1 << 32, size=w * h, dtype=np.uint32) if mode == "rand"
                   else O.gen_image(mode, w, h, 7, 200))
            if prm.get("use_palette") and np.unique(img).size > 256:
                continue
            rc, data = O.encode_ext(img, w, h, **prm)
            assert rc == 0
            rc, dec, dw, dh = O.ref_decode(data, 1 << 16)
            assert rc == 0, (prm, mode, w, h)
            assert (dw, dh) == (w, h) and np.array_equal(dec, img), (prm, mode, w, h)


def _verdicts_agree(rc_ref, dec_ref, rc, dec):
    """The oracle checks what the reference leaves unchecked (back references before the image,
    pixels beyond it: vp8l.h:298-300): there it says InvalidVP8L while the reference reads or
    writes outside its buffers and then does anything.  Everywhere else both must agree."""
    if rc_ref == O.REF_CRASHED:
        return rc != 0
    if rc_ref == 0:
        return (rc == 0 and np.array_equal(dec, dec_ref)) or rc == -5
    return rc == rc_ref or rc == -5


def test_verdicts_on_damaged_streams():
    rng = np.random.default_rng(23)
    streams = []
    for mode, w, h in ((3, 24, 16), (4, 40, 20), (5, 32, 32)):
        img = O.gen_image(mode, w, h, 99, 200)
        streams.append(O.encode(img, w, h)[1])
    img = O.gen_image(3, 32, 24, 5, 200)
    streams.append(O.encode_ext(img, 32, 24, predictor_bits=3, color_bits=2, use_subtract_green=1, cache_bits=4)[1])
    streams.append(O.encode_ext(img, 32, 24, meta_bits=2, meta_groups=3)[1])
    n = agree = exact = 0
    for s in streams:
        for _ in range(60):
            b = bytearray(s)
            kind = rng.integers(0, 3)
            if kind == 0:
                pos = int(rng.integers(4, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b = b[: int(rng.integers(5, len(b)))]
            else:
                pos = int(rng.integers(4, len(b)))
                b[pos] = int(rng.integers(0, 256))
            rc_ref, dec_ref, _, _ = O.ref_decode(bytes(b), 1 << 16)
            rc, dec, _, _ = O.decode(bytes(b))
            n += 1
            ok = _verdicts_agree(rc_ref, dec_ref, rc, dec)
            agree += ok
            exact += (rc == rc_ref) and (rc != 0 or np.array_equal(dec, dec_ref))
            assert ok, (len(s), kind, rc_ref, rc)
    assert n == 300 and agree == n
    assert exact > n // 2  # most damage is seen the same way by both
