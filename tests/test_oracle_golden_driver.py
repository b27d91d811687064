"""The oracle against what the reference's own compiled encoder and decoder drivers did
(webp/vp8l/vp8l.h:585-615, :330-351), recorded in tests/golden/ref_driver.json by
tests/golden/make_ref_driver_golden.py.  Runs everywhere (no reference needed)."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_driver.json")))


def test_streams_equal_the_reference_encoder(G):
    assert len(G["encode"]) >= 70
    for ent in G["encode"]:
        img, w, h = O.case_image(ent["image"])
        rc, data = O.encode(img, w, h)
        assert rc == ent["status"], ent["image"]
        if rc == 0:
            assert len(data) == ent["bytes"], ent["image"]
            assert hashlib.sha256(data).hexdigest() == ent["sha256"], ent["image"]


def test_streams_the_reference_decoder_read(G):
    """Each recorded hash is a stream that the reference's decoder turned back into the image: the
    oracle's test-vector encoder must still write exactly that stream, and its decoder read it."""
    assert len(G["decode"]) >= 80
    for ent in G["decode"]:
        img, w, h = O.case_image(ent["image"])
        rc, data = O.encode_ext(img, w, h, **ent["params"])
        assert rc == 0
        assert hashlib.sha256(data).hexdigest() == ent["sha256"], (ent["image"], ent["params"])
        rc, dec, dw, dh = O.decode(data)
        assert rc == 0 and (dw, dh) == (w, h) and np.array_equal(dec, img)


def test_verdicts_on_damaged_streams(G):
    """The oracle checks what the reference leaves unchecked (back references before the image,
    pixels beyond it: vp8l.h:298-300) and says InvalidVP8L there, where the reference reads or
    writes outside its buffers; everywhere else the verdict and the pixels are the reference's."""
    assert len(G["damaged"]) == 200
    same = 0
    for i, ent in enumerate(G["damaged"]):
        rc, dec, w, h = O.decode(bytes.fromhex(ent["stream_hex"]))
        ref = ent["status"]
        if ref == O.REF_CRASHED:
            assert rc != 0, i
        elif ref == 0:
            if rc == 0:
                assert (w, h) == (ent["w"], ent["h"]), i
                assert hashlib.sha256(dec.tobytes()).hexdigest() == ent["pixels_sha256"], i
                same += 1
            else:
                assert rc == -5, i
        else:
            assert rc in (ref, -5), i
            same += rc == ref
    assert same > 100
