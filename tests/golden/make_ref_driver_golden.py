#!/usr/bin/env python3
"""Records tests/golden/ref_driver.json from the reference's own compiled encoder and decoder
DRIVERS (oracle/_ref/libref_driver.so, built by `make -C oracle ref` from /root/reference in
place, against an installed libpng header).  Build container only:
    python tests/golden/make_ref_driver_golden.py

Data only (inputs as small descriptors, outputs as sizes and hashes):
  encode   image descriptor -> size and sha256 of the reference encoder's stream
  decode   image descriptor + the oracle's test-vector-encoder parameters -> sha256 of the stream
           that the reference DECODER turned back into the image (transform list, sub-images,
           colour cache, meta codes: what the reference encoder cannot write)
  damaged  a damaged stream (hex) -> the reference decoder's verdict and, when it decoded, the
           sha256 of its pixels
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from test_oracle_ext import CASES as EXT_CASES  # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    if O.ref_driver() is None:
        sys.exit("oracle/_ref/libref_driver.so missing: run `make -C oracle ref` in the build container")
    out = {"encode": [], "decode": [], "damaged": []}

    descs = []
    for mode in (3, 4, 5):
        for (w, h) in ((1, 1), (2, 1), (1, 2), (7, 5), (61, 19), (130, 3), (3, 130), (96, 64), (256, 256)):
            descs.append(["gen", mode, w, h, 12345 + w])
    for ncol in (1, 2, 3, 4, 5, 16, 17, 200, 256, 257):
        for (w, h) in ((1, 1), (9, 4), (63, 17), (128, 40)):
            descs.append(["pal", ncol, w, h, 100 + ncol])
    for (w, h) in ((1, 1), (5, 3), (40, 40)):
        descs.append(["rand", w, h, 3])
    for d in descs:
        img, w, h = O.case_image(d)
        rc, data = O.ref_encode(img, w, h)
        ent = {"image": d, "status": rc}
        if rc == 0:
            ent["bytes"] = len(data)
            ent["sha256"] = sha(data)
        out["encode"].append(ent)

    for prm in EXT_CASES:
        for d in (["gen", 3, 33, 18, 7], ["gen", 4, 64, 64, 7], ["gen", 5, 40, 24, 7], ["rand", 7, 5, 7],
                  ["pal", 3, 16, 9, 7], ["pal", 12, 24, 9, 7], ["pal", 100, 31, 9, 7]):
            img, w, h = O.case_image(d)
            if prm.get("use_palette"):
                if np.unique(img).size > 256 or w % 8:
                    continue  # transform.h:70-86 overruns rows that the width does not fill
            rc, data = O.encode_ext(img, w, h, **prm)
            assert rc == 0
            rc, dec, dw, dh = O.ref_decode(data, 1 << 16)
            assert rc == 0 and (dw, dh) == (w, h) and np.array_equal(dec, img), (prm, d, rc)
            out["decode"].append({"image": d, "params": prm, "bytes": len(data), "sha256": sha(data)})

    rng = np.random.default_rng(23)
    streams = []
    for d in (["gen", 3, 24, 16, 99], ["gen", 4, 40, 20, 99], ["gen", 5, 32, 32, 99]):
        img, w, h = O.case_image(d)
        streams.append(O.encode(img, w, h)[1])
    img, w, h = O.case_image(["gen", 3, 32, 24, 5])
    streams.append(O.encode_ext(img, w, h, predictor_bits=3, color_bits=2, use_subtract_green=1, cache_bits=4)[1])
    streams.append(O.encode_ext(img, w, h, meta_bits=2, meta_groups=3)[1])
    for s in streams:
        for _ in range(40):
            b = bytearray(s)
            kind = int(rng.integers(0, 3))
            if kind == 0:
                b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b = b[: int(rng.integers(5, len(b)))]
            else:
                b[int(rng.integers(4, len(b)))] = int(rng.integers(0, 256))
            rc, dec, dw, dh = O.ref_decode(bytes(b), 1 << 16)
            ent = {"stream_hex": bytes(b).hex(), "status": rc}
            if rc == 0:
                ent["w"], ent["h"], ent["pixels_sha256"] = dw, dh, sha(dec.tobytes())
            out["damaged"].append(ent)

    with open(os.path.join(HERE, "ref_driver.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote ref_driver.json: %d encode, %d decode, %d damaged" % tuple(len(out[k]) for k in ("encode", "decode", "damaged")))


if __name__ == "__main__":
    main()
