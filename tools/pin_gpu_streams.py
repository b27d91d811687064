#!/usr/bin/env python3
"""Pins the GPU encoder's optional coders (predictor, cross-colour, colour cache, meta codes: what
the reference ENCODER cannot write) against the reference's own DECODER.

  on the GPU box :  python tools/pin_gpu_streams.py --emit     -> gpurun_out/gpu_ext_streams.npz
  in the container: python tools/pin_gpu_streams.py --check    -> tests/golden/gpu_ext_streams.json

--check decodes every stream with the reference's compiled VP8_LOSSLESS_DECODER
(oracle/_ref/libref_driver.so), requires the input image back, and records the stream's sha256;
tests/test_gpu_predictor.py::test_streams_the_reference_decoder_read then asserts on the GPU that
the encoder still writes exactly those streams."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (image descriptor for tests/_oracle.case_image, encode options)
CASES = []
for desc in (["gen", 3, 96, 64, 21], ["gen", 4, 128, 72, 22], ["gen", 3, 33, 18, 23], ["rand", 24, 16, 24], ["gen", 4, 260, 130, 25]):
    for opt in (dict(predictor_bits=2), dict(predictor_bits=4), dict(cross_color_bits=3),
                dict(predictor_bits=3, cross_color_bits=2), dict(cache_bits=4), dict(cache_bits=10),
                dict(meta_bits=2), dict(meta_bits=3, cache_bits=6), dict(predictor_bits=3, cross_color_bits=3, cache_bits=5, meta_bits=2)):
        CASES.append((desc, opt))
for desc in (["pal", 3, 64, 20, 31], ["pal", 12, 64, 20, 32], ["pal", 100, 64, 20, 33]):  # widths that fill their groups (transform.h:70-86)
    for opt in (dict(cache_bits=3), dict(meta_bits=2)):
        CASES.append((desc, opt))

NPZ = os.path.join(ROOT, "gpurun_out", "gpu_ext_streams.npz")
JSON = os.path.join(ROOT, "tests", "golden", "gpu_ext_streams.json")


def emit():
    import _oracle as O
    import webp_amd as W
    W.init()
    z = {}
    for i, (desc, opt) in enumerate(CASES):
        img, w, h = O.case_image(desc)
        z["s%d" % i] = np.frombuffer(W.encode(img, w, h, **opt), dtype=np.uint8)
    os.makedirs(os.path.dirname(NPZ), exist_ok=True)
    np.savez_compressed(NPZ, **z)
    print("wrote %s (%d streams)" % (NPZ, len(z)))


def check():
    import _oracle as O
    if O.ref_driver() is None:
        sys.exit("oracle/_ref/libref_driver.so missing: run `make -C oracle ref`")
    z = np.load(NPZ, allow_pickle=False)
    out = []
    for i, (desc, opt) in enumerate(CASES):
        data = z["s%d" % i].tobytes()
        img, w, h = O.case_image(desc)
        rc, dec, dw, dh = O.ref_decode(data, 1 << 18)
        assert rc == 0 and (dw, dh) == (w, h) and np.array_equal(dec, img), (desc, opt, rc)
        out.append({"image": desc, "options": opt, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()})
    with open(JSON, "w") as f:
        json.dump(out, f, indent=0)
    print("reference decoder read all %d streams; wrote %s" % (len(out), JSON))


if __name__ == "__main__":
    if sys.argv[1:] == ["--emit"]:
        emit()
    elif sys.argv[1:] == ["--check"]:
        check()
    else:
        sys.exit(__doc__)
