#!/usr/bin/env python3
"""Makes tests/golden/libwebp_streams.npz: lossless VP8L streams written by libwebp (through Pillow) for small synthetic
images, and the pixels libwebp itself decodes from them.  Data only: the streams are libwebp's output for inputs generated
here.  Run in the build container (needs Pillow with WebP support):  python tests/golden/make_libwebp_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _webpstd as S  # noqa: E402

assert S.pillow() is not None
rng = np.random.default_rng(2024)
out = {}
for name, argb, w, h in S.test_images(rng):
    for tag, opts in (("fast", dict(quality=0, method=0)), ("best", dict(quality=100, method=6))):
        if w * h > 30000 and tag == "fast":
            continue
        f = S.libwebp_encode(argb, w, h, **opts)
        pixels, w0, h0 = S.libwebp_decode(f)
        assert (w0, h0) == (w, h)
        key = "%s_%s" % (name, tag)
        out[key + "/stream"] = np.frombuffer(S.vp8l_payload(f), dtype=np.uint8)
        out[key + "/argb"] = pixels
        out[key + "/size"] = np.array([w, h], dtype=np.uint32)
np.savez_compressed(os.path.join(HERE, "libwebp_streams.npz"), **out)
print(len(out) // 3, "streams,", os.path.getsize(os.path.join(HERE, "libwebp_streams.npz")), "bytes")
