#!/usr/bin/env python3
"""Known answer for the largest VP8L image (BASELINE configs[2]): size and sha256 of the ORACLE's stream for one
16383x16383 synthetic image (generator mode 4 "graphics", seed 4242), written to tests/golden/big16k.json.
The oracle's brute-force match search needs several minutes and about 6 GB for it, so this runs once, in the build
container; the GPU test compares the device encoder's stream with the recorded size and hash.
    python tests/golden/make_big16k.py [mode] [seed]"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
side = 16383
t0 = time.time()
img = O.gen_image(mode, side, side, seed, 200)
rc, data = O.encode(img, side, side)
assert rc == 0, rc
out = {"w": side, "h": side, "mode": mode, "seed": seed, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
       "pixels_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
       "source": "oracle/vp8l_oracle.c (restatement pinned against the reference's compiled components), %.0f s" % (time.time() - t0)}
path = os.path.join(HERE, "big16k.json")
cur = json.load(open(path)) if os.path.exists(path) else []
cur = [c for c in cur if (c["mode"], c["seed"]) != (mode, seed)] + [out]
json.dump(cur, open(path, "w"), indent=1)
print(out)
