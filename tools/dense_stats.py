"""Development aid: statistics of the exhaustive match search on bundled palette content (CPU, the oracle's generator).

For sampled positions: the uncapped winner (largest first mismatch, farthest among equals), whether the limits (distance,
128 pixels) cut it, the match length, and how many of the 1024 cells hold an equal first pixel.
usage: python tools/dense_stats.py [ncolors] [size]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _oracle as O

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
img = O.gen_image(5, size, size, 777, ncol)
rc, data, tr = O.encode(img, size, size, trace=True)
res = tr["residual"]
M = res.size
print("colours", ncol, "coded pixels", M, "bits/pixel", 8.0 * len(data) / (size * size), "tokens", len(tr["tokens"]), "positions per token", M / len(tr["tokens"]))
rng = np.random.default_rng(1)
ps = rng.integers(1100, M - 200, 3000)
d = np.arange(1, 1025)
cut_d = cut_l = 0
lens = []
eqc = []
for p in ps:
    tgt = res[p:p + 128]
    src = np.lib.stride_tricks.sliding_window_view(res[p - 1024:p + 127], 128)[::-1][:1024]  # row k: distance k+1
    ne = src != tgt[None, :]
    run = np.where(ne.any(1), ne.argmax(1), 128)
    eqc.append(int((run > 0).sum()))
    best = run.max()
    far = d[run == best].max()
    if best > min(far, 128):
        if best > far: cut_d += 1
        else: cut_l += 1
    t = np.minimum(run, np.minimum(d, 128))
    lens.append(int(t.max()))
lens = np.array(lens); eqc = np.array(eqc)
print("winner cut by its distance %.3f, by 128 pixels %.3f" % (cut_d / len(ps), cut_l / len(ps)))
print("match length: mean %.1f median %d, share >= 128: %.3f" % (lens.mean(), np.median(lens), (lens >= 128).mean()))
print("cells with an equal first pixel: mean %.0f of 1024" % eqc.mean())
