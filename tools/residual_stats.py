"""Development aid (GPU box): what the match search sees after the predictor transform.

Encodes one photo-like image with predictor tiles of 32, fetches the residual plane and prints, for sampled positions,
the best match's length and distance, how many of the 1024 distances start with an equal pixel, and the commonest values.
usage: python tools/residual_stats.py [mode] [w] [h]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
webp_amd.init()
d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
webp_amd.generate_device(mode, w, h, 12345, 200, d_in.data_ptr())
enc = webp_amd.Encoder()
enc.set_predictor_bits(5)
cap = enc.bound(w, h)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
assert st[0] == 0
res = enc.fetch_stage(0, webp_amd.STAGE_RESIDUAL)
m = enc.fetch_stage(0, webp_amd.STAGE_MATCH)
print("dense tiles", enc.last_dense_tiles(), "bpp", 8.0 * int(lens[0]) / (w * h))
M = res.size
vals, cnt = np.unique(res, return_counts=True)
o = np.argsort(-cnt)
print("distinct values", vals.size, "top:", [(hex(int(vals[i])), round(cnt[i] / M, 4)) for i in o[:8]])
runs = np.flatnonzero(np.diff(res) != 0)
print("mean run of equal pixels", M / (runs.size + 1))
ln = m & 0xFF
ds = (m >> 8) & 0x7FF
print("match length at every position: mean %.1f, share 0: %.3f, 1-3: %.3f, 4-15: %.3f, 16-127: %.3f, 128: %.3f" % (
    ln.mean(), (ln == 0).mean(), ((ln >= 1) & (ln <= 3)).mean(), ((ln >= 4) & (ln <= 15)).mean(), ((ln >= 16) & (ln < 128)).mean(), (ln == 128).mean()))
big = ln == 128
if big.any():
    dv, dc = np.unique(ds[big], return_counts=True)
    oo = np.argsort(-dc)
    print("distances of the 128-pixel matches:", [(int(dv[i]), round(dc[i] / big.sum(), 3)) for i in oo[:8]])
rng = np.random.default_rng(1)
ps = rng.integers(1100, M - 200, 2000)
eqc = []
longc = []
for p in ps:
    tgt = res[p:p + 128]
    src = np.lib.stride_tricks.sliding_window_view(res[p - 1024:p + 127], 128)[::-1][:1024]
    ne = src != tgt[None, :]
    run = np.where(ne.any(1), ne.argmax(1), 128)
    eqc.append(int((run > 0).sum()))
    longc.append(int((run >= 128).sum()))
print("distances with an equal first pixel: mean %.0f of 1024; with 128 equal pixels: mean %.0f" % (np.mean(eqc), np.mean(longc)))
