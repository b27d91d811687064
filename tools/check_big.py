#!/usr/bin/env python3
"""BASELINE configs[2]: one 16383x16383 image (the largest VP8L size) through encode and decode on one GPU.
usage: check_big.py [MODE] [SIDE] [--no-decode]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
side = int(sys.argv[2]) if len(sys.argv) > 2 else 16383
w = h = side
webp_amd.init()
d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
webp_amd.generate_device(mode, w, h, 777, 200, d_in.data_ptr())
enc = webp_amd.Encoder()
cap = enc.bound(w, h)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(2):
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0, st
print("encode %dx%d mode %d: %.1f ms device, %.1f MP/s, %.2f bpp, stages(ms) %s, dense tiles %s" % (
    w, h, mode, 1e3 * enc.last_device_seconds(), w * h / 1e6 / enc.last_device_seconds(), 8.0 * lens[0] / (w * h),
    {k: round(1e3 * v, 2) for k, v in enc.last_stage_seconds().items()}, enc.last_dense_tiles()), flush=True)
if "--no-decode" not in sys.argv:
    dec = webp_amd.Decoder()
    d_dec = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    t = time.perf_counter()
    stt = dec.decode_device([d_out.data_ptr() + int(off[0])], [int(lens[0])], [w], [h], [d_dec.data_ptr()])
    torch.cuda.synchronize()
    print("decode: status %d, %.1f s, identical %s" % (stt[0], time.perf_counter() - t, bool(torch.equal(d_dec, d_in))), flush=True)
