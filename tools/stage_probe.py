#!/usr/bin/env python3
"""Per-stage device times of the encoder for one kind of synthetic content (development aid).

usage: stage_probe.py MODE [BATCH] [W] [H] [--exhaustive] [--predictor=BITS] [--cross=BITS] [--colors=N] [--dec-threads=N] [--whole-device=N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
mode = int(pos[0]); batch = int(pos[1]) if len(pos) > 1 else 64
w = int(pos[2]) if len(pos) > 2 else 3840; h = int(pos[3]) if len(pos) > 3 else 2160
ncolors = 200
for a in sys.argv:
    if a.startswith("--colors="):
        ncolors = int(a.split("=")[1])
webp_amd.init()
d_in = torch.empty((batch, w * h), dtype=torch.int32, device="cuda")
for k in range(batch):
    webp_amd.generate_device(mode, w, h, 12345 + k, ncolors, d_in[k].data_ptr())
enc = webp_amd.Encoder()
if "--exhaustive" in sys.argv:
    enc.set_exhaustive_search(True)
for a in sys.argv:
    if a.startswith("--predictor="):
        enc.set_predictor_bits(int(a.split("=")[1]))
    if a.startswith("--cross="):
        enc.set_cross_color_bits(int(a.split("=")[1]))
cap = enc.bound(w, h)
d_out = torch.empty(batch * cap, dtype=torch.uint8, device="cuda")
ptrs = [d_in[k].data_ptr() for k in range(batch)]
for it in range(3):
    off, lens, st = enc.encode_device(ptrs, [w] * batch, [h] * batch, d_out.data_ptr(), d_out.numel())
assert (st == 0).all()
ss = enc.last_stage_seconds()
mp = batch * w * h / 1e6
print("mode %d batch %d %dx%d: %.1f MP/s, bpp %.2f, dense tiles %s" % (mode, batch, w, h, mp / enc.last_device_seconds(), 8.0 * lens.sum() / (batch * w * h), enc.last_dense_tiles()))
print("  " + "  ".join("%s %.2f" % (k, 1e3 * v) for k, v in ss.items()) + "  (ms)")
print("  ns/px: " + "  ".join("%s %.3f" % (k, 1e9 * v / (mp * 1e6)) for k, v in ss.items()))

dec = webp_amd.Decoder()
for a in sys.argv:
    if a.startswith("--dec-threads="):
        dec.set_threads_per_stream(int(a.split("=")[1]))
    if a.startswith("--whole-device="):
        dec.set_whole_device(int(a.split("=")[1]))
d_dec = torch.empty((batch, w * h), dtype=torch.int32, device="cuda")
args = ([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], [w] * batch, [h] * batch, [d_dec[k].data_ptr() for k in range(batch)])
for it in range(2):
    stt = dec.decode_device(*args)
assert (stt == 0).all() and torch.equal(d_dec, d_in)
print("  decode: %.1f MP/s (device time %.1f ms, %d streams by the whole device)" % (mp / dec.last_device_seconds(), 1e3 * dec.last_device_seconds(), dec.last_whole_device_streams()))
L = webp_amd.lib()
if hasattr(L, "vp8l_debug_phase_cycles"):  # the measurement build: cycles between the phase marks of the candidate-list kernels
    import ctypes as C
    buf = (C.c_ulonglong * 8)()
    L.vp8l_debug_phase_cycles(buf)
    enc.encode_device(ptrs, [w] * batch, [h] * batch, d_out.data_ptr(), d_out.numel())
    L.vp8l_debug_phase_cycles(buf)
    tot = float(sum(buf)) or 1.0
    print("  phase shares: " + "  ".join("%d: %.3f" % (k, buf[k] / tot) for k in range(8)))
