#!/usr/bin/env python3
"""Encode/decode speed and size of the headline content with the optional coder features.
usage: ext_probe.py [BATCH]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import webp_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w, h = 3840, 2160
webp_amd.init()
d = torch.empty((batch, w * h), dtype=torch.int32, device="cuda")
for k in range(batch):
    webp_amd.generate_device(4 if k % 4 == 3 else 3, w, h, 12345 + k, 0, d[k].data_ptr())
for name, cache, meta, groups, pred in (("default", 0, 0, 4, 0), ("meta6x4", 0, 6, 4, 0), ("meta5x8", 0, 5, 8, 0), ("meta4x8", 0, 4, 8, 0),
                                        ("pred5", 0, 0, 4, 5), ("pred5+meta6x8", 0, 6, 8, 5), ("cache10", 10, 0, 4, 0)):
    enc = webp_amd.Encoder()
    enc.set_cache_bits(cache); enc.set_meta(meta, groups); enc.set_predictor_bits(pred)
    cap = enc.bound(w, h)
    out = torch.empty(batch * cap, dtype=torch.uint8, device="cuda")
    args = ([d[k].data_ptr() for k in range(batch)], [w] * batch, [h] * batch, out.data_ptr(), out.numel())
    enc.encode_device(*args)
    off, lens, st = enc.encode_device(*args)
    assert (st == 0).all()
    es = enc.last_device_seconds()
    stages = {k: round(1e3 * v, 2) for k, v in enc.last_stage_seconds().items()}
    dec = webp_amd.Decoder()
    back = torch.zeros((batch, w * h), dtype=torch.int32, device="cuda")
    dargs = ([out.data_ptr() + int(o) for o in off], [int(x) for x in lens], [w] * batch, [h] * batch, [back[k].data_ptr() for k in range(batch)])
    dec.decode_device(*dargs)
    st = dec.decode_device(*dargs)
    ds = dec.last_device_seconds()
    ok = bool((st == 0).all() and torch.equal(back, d))
    print("%-24s bpp %.3f  encode %.1f ms (%.1f GP/s)  decode %.1f ms (%.1f GP/s) whole-device streams %d  ok %s  stages %s" % (
        name, 8.0 * float(lens.sum()) / (batch * w * h), 1e3 * es, batch * w * h / es / 1e9, 1e3 * ds, batch * w * h / ds / 1e9,
        dec.last_whole_device_streams(), ok, stages), flush=True)
    enc.close(); dec.close()
