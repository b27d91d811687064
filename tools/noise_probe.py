#!/usr/bin/env python3
"""Encode speed with the predictor transform on photo-like content with added sensor-like noise (development aid).

The synthetic photo-like generator is a smooth ramp with very little noise; after prediction its residuals are almost all
0 / +-1, which is the dense case of the match search.  Real photographs keep several bits of noise per channel.
usage: noise_probe.py [BATCH] [AMPLITUDE ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
amps = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 4, 8, 16]
w, h = 3840, 2160
webp_amd.init()
base = torch.empty((batch, w * h), dtype=torch.int32, device="cuda")
for k in range(batch):
    webp_amd.generate_device(3, w, h, 12345 + k, 200, base[k].data_ptr())
g = torch.Generator(device="cuda")
g.manual_seed(7)
for amp in amps:
    d_in = base.clone()
    if amp:
        b = d_in.view(torch.uint8).view(batch, w * h, 4).to(torch.int16)
        noise = torch.randint(-amp, amp + 1, b.shape, generator=g, device="cuda", dtype=torch.int16)
        noise[..., 3] = 0  # alpha stays opaque
        d_in = (b + noise).clamp_(0, 255).to(torch.uint8).view(batch, w * h * 4).view(torch.int32).contiguous()
        del b, noise
    for pred in (0, 5):
        enc = webp_amd.Encoder()
        if pred:
            enc.set_predictor_bits(pred)
        cap = enc.bound(w, h)
        d_out = torch.empty(batch * cap, dtype=torch.uint8, device="cuda")
        ptrs = [d_in[k].data_ptr() for k in range(batch)]
        for it in range(2):
            off, lens, st = enc.encode_device(ptrs, [w] * batch, [h] * batch, d_out.data_ptr(), d_out.numel())
        assert (st == 0).all()
        ss = enc.last_stage_seconds()
        print("noise +-%-2d predictor %d: %6.2f bits/pixel  %7.1f GP/s  match %6.2f ms (fallback %6.2f)  dense tiles %s" % (
            amp, pred, 8.0 * lens.sum() / (batch * w * h), batch * w * h / 1e9 / enc.last_device_seconds(),
            1e3 * ss["match"], 1e3 * ss["match_fallback"], enc.last_dense_tiles()), flush=True)
        enc.close()
        del d_out
