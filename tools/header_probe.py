"""Development aid (GPU box): encode stage times of one small image and of a batch of many small ones.
usage: python tools/header_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd
webp_amd.init()
for batch, w, h, mode, ncol in ((1, 512, 512, 3, 0), (1, 512, 512, 4, 0), (1, 512, 512, 5, 200), (1, 1920, 1080, 3, 0), (4096, 64, 64, 3, 0), (4096, 256, 256, 3, 0)):
    ins = []
    for k in range(batch):
        d = torch.empty(w * h, dtype=torch.int32, device="cuda")
        webp_amd.generate_device(mode, w, h, 12345 + k, ncol, d.data_ptr())
        ins.append(d)
    enc = webp_amd.Encoder()
    cap = enc.bound(w, h) * batch
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ptrs = [d.data_ptr() for d in ins]
    for rep in range(3):
        off, lens, st = enc.encode_device(ptrs, [w] * batch, [h] * batch, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(10):
        off, lens, st = enc.encode_device(ptrs, [w] * batch, [h] * batch, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 10
    print("%d x %dx%d mode %d: %.3f ms per call;" % (batch, w, h, mode, ms), {k: round(1e3 * v, 3) for k, v in enc.last_stage_seconds().items()}, enc.last_tile_kinds(), flush=True)
    enc.close()
