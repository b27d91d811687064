"""Development aid (GPU box): saves what the match search sees (the residual plane) and what it answers (length and
distance at every position) for a few kinds of dense content, for study on the CPU.
usage: python tools/dump_residuals.py -> gpurun_out/residuals.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

webp_amd.init()
z = {}
for name, mode, w, h, pred, ncol in (("graphics_pred", 4, 1920, 1080, 5, 0), ("photo_pred", 3, 1920, 1080, 5, 0),
                                     ("pal2", 5, 2048, 512, 0, 2), ("pal4", 5, 2048, 512, 0, 4), ("pal16", 5, 2048, 512, 0, 16)):
    d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
    webp_amd.generate_device(mode, w, h, 4242, ncol, d_in.data_ptr())
    enc = webp_amd.Encoder()
    if pred:
        enc.set_predictor_bits(pred)
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0
    info = enc.fetch_stage(0, webp_amd.STAGE_INFO)
    z[name + "_res"] = enc.fetch_stage(0, webp_amd.STAGE_RESIDUAL)
    z[name + "_match"] = enc.fetch_stage(0, webp_amd.STAGE_MATCH)
    z[name + "_info"] = np.array([w, h, info["xsize"], info["M"]], dtype=np.int64)
    print(name, info, "dense tiles", enc.last_dense_tiles(), "bpp", 8.0 * int(lens[0]) / (w * h))
    enc.close()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/residuals.npz", **z)
