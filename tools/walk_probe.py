"""Development aid (GPU box): the match stage of batches of palette images of one colour count each.
usage: python tools/walk_probe.py [batch] [size] [match mode: 4 = search along the parse, 0 = default]
With the measurement build (VP8L_AMD_LIB=.../libvp8l_amd_phases.so) the walk kernel's counters are printed too."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = h = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 4
webp_amd.init()
L = webp_amd.lib()
for ncol in (2, 4, 16):
    ins = []
    for k in range(batch):
        d = torch.empty(w * h, dtype=torch.int32, device="cuda")
        webp_amd.generate_device(5, w, h, 777 + k, ncol, d.data_ptr())
        ins.append(d)
    enc = webp_amd.Encoder()
    enc.set_match_search(mode)
    cap = enc.bound(w, h) * batch
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        off, lens, st = enc.encode_device([d.data_ptr() for d in ins], [w] * batch, [h] * batch, d_out.data_ptr(), cap)
        assert all(s == 0 for s in st)
    ms = {k: round(1e3 * v, 2) for k, v in enc.last_stage_seconds().items()}
    print(ncol, "colours:", enc.last_tile_kinds(), "match ms", ms.get("match"), "of which fallback", ms.get("match_fallback"), flush=True)
    if hasattr(L, "vp8l_debug_walk_stats"):
        st8 = (C.c_ulonglong * 8)()
        L.vp8l_debug_walk_stats(st8)
        enc.encode_device([d.data_ptr() for d in ins], [w] * batch, [h] * batch, d_out.data_ptr(), cap)
        L.vp8l_debug_walk_stats(st8)
        t, steps, rounds, live, ext, gone = [int(x) for x in st8[:6]]
        if t:
            print("   per tile: %.0f positions searched, %.0f rounds (%.0f with a candidate), %.0f extension steps; %d of %d tiles given up" % (
                steps / t, rounds / t, live / t, ext / t, gone, t), flush=True)
    enc.close()
