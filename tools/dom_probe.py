"""Development aid (GPU box): how the tiles of one image are searched and why the frequent-value kernel leaves some.
usage: python tools/dom_probe.py mode w h [predictor_bits] [ncolors]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd

mode, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pred = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ncol = int(sys.argv[5]) if len(sys.argv) > 5 else 0
webp_amd.init()
L = webp_amd.lib()
d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
webp_amd.generate_device(mode, w, h, 4242, ncol, d_in.data_ptr())
enc = webp_amd.Encoder()
if pred:
    enc.set_predictor_bits(pred)
cap = enc.bound(w, h)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
buf = (C.c_uint * 4)()
L.vp8l_debug_dominant_reasons(buf)
for rep in range(2):
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0
L.vp8l_debug_dominant_reasons(buf)
print("kinds", enc.last_tile_kinds(), "reasons over 2 calls [no frequent value, run starts, visits, pairs]", list(buf), "stage ms", {k: round(1e3 * v, 2) for k, v in enc.last_stage_seconds().items()})
if hasattr(L, "vp8l_debug_dominant_phase_cycles"):
    ph = (C.c_ulonglong * 12)()
    L.vp8l_debug_dominant_phase_cycles(ph)
    enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    L.vp8l_debug_dominant_phase_cycles(ph)
    names = ["load", "frequent value + segment pass", "buckets + run groups", "queries numbered", "walk", "offsets (wait for walks)", "pairs",
             "wait for pairs", "own matches", "v0 positions", "wait for v0 positions", "-"]
    tot = float(sum(ph)) or 1.0
    print("phases:", ", ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names, ph) if c))
if hasattr(L, "vp8l_debug_grams_phase_cycles"):
    ph = (C.c_ulonglong * 12)()
    L.vp8l_debug_grams_phase_cycles(ph)
    enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    L.vp8l_debug_grams_phase_cycles(ph)
    names = ["words numbered", "grams numbered", "lists", "search 16", "search 4", "search 2", "search 1", "wait 16", "wait 4", "wait 2", "wait 1", "-"]
    tot = float(sum(ph)) or 1.0
    print("grams phases:", ", ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names, ph) if c))
