#!/usr/bin/env python3
"""Per-phase cycle split of k_lz77_sparse on the benchmark's content.  Needs the measurement build of the library:
    make -C webp_amd/csrc libvp8l_amd_phases.so
    VP8L_AMD_LIB=$PWD/webp_amd/csrc/libvp8l_amd_phases.so python tools/phase_probe.py [BATCH]
Cycles are those of the first wave of each workgroup between phase marks, summed over workgroups."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import webp_amd

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = 3840, 2160
webp_amd.init()
L = webp_amd.lib()
d = torch.empty((batch, w * h), dtype=torch.int32, device="cuda")
for k in range(batch):
    webp_amd.generate_device(4 if k % 4 == 3 else 3, w, h, 12345 + k, 0, d[k].data_ptr())
enc = webp_amd.Encoder()
cap = enc.bound(w, h)
out = torch.empty(batch * cap, dtype=torch.uint8, device="cuda")
args = ([d[k].data_ptr() for k in range(batch)], [w] * batch, [h] * batch, out.data_ptr(), out.numel())
enc.encode_device(*args)
buf = (C.c_ulonglong * 8)()
L.vp8l_debug_phase_cycles(buf)
enc.encode_device(*args)
L.vp8l_debug_phase_cycles(buf)
names = ["load window (+ subtract green, store residual)", "hash, run starts, run lengths", "count buckets", "scan + scatter",
         "search (first wave)", "wait for the slowest wave", "store results", "-"]
tot = float(sum(buf[:7]))
res = {n: {"cycles": int(c), "share": round(c / tot, 4)} for n, c in zip(names[:7], buf[:7])}
res["match_stage_ms"] = 1e3 * enc.last_stage_seconds()["match"]
res["batch"] = batch
print(json.dumps(res, indent=1))
