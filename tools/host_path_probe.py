#!/usr/bin/env python3
"""Times the host-memory entry points (vp8l_encode_batch / vp8l_decode_batch): images in host memory in,
streams in host memory out, and back.  usage: host_path_probe.py [BATCH] [W] [H] [--pinned]
--pinned keeps the inputs in page-locked host memory (torch pinned tensors)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import webp_amd  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pinned = "--pinned" in sys.argv
    batch = int(args[0]) if len(args) > 0 else 64
    w = int(args[1]) if len(args) > 1 else 3840
    h = int(args[2]) if len(args) > 2 else 2160
    npx = w * h
    webp_amd.init()
    L = webp_amd.lib()
    d = torch.empty((batch, npx), dtype=torch.int32, device="cuda:0")
    for k in range(batch):
        webp_amd.generate_device(4 if k % 4 == 3 else 3, w, h, 12345 + k, 0, d[k].data_ptr(), device=0)
    host = torch.empty((batch, npx), dtype=torch.int32, pin_memory=pinned)
    host.copy_(d)
    torch.cuda.synchronize()
    del d
    ptrs = (C.c_void_p * batch)(*[host[k].data_ptr() for k in range(batch)])
    ws = (C.c_uint32 * batch)(*([w] * batch))
    hs = (C.c_uint32 * batch)(*([h] * batch))
    for rep in range(3):
        outs = (C.c_void_p * batch)()
        lens = (C.c_size_t * batch)()
        status = (C.c_int * batch)()
        t0 = time.perf_counter()
        rc = L.vp8l_encode_batch(0, batch, ptrs, ws, hs, outs, lens, status)
        dt = time.perf_counter() - t0
        assert rc == 0 and all(s == 0 for s in status), (rc, list(status))
        total = sum(lens)
        print("encode_batch %d x %dx%d (%s host memory): %.1f ms, %.0f MP/s, %.2f GB/s in, bpp %.2f"
              % (batch, w, h, "pinned" if pinned else "pageable", dt * 1e3, batch * npx / 1e6 / dt,
                 batch * npx * 4 / 1e9 / dt, 8.0 * total / (batch * npx)), flush=True)
        if rep < 2:
            for k in range(batch):
                L.vp8l_free(C.c_void_p(outs[k]))
    # decode the streams of the last pass
    for rep in range(0 if "--no-decode" in sys.argv else 2):
        argb = (C.c_void_p * batch)()
        ow = (C.c_uint32 * batch)()
        oh = (C.c_uint32 * batch)()
        t0 = time.perf_counter()
        rc = L.vp8l_decode_batch(0, batch, outs, lens, argb, ow, oh, status)
        dt = time.perf_counter() - t0
        assert rc == 0 and all(s == 0 for s in status), (rc, list(status))
        print("decode_batch: %.1f ms, %.0f MP/s, %.2f GB/s out" % (dt * 1e3, batch * npx / 1e6 / dt, batch * npx * 4 / 1e9 / dt), flush=True)
        if rep == 1:
            ok = True
            for k in range(0, batch, max(1, batch // 8)):
                got = np.ctypeslib.as_array(C.cast(argb[k], C.POINTER(C.c_uint32)), shape=(npx,))
                ok = ok and np.array_equal(got, host[k].numpy().view(np.uint32))
            print("round trip", "ok" if ok else "MISMATCH")
        for k in range(batch):
            L.vp8l_free(C.c_void_p(argb[k]))


if __name__ == "__main__":
    main()
