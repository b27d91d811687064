"""Development aid (GPU box): one 512x512 stream decoded in each form of the decoder (device-resident, mean of 20 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import webp_amd
webp_amd.init()
for mode, w, h in ((3, 512, 512), (4, 512, 512), (3, 1024, 1024), (3, 1920, 1080)):
    d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
    webp_amd.generate_device(mode, w, h, 12345, 0, d_in.data_ptr())
    enc = webp_amd.Encoder()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    for name, setup in (("default", lambda d: None), ("whole device", lambda d: d.set_whole_device(2)), ("workgroup", lambda d: d.set_whole_device(1)),
                        ("wave", lambda d: (d.set_whole_device(1), d.set_threads_per_stream(64)))):
        dec = webp_amd.Decoder()
        setup(dec)
        d_dec = torch.empty(w * h, dtype=torch.int32, device="cuda")
        args = ([d_out.data_ptr() + int(off[0])], [int(lens[0])], [w], [h], [d_dec.data_ptr()])
        for _ in range(3):
            s = dec.decode_device(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s = dec.decode_device(*args)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        print("mode %d %dx%d (%d bytes) %-12s %.3f ms wall, %.3f ms kernels, ok %s, whole-device streams %d" % (
            mode, w, h, int(lens[0]), name, ms, 1e3 * dec.last_device_seconds(), bool(s[0] == 0 and torch.equal(d_dec, d_in)), dec.last_whole_device_streams()))
