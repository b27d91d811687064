import sys, os
sys.path.insert(0, os.getcwd())
import torch, webp_amd
webp_amd.init()
for (w,h,n) in ((64,64,4096),(640,360,3584),(256,256,4096)):
    d_in = torch.empty((n, w*h), dtype=torch.int32, device="cuda")
    for k in range(n): webp_amd.generate_device(3 if k%4 else 4, w, h, 100+k, 200, d_in[k].data_ptr())
    enc = webp_amd.Encoder(); cap = enc.bound(w,h); d_out = torch.empty(n*cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in[k].data_ptr() for k in range(n)], [w]*n, [h]*n, d_out.data_ptr(), d_out.numel())
    dec = webp_amd.Decoder(); d_dec = torch.empty_like(d_in)
    args = ([d_out.data_ptr()+int(o) for o in off], [int(x) for x in lens], [w]*n, [h]*n, [d_dec[k].data_ptr() for k in range(n)])
    for it in range(3): s2 = dec.decode_device(*args)
    assert (s2==0).all() and torch.equal(d_dec, d_in)
    t = dec.last_device_seconds()
    print("%dx%d x%d: decode %.2f ms (%.1f MP/s), per stream slot %.1f us, encode %.2f ms" % (w,h,n,1e3*t, n*w*h/1e6/t, 1e6*t/(n/256.0), 1e3*enc.last_device_seconds()), flush=True)
    enc.close(); del d_in, d_out, d_dec
