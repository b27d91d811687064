"""Size-independent properties at the benchmark configurations' image sizes (BASELINE configs 2-5), where the
CPU oracle would take minutes: encode -> decode round trips on the device, determinism, and agreement of the
two match-search kernels.  Byte parity with the oracle at small sizes is in test_gpu_encode.py."""
import numpy as np
import pytest

import webp_amd

pytestmark = pytest.mark.gpu


def gen(torch, plan, w, h):
    d = torch.empty((len(plan), w * h), dtype=torch.int32, device="cuda")
    for k, (mode, seed, ncol) in enumerate(plan):
        webp_amd.generate_device(mode, w, h, seed, ncol, d[k].data_ptr())
    return d


def encode(torch, d_in, w, h, exhaustive=False):
    n = d_in.shape[0]
    enc = webp_amd.Encoder()
    enc.set_exhaustive_search(exhaustive)
    cap = enc.bound(w, h)
    d_out = torch.empty(n * cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in[k].data_ptr() for k in range(n)], [w] * n, [h] * n, d_out.data_ptr(), d_out.numel())
    assert (st == 0).all()
    info = [enc.fetch_stage(k, webp_amd.STAGE_INFO) for k in range(n)]
    enc.close()
    return d_out, off, lens, info


def decode(torch, d_out, off, lens, w, h):
    n = len(off)
    dec = webp_amd.Decoder()
    d_dec = torch.zeros((n, w * h), dtype=torch.int32, device="cuda")
    st = dec.decode_device([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], [w] * n, [h] * n,
                           [d_dec[k].data_ptr() for k in range(n)])
    dec.close()
    assert (st == 0).all(), st
    return d_dec


def streams(d_out, off, lens):
    return [bytes(d_out[int(o): int(o) + int(l)].cpu().numpy()) for o, l in zip(off, lens)]


def test_batch_of_1080p_round_trip_and_determinism():   # configs[3] image size
    import torch
    w, h = 1920, 1080
    plan = [(3 + (k % 2), 100 + k, 0) for k in range(12)] + [(5, 7, 200), (5, 8, 16), (5, 9, 3), (5, 10, 2)]
    d_in = gen(torch, plan, w, h)
    d_out, off, lens, info = encode(torch, d_in, w, h)
    again = encode(torch, d_in, w, h)
    assert streams(d_out, off, lens) == streams(again[0], again[1], again[2])
    assert torch.equal(decode(torch, d_out, off, lens, w, h), d_in)
    assert [i["ncolors"] for i in info[12:]] == [200, 16, 3, 2] and [i["bits"] for i in info[12:]] == [0, 1, 2, 3]


def test_4k_streams_same_from_both_match_kernels():     # configs[1] image size
    import torch
    w, h = 3840, 2160
    d_in = gen(torch, [(3, 12345, 0), (4, 12348, 0)], w, h)
    a = encode(torch, d_in, w, h)
    b = encode(torch, d_in, w, h, exhaustive=True)
    assert streams(a[0], a[1], a[2]) == streams(b[0], b[1], b[2])
    assert torch.equal(decode(torch, a[0], a[1], a[2], w, h), d_in)


def test_palette_2048_round_trip():                      # configs[4] image size
    import torch
    w, h = 2048, 2048
    plan = [(5, 21, 256), (5, 22, 17), (5, 23, 5), (5, 24, 1)]
    d_in = gen(torch, plan, w, h)
    d_out, off, lens, info = encode(torch, d_in, w, h)
    assert [i["ncolors"] for i in info] == [256, 17, 5, 1]
    assert [i["xsize"] for i in info] == [2048, 2048, 1024, 256]
    assert torch.equal(decode(torch, d_out, off, lens, w, h), d_in)


@pytest.mark.parametrize("w,h", [(16383, 300), (300, 16383), (16384, 16)])
def test_largest_dimensions_round_trip(w, h):            # configs[2]: 16383 is the format's largest side
    import torch
    d_in = gen(torch, [(3, 5, 0), (4, 6, 0)], w, h)
    d_out, off, lens, info = encode(torch, d_in, w, h)
    assert all(i["M"] == w * h for i in info)
    assert torch.equal(decode(torch, d_out, off, lens, w, h), d_in)
