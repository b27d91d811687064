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


def _big_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big16k.json")
    return json.load(open(path))


@pytest.mark.parametrize("case", _big_cases(), ids=lambda c: "mode%d" % c["mode"])
def test_single_16383x16383_image_against_the_oracles_known_answer(case):   # configs[2] as BASELINE names it
    """One image of the largest VP8L size on one GPU.  Default path: the stream's size and sha256 equal what the CPU
    oracle produced for the same pixels (tests/golden/make_big16k.py, run once in the build container: minutes of
    brute-force match search), and the stream decodes back on the device.  With the predictor transform (tile-parallel
    mode search + residuals in front of the same LZ77 coder): decodes back to the input."""
    import hashlib
    import torch
    w, h = case["w"], case["h"]
    d_in = gen(torch, [(case["mode"], case["seed"], 0)], w, h)
    assert hashlib.sha256(d_in[0].cpu().numpy().tobytes()).hexdigest() == case["pixels_sha256"]   # same pixels as the oracle's generator
    d_out, off, lens, info = encode(torch, d_in, w, h)
    assert info[0]["M"] == w * h and info[0]["ncolors"] == 0
    assert int(lens[0]) == case["bytes"]
    (s,) = streams(d_out, off, lens)
    assert hashlib.sha256(s).hexdigest() == case["sha256"]
    del s
    assert torch.equal(decode(torch, d_out, off, lens, w, h), d_in)
    del d_out
    # predictor + LZ77 (not the reference encoder's bytes: checked by decoding)
    enc = webp_amd.Encoder()
    enc.set_predictor_bits(5)
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d_in[0].data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert st[0] == 0
    modes = enc.fetch_stage(0, webp_amd.STAGE_PRED_MODES)
    enc.close()
    assert modes is not None and modes["tx"] * modes["ty"] <= 1024 and (modes["tx"] << modes["bits"]) >= w
    assert int(lens[0]) < case["bytes"]          # the predictor pays on both generators
    assert torch.equal(decode(torch, d_out, off, lens, w, h), d_in)


def test_chain_walk_of_an_image_of_many_superblocks_against_the_oracle():
    """An image of more than 1024 superblocks of 16384 positions has its parse's chain walked by a workgroup per round of 128
    superblocks (two launches) instead of one workgroup per image; any walk gives a stream that decodes, only the right
    one gives the oracle's bytes.  In one call with a small image, so that the rounds of two images lie side by side."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle as O
    sizes = [(4200, 4100), (640, 480)]
    imgs = [O.gen_image(4, w, h, 31 + k, 200) for k, (w, h) in enumerate(sizes)]
    d_in = [torch.from_numpy(a.view(np.int32)).cuda() for a in imgs]
    enc = webp_amd.Encoder()
    caps = [enc.bound(w, h) for w, h in sizes]
    d_out = torch.empty(sum(caps), dtype=torch.uint8, device="cuda")
    off, lens, st = enc.encode_device([d.data_ptr() for d in d_in], [w for w, _ in sizes], [h for _, h in sizes], d_out.data_ptr(), d_out.numel())
    enc.close()
    assert (st == 0).all()
    for k, (w, h) in enumerate(sizes):
        rc, ref = O.encode(imgs[k], w, h)
        assert rc == 0
        assert bytes(d_out[int(off[k]): int(off[k]) + int(lens[k])].cpu().numpy()) == ref, (w, h)
