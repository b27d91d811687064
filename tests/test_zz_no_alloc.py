"""vp8l_encode_device_noalloc (workspace in a static device array, no allocation) against
vp8l_encoder_encode_device (workspace from hipMalloc): the stream bytes must be identical, bit for bit, on two calls
in a row, the second with a smaller image (the shared array keeps the first call's contents), and identical to the
oracle's.  Also: sizes beyond the stated maximum are refused, and the fully asynchronous form (result read from device
memory) gives the same bytes."""
import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def _both(torch, enc, img, w, h):
    d_in = torch.from_numpy(img.view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_ref = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    off, lens, status = enc.encode_device([d_in.data_ptr()], [w], [h], d_ref.data_ptr(), cap)
    assert status[0] == 0
    ref = bytes(d_ref[int(off[0]):int(off[0]) + int(lens[0])].cpu().numpy())
    d_out = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")   # the entry point must not rely on a cleared buffer
    torch.cuda.synchronize()
    n = webp_amd.encode_device_noalloc(d_in.data_ptr(), w, h, d_out.data_ptr(), cap)
    got = bytes(d_out[:n].cpu().numpy())
    return ref, got


def test_noalloc_equals_context_encoder_twice_second_smaller():
    import torch
    assert torch.cuda.is_available()
    webp_amd.init()
    enc = webp_amd.Encoder()
    # first call: a photo-like image (subtract-green path, candidate lists); second, smaller: a palette image with
    # bundling (colour indexing path) -- the array still holds the first call's planes, tables and histograms
    for mode, w, h, ncol in ((3, 640, 360, 0), (5, 211, 97, 13), (4, 97, 31, 0)):
        img = O.gen_image(mode, w, h, 777, ncol or 200)
        ref, got = _both(torch, enc, img, w, h)
        assert len(got) == len(ref)
        assert got == ref
        rc, orc = O.encode(img, w, h)
        assert rc == 0 and got == orc


def test_noalloc_largest_size_and_limit():
    import torch
    webp_amd.init()
    enc = webp_amd.Encoder()
    w, h = 1920, 1080   # fits VP8L_NOALLOC_MAX_PIXELS
    assert w * h <= webp_amd.NOALLOC_MAX_PIXELS
    d_in = torch.empty(w * h, dtype=torch.int32, device="cuda")
    webp_amd.generate_device(4, w, h, 99, 0, d_in.data_ptr())
    cap = enc.bound(w, h)
    d_ref = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    off, lens, status = enc.encode_device([d_in.data_ptr()], [w], [h], d_ref.data_ptr(), cap)
    assert status[0] == 0
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    n = webp_amd.encode_device_noalloc(d_in.data_ptr(), w, h, d_out.data_ptr(), cap)
    assert n == int(lens[0])
    assert torch.equal(d_out[:n], d_ref[:n])
    # one pixel row more than the maximum: refused before anything is launched
    w2, h2 = 2048, 1025
    d_big = torch.zeros(w2 * h2, dtype=torch.int32, device="cuda")
    cap2 = enc.bound(w2, h2)
    d_out2 = torch.zeros(cap2, dtype=torch.uint8, device="cuda")
    with pytest.raises(webp_amd.Vp8lError) as ei:
        webp_amd.encode_device_noalloc(d_big.data_ptr(), w2, h2, d_out2.data_ptr(), cap2)
    assert ei.value.status == -8
    # too small an output buffer, bad arguments
    with pytest.raises(webp_amd.Vp8lError) as ei:
        webp_amd.encode_device_noalloc(d_in.data_ptr(), w, h, d_out.data_ptr(), cap - 256)
    assert ei.value.status == -8
    with pytest.raises(webp_amd.Vp8lError) as ei:
        webp_amd.encode_device_noalloc(d_in.data_ptr(), 0, h, d_out.data_ptr(), cap)
    assert ei.value.status == -1


def test_noalloc_async_result_in_device_memory():
    import torch
    webp_amd.init()
    w, h = 300, 200
    img = O.gen_image(3, w, h, 5, 200)
    rc, ref = O.encode(img, w, h)
    assert rc == 0
    d_in = torch.from_numpy(img.view(np.int32)).cuda()
    cap = webp_amd.Encoder.bound(w, h)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_res = torch.full((2,), -1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    webp_amd.encode_device_noalloc(d_in.data_ptr(), w, h, d_out.data_ptr(), cap, stream=s.cuda_stream,
                                   d_result_ptr=d_res.data_ptr(), wait=False)
    s.synchronize()
    res = d_res.cpu().numpy()
    assert res[0] == 0 and res[1] == len(ref)
    assert bytes(d_out[:len(ref)].cpu().numpy()) == ref
