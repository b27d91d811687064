"""The host-memory batch entry points with many small chunks and few lanes (set through the environment,
which the library reads when it is loaded: hence a child process): results must equal the oracle's image by
image, invalid images and damaged streams keep their own status wherever they fall in a chunk."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import sys
    import numpy as np
    sys.path.insert(0, %r)
    sys.path.insert(0, %r)
    import _oracle as O
    import webp_amd

    rng = np.random.default_rng(4)
    imgs = []
    for k in range(60):
        if k %% 11 == 5:
            imgs.append((np.zeros(0, np.uint32), 0, 7))            # invalid: reported, the others unaffected
        elif k %% 13 == 7:
            imgs.append((np.zeros(4, np.uint32), 16385, 1))        # too big
        else:
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
            imgs.append((O.gen_image((3, 4, 5)[k %% 3], w, h, 100 + k, 40), w, h))
    res = webp_amd.encode_batch(imgs)
    streams = []
    for (img, w, h), (st, data) in zip(imgs, res):
        rc, ref = O.encode(img, w, h)
        assert st == rc, (st, rc, w, h)
        if rc == 0:
            assert data == ref, (w, h)
            streams.append((data, img, w, h))
    assert len(streams) > 40
    # decode: damage every seventh stream
    inputs = []
    for k, (data, img, w, h) in enumerate(streams):
        inputs.append(data[: len(data) // 2] if k %% 7 == 3 else data)
    inputs.insert(5, b"")
    inputs.insert(9, b"\\0\\0\\0\\0\\x2e\\0\\0\\0\\0")
    out = webp_amd.decode_batch(inputs)
    for s, (st, arr, w, h) in zip(inputs, out):
        rc0, ref, w0, h0 = O.decode(s)
        assert st == rc0, (st, rc0, len(s))
        if rc0 == 0:
            assert (w, h) == (w0, h0) and np.array_equal(arr, ref)
    print("pipeline ok", len(streams))
""")


def test_many_chunks_few_lanes():
    env = dict(os.environ, VP8L_HOST_LANES="3", VP8L_HOST_CHUNK_MB="1", VP8L_HOST_DECODE_STREAMS="5")
    code = CHILD % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pipeline ok" in r.stdout


CHILD_LARGE = textwrap.dedent("""
    import sys
    import numpy as np
    sys.path.insert(0, %r)
    sys.path.insert(0, %r)
    import _oracle as O
    import webp_amd

    # streams of more than a MiB each: a decode chunk closes by its pixel bytes as soon as it holds a few MiB of streams
    imgs = [(O.gen_image(3, 1000, 700, 300 + k), 1000, 700) for k in range(12)]
    res = webp_amd.encode_batch(imgs)
    assert all(st == 0 for st, _ in res)
    streams = [data for _, data in res]
    assert sum(len(s) for s in streams[:5]) > (4 << 20)
    streams[7] = streams[7][: len(streams[7]) // 3]   # one damaged stream in the middle of a chunk
    out = webp_amd.decode_batch(streams)
    for k, ((st, arr, w, h), (img, w0, h0)) in enumerate(zip(out, imgs)):
        if k == 7:
            rc0, ref, w1, h1 = O.decode(streams[7])   # the oracle's verdict on the damaged stream, whatever it is
            assert st == rc0
            assert rc0 != 0 or ((w, h) == (w1, h1) and np.array_equal(arr, ref))
        else:
            assert st == 0 and (w, h) == (w0, h0) and np.array_equal(arr, img), k
    print("large ok")
""")


def test_large_streams_chunked_by_pixel_bytes():
    env = dict(os.environ, VP8L_HOST_LANES="2", VP8L_HOST_CHUNK_MB="4")
    code = CHILD_LARGE % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "large ok" in r.stdout


def test_batch_dealt_out_to_a_device_list():
    """vp8l_encode_batch_devices / vp8l_decode_batch_devices: the device list may name one GPU several times (this box has
    one), which runs every block on its own host thread against the same device: every image is encoded exactly once, in
    its place, with the oracle's bytes, and decodes back."""
    import numpy as np
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    import webp_amd
    rng = np.random.default_rng(4)
    images = []
    for k in range(11):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        images.append((O.gen_image(3 + k % 3, w, h, 50 + k, 40), w, h))
    images.insert(4, (np.zeros(4, dtype=np.uint32), 16385, 1))   # refused by itself, inside a block
    res = webp_amd.encode_batch(images, devices=[0, 0, 0])
    assert res[4][0] == -2
    streams = []
    for k, ((st, data), (img, w, h)) in enumerate(zip(res, images)):
        if k == 4:
            continue
        rc, ref = O.encode(img, w, h)
        assert st == rc == 0 and data == ref, k
        streams.append(data)
    streams.insert(2, b"\0\0\0\0\x2e\0\0\0\0")
    out = webp_amd.decode_batch(streams, devices=[0, 0])
    assert out[2][0] == -4
    good = [x for k, x in enumerate(images) if k != 4]
    for (st, arr, w, h), (img, w0, h0) in zip([o for k, o in enumerate(out) if k != 2], good):
        assert st == 0 and (w, h) == (w0, h0) and np.array_equal(arr, img)
    with pytest.raises(webp_amd.Vp8lError):
        webp_amd.encode_batch(images[:2], devices=[99])
