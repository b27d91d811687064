"""The command line tool (tools/webp_, mirror of the reference's webp.cpp main) end to end on the GPU:
PNG -> .webp (RIFF container around the stream the oracle gives for the same pixels) -> PNG."""
import os
import struct
import subprocess

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "webp_")


# The reference's own webp.cpp, compiled unchanged against include/compat/webp/webp.h (the mirror) and linked with the
# GPU library by oracle/Makefile where the reference checkout exists; the binary travels to the GPU box.
REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "webp_ref_main")
TOOLS = [CLI] + ([REF_MAIN] if os.path.exists(REF_MAIN) else [])


def run(*args, tool=CLI):
    return subprocess.run([tool, *args], capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("tool", TOOLS, ids=[os.path.basename(t) for t in TOOLS])
@pytest.mark.parametrize("kind", ["RGB", "RGBA", "P"])
def test_encode_decode_files(tmp_path, kind, tool):
    from PIL import Image
    assert os.path.exists(tool), "tools/webp_ not built"
    w, h = 200, 77
    argb = O.gen_image(3 if kind != "P" else 5, w, h, 9, 40)
    rgb = np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], axis=-1).astype(np.uint8).reshape(h, w, 3)
    if kind == "RGB":
        im = Image.fromarray(rgb, "RGB")
        want = argb | np.uint32(0xFF000000)
    elif kind == "RGBA":
        # not a uniformly used alpha channel: 256 codes of length 8 is the stream the reference itself cannot read back
        a = (rgb[..., 1:2] // 2 + 60).astype(np.uint8)
        rgba = np.concatenate([rgb, a], axis=-1)
        im = Image.fromarray(rgba, "RGBA")
        r4 = rgba.astype(np.uint32)
        want = ((r4[..., 0] << 24) | (r4[..., 1] << 16) | (r4[..., 2] << 8) | r4[..., 3]).reshape(-1)  # webp.cpp:98-103
    else:
        im = Image.fromarray(rgb, "RGB").quantize(40)
        q = np.asarray(im.convert("RGB")).astype(np.uint32)
        want = (np.uint32(0xFF000000) | (q[..., 0] << 16) | (q[..., 1] << 8) | q[..., 2]).reshape(-1)
    src, mid, dst = str(tmp_path / "in.png"), str(tmp_path / "out.webp"), str(tmp_path / "back.png")
    im.save(src)
    r = run("-e", src, mid, tool=tool)
    assert r.returncode == 0, r.stdout + r.stderr
    rc, stream = O.encode(np.ascontiguousarray(want, dtype=np.uint32), w, h)
    assert rc == 0
    data = open(mid, "rb").read()
    assert data[:4] == b"RIFF" and data[8:16] == b"WEBPVP8L"            # webp.h:167-173
    assert struct.unpack("<I", data[4:8])[0] == len(stream) + 8
    assert data[16:] == stream
    r = run("-d", mid, dst, tool=tool)
    assert r.returncode == 0, r.stdout + r.stderr
    back = np.asarray(Image.open(dst)).astype(np.uint32)
    assert np.array_equal(((back[..., 0] << 16) | (back[..., 1] << 8) | back[..., 2]).reshape(-1), want & 0xFFFFFF)


def test_usage_and_errors(tmp_path):
    assert "Specify key -d or -e" in run("a", "b").stdout
    assert run("a", "b").returncode == 1
    assert "WebP Decoded/Encoder" in run("-h").stdout
    assert "Specify output file name" in run("-e", "a").stdout
    r = run("-d", str(tmp_path / "missing.webp"), str(tmp_path / "x.png"))
    assert r.returncode == 0 and "File operation exception" in r.stdout      # webp.cpp:183-186: message, exit 0
    bad = str(tmp_path / "bad.webp")
    open(bad, "wb").write(b"RIFF\x10\x00\x00\x00WEBPVP8 " + bytes(16))
    assert "VP8 is unsupported" in run("-d", bad, str(tmp_path / "x.png")).stdout
