"""Helpers for the standard-dialect tests: libwebp (through Pillow) is the third-party checker there.  Test infrastructure only."""
import io

import numpy as np


def pillow():
    try:
        from PIL import Image, features
    except Exception:
        return None
    return Image if features.check("webp") else None


def argb_to_rgba(argb, w, h):
    a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(h, w)
    return np.stack([(a >> 16) & 0xFF, (a >> 8) & 0xFF, a & 0xFF, a >> 24], axis=-1).astype(np.uint8)


def rgba_to_argb(rgba):
    r = rgba.astype(np.uint32)
    return ((r[..., 3] << 24) | (r[..., 0] << 16) | (r[..., 1] << 8) | r[..., 2]).reshape(-1)


def libwebp_encode(argb, w, h, **opts):
    """A lossless .webp file made by libwebp; returns the whole file."""
    Image = pillow()
    im = Image.fromarray(argb_to_rgba(argb, w, h), "RGBA")
    b = io.BytesIO()
    im.save(b, "WEBP", lossless=True, exact=True, **opts)
    return b.getvalue()


def libwebp_decode(file_bytes):
    """(argb uint32[w*h], w, h) as libwebp decodes the file."""
    Image = pillow()
    im = Image.open(io.BytesIO(file_bytes))
    w, h = im.size
    return rgba_to_argb(np.asarray(im.convert("RGBA"))), w, h


def vp8l_payload(file_bytes):
    """The bytes that follow the "VP8L" fourcc of a simple-format file: chunk size, then the stream proper (webp.h:61-66)."""
    assert file_bytes[:4] == b"RIFF" and file_bytes[8:12] == b"WEBP"
    pos = 12
    while pos + 8 <= len(file_bytes):
        four, size = file_bytes[pos:pos + 4], int.from_bytes(file_bytes[pos + 4:pos + 8], "little")
        if four == b"VP8L":
            return file_bytes[pos + 4:pos + 8 + size]
        pos += 8 + size + (size & 1)
    raise ValueError("no VP8L chunk")


def riff(stream):
    pad = len(stream) & 1
    return b"RIFF" + (8 + len(stream) + pad).to_bytes(4, "little") + b"WEBPVP8L" + stream + b"\0" * pad


def test_images(rng):
    """(name, argb, w, h): content that makes libwebp use each of its tools."""
    out = []
    w, h = 97, 61
    yy, xx = np.mgrid[0:h, 0:w]
    noise = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    out.append(("noise_rgba", rgba_to_argb(noise), w, h))
    grad = np.stack([(xx * 2 + yy) & 255, (xx + yy * 3) & 255, (xx * yy // 7) & 255, np.full_like(xx, 255)], axis=-1).astype(np.uint8)
    out.append(("gradient", rgba_to_argb(grad), w, h))
    soft = grad.astype(np.int32) + rng.integers(-2, 3, size=grad.shape)
    soft[..., 3] = 255
    out.append(("gradient_noise", rgba_to_argb(np.clip(soft, 0, 255).astype(np.uint8)), w, h))
    for ncol in (2, 3, 5, 16, 17, 200):
        pal = rng.integers(0, 256, size=(ncol, 4), dtype=np.uint8)
        idx = ((xx // 5) * 7 + (yy // 3) * 3 + (rng.integers(0, 16, size=(h, w)) == 0)) % ncol
        out.append(("palette%d" % ncol, rgba_to_argb(pal[idx]), w, h))
    flat = np.zeros((h, w, 4), dtype=np.uint8)
    flat[...] = (10, 200, 30, 255)
    flat[20:40, 30:60] = (200, 10, 90, 128)
    out.append(("flat_blocks", rgba_to_argb(flat), w, h))
    out.append(("one_pixel", np.array([0x80112233], dtype=np.uint32), 1, 1))
    out.append(("one_row", rgba_to_argb(noise[:1]), w, 1))
    out.append(("one_column", rgba_to_argb(noise[:, :1]), 1, h))
    big_w, big_h = 300, 220
    y2, x2 = np.mgrid[0:big_h, 0:big_w]
    photo = np.stack([(x2 * 160 // big_w + y2 * 80 // big_h + rng.integers(-3, 4, size=x2.shape)) & 255,
                      (x2 * 150 // big_w + y2 * 90 // big_h + rng.integers(-3, 4, size=x2.shape)) & 255,
                      (x2 * 100 // big_w + y2 * 60 // big_h + rng.integers(-3, 4, size=x2.shape)) & 255,
                      np.full_like(x2, 255)], axis=-1).astype(np.uint8)
    out.append(("photo_like", rgba_to_argb(photo), big_w, big_h))
    text = np.full((big_h, big_w, 4), 255, dtype=np.uint8)
    text[(y2 // 9 + x2 // 4) % 5 == 0] = (0, 0, 0, 255)
    text[(y2 % 37 == 0)] = (255, 0, 0, 255)
    out.append(("graphics", rgba_to_argb(text), big_w, big_h))
    return out
