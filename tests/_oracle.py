"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it was built, the
reference-component library (oracle/_ref/libref_parts.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_parts.so")
_REF_DRIVER_SO = os.path.join(ORACLE_DIR, "_ref", "libref_driver.so")

ALPHABET = (280, 256, 256, 256, 40)

STATUS = {
    0: "OK", -1: "InvalidARGBImage", -2: "TooBigARGBImage", -3: "TooBigCodeLength",
    -4: "UnsupportedVP8", -5: "InvalidVP8L", -6: "InvalidHuffman", -8: "NoMem",
}


class Token(C.Structure):
    _fields_ = [("distance", C.c_uint32), ("length", C.c_uint32), ("symbol", C.c_uint32)]


class Trace(C.Structure):
    _fields_ = [
        ("use_palette", C.c_int), ("ncolors", C.c_uint32), ("palette", C.c_uint32 * 256),
        ("bits", C.c_uint32), ("xsize", C.c_uint32), ("ysize", C.c_uint32),
        ("residual", C.POINTER(C.c_uint32)), ("tokens", C.POINTER(Token)), ("ntokens", C.c_size_t),
        ("histo", (C.c_uint32 * 280) * 5), ("lengths", (C.c_uint8 * 280) * 5),
        ("codes", (C.c_uint32 * 280) * 5),
    ]


class ExtParams(C.Structure):
    _fields_ = [
        ("use_subtract_green", C.c_int), ("predictor_bits", C.c_int),
        ("predictor_modes", C.c_void_p), ("color_bits", C.c_int), ("color_elems", C.c_void_p),
        ("use_palette", C.c_int), ("cache_bits", C.c_int), ("meta_bits", C.c_int),
        ("meta_groups", C.c_int), ("transform_order", C.c_int), ("meta_image", C.c_void_p),
        ("stray_cache", C.c_int),
    ]


def _build_oracle():
    src = os.path.join(ORACLE_DIR, "vp8l_oracle.c")
    if (not os.path.exists(_ORACLE_SO)) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build_oracle()
        L = C.CDLL(_ORACLE_SO)
        L.ora_gen_image.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ora_gen_image.restype = None
        L.ora_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_size_t), C.c_void_p]
        L.ora_encode_ext.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ora_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ora_get_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_free.restype = None
        L.ora_trace_free.argtypes = [C.c_void_p]
        L.ora_trace_free.restype = None
        L.ora_lz77.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ora_lz77_best.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
        L.ora_lz77_best.restype = None
        L.ora_prefix_encode.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
        L.ora_prefix_encode.restype = None
        L.ora_dist_to_code.argtypes = [C.c_uint64, C.c_uint64]
        L.ora_dist_to_code.restype = C.c_uint32
        L.ora_code_to_dist.argtypes = [C.c_uint32, C.c_uint32]
        L.ora_code_to_dist.restype = C.c_uint32
        L.ora_huffman_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_write_code.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ora_write_code.restype = C.c_long
        L.ora_read_code.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_int, C.c_void_p,
                                    C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
        L.ora_inverse_transform.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_uint32, C.c_uint32]
        L.ora_subtract_green.argtypes = [C.c_void_p, C.c_size_t]
        L.ora_subtract_green.restype = None
        L.ora_bundle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ora_bundle.restype = None
        _lib = L
    return _lib


_ref = None


def ref_available():
    """Whether oracle/_ref was built (decided from the file's presence alone: nothing is loaded)."""
    return os.path.exists(_REF_SO)


def ref():
    """The compiled reference components, or None when oracle/_ref was not built (GPU box)."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_SO):
            return None
        R = C.CDLL(_REF_SO)
        R.ref_lz77.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        R.ref_lz77.restype = C.c_size_t
        R.ref_prefix_encode.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
        R.ref_prefix_encode.restype = None
        R.ref_prefix_decode.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
        R.ref_prefix_decode.restype = C.c_uint32
        R.ref_dist_to_code.argtypes = [C.c_uint64, C.c_uint64]
        R.ref_dist_to_code.restype = C.c_uint32
        R.ref_code_to_dist.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_code_to_dist.restype = C.c_uint32
        R.ref_huffman_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_write_code.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        R.ref_write_code.restype = C.c_long
        R.ref_read_group.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        R.ref_inverse_transform.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_uint32, C.c_uint32, C.c_uint32]
        R.ref_bundle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        R.ref_bundle.restype = None
        R.ref_pixels_sub.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_pixels_sub.restype = C.c_uint32
        R.ref_pixels_sum.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_pixels_sum.restype = C.c_uint32
        R.ref_color_cache.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        R.ref_color_cache.restype = None
        R.ref_bit_writer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        R.ref_bit_writer.restype = C.c_size_t
        _ref = R
    return _ref


_ref_driver = None

REF_CRASHED = -99


def ref_driver_available():
    """Whether the reference's encoder/decoder drivers were compiled (oracle/_ref/libref_driver.so)."""
    return os.path.exists(_REF_DRIVER_SO)


def ref_driver():
    global _ref_driver
    if _ref_driver is None:
        if not os.path.exists(_REF_DRIVER_SO):
            return None
        R = C.CDLL(_REF_DRIVER_SO)
        R.ref_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        R.ref_encode.restype = C.c_long
        R.ref_decode.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32)]
        R.ref_decode.restype = C.c_int
        _ref_driver = R
    return _ref_driver


def ref_encode(argb, w, h):
    """The reference's VP8_LOSSLESS_ENCODER (vp8l.h:585-615): (status, stream bytes)."""
    argb = np.ascontiguousarray(argb, dtype=np.uint32)
    cap = 8 * argb.size + (1 << 16)
    buf = np.zeros(cap, dtype=np.uint8)
    n = ref_driver().ref_encode(argb.ctypes.data if argb.size else None, argb.size, w, h, buf.ctypes.data, cap)
    if n < 0:
        return int(n), b""
    return 0, buf[:n].tobytes()


def ref_decode(data, max_pixels=1 << 24):
    """The reference's VP8_LOSSLESS_DECODER (vp8l.h:330-351) in a forked child:
    (status, argb | None, w, h); status REF_CRASHED when the child died."""
    out = np.zeros(max_pixels, dtype=np.uint32)
    w = C.c_uint32()
    h = C.c_uint32()
    rc = ref_driver().ref_decode(bytes(data), len(data), out.ctypes.data, max_pixels, C.byref(w), C.byref(h))
    if rc != 0:
        return int(rc), None, 0, 0
    return 0, out[: w.value * h.value].copy(), w.value, h.value


# --------------------------------------------------------------------------- helpers
def gen_image(mode, w, h, seed=12345, ncolors=200):
    img = np.empty(w * h, dtype=np.uint32)
    lib().ora_gen_image(mode, w, h, seed, ncolors, img.ctypes.data)
    return img


def encode(argb, w, h, trace=False):
    """Returns (status, bytes) or (status, bytes, dict) with trace=True."""
    argb = np.ascontiguousarray(argb, dtype=np.uint32)
    out = C.c_void_p()
    n = C.c_size_t()
    tr = Trace() if trace else None
    rc = lib().ora_encode(argb.ctypes.data if argb.size else None, w, h, C.byref(out), C.byref(n),
                          C.byref(tr) if trace else None)
    data = b""
    if rc == 0:
        data = C.string_at(out.value, n.value)
        lib().ora_free(out)
    if not trace:
        return rc, data
    info = None
    if rc == 0:
        m = tr.xsize * tr.ysize
        toks = np.ctypeslib.as_array(C.cast(tr.tokens, C.POINTER(C.c_uint32)), shape=(tr.ntokens, 3)).copy()
        info = dict(
            use_palette=bool(tr.use_palette), ncolors=tr.ncolors,
            palette=np.array(tr.palette[: tr.ncolors], dtype=np.uint32), bits=tr.bits,
            xsize=tr.xsize, ysize=tr.ysize,
            residual=np.ctypeslib.as_array(tr.residual, shape=(m,)).copy(),
            tokens=toks,
            histo=[np.array(tr.histo[c][: ALPHABET[c]], dtype=np.uint32) for c in range(5)],
            lengths=[np.array(tr.lengths[c][: ALPHABET[c]], dtype=np.uint8) for c in range(5)],
            codes=[np.array(tr.codes[c][: ALPHABET[c]], dtype=np.uint32) for c in range(5)],
        )
        lib().ora_trace_free(C.byref(tr))
    return rc, data, info


def encode_ext(argb, w, h, **kw):
    argb = np.ascontiguousarray(argb, dtype=np.uint32)
    keep = []
    prm = ExtParams()
    for k, v in kw.items():
        if k in ("predictor_modes", "color_elems", "meta_image") and v is not None:
            arr = np.ascontiguousarray(v, dtype=np.uint32)
            keep.append(arr)
            setattr(prm, k, arr.ctypes.data)
        else:
            setattr(prm, k, v)
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().ora_encode_ext(argb.ctypes.data, w, h, C.byref(prm), C.byref(out), C.byref(n))
    data = b""
    if rc == 0:
        data = C.string_at(out.value, n.value)
        lib().ora_free(out)
    return rc, data


def decode(data):
    """Returns (status, argb ndarray | None, w, h)."""
    out = C.c_void_p()
    w = C.c_uint32()
    h = C.c_uint32()
    rc = lib().ora_decode(data, len(data), C.byref(out), C.byref(w), C.byref(h))
    if rc != 0:
        return rc, None, 0, 0
    arr = np.frombuffer(C.string_at(out.value, w.value * h.value * 4), dtype=np.uint32).copy()
    lib().ora_free(out)
    return rc, arr, w.value, h.value


def lz77(data):
    data = np.ascontiguousarray(data, dtype=np.uint32)
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().ora_lz77(data.ctypes.data, data.size, C.byref(out), C.byref(n))
    assert rc == 0
    toks = np.frombuffer(C.string_at(out.value, n.value * 12), dtype=np.uint32).reshape(-1, 3).copy()
    lib().ora_free(out)
    return toks


def lz77_best_all(data):
    """(len, dist) the greedy parser would pick at EVERY position (not just parse positions)."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    ln = np.zeros(n, dtype=np.uint32)
    ds = np.zeros(n, dtype=np.uint32)
    a = C.c_uint32()
    b = C.c_uint32()
    L = lib()
    for p in range(n):
        L.ora_lz77_best(data.ctypes.data, n, p, C.byref(a), C.byref(b))
        ln[p] = a.value
        ds[p] = b.value
    return ln, ds


def huffman_build(histo, max_len=64):
    histo = np.ascontiguousarray(histo, dtype=np.uint32)
    n = histo.size
    lengths = np.zeros(n, dtype=np.uint8)
    codes = np.zeros(n, dtype=np.uint32)
    rc = lib().ora_huffman_build(histo.ctypes.data, n, max_len, lengths.ctypes.data, codes.ctypes.data)
    return rc, lengths, codes


def write_code(lengths):
    lengths = np.ascontiguousarray(lengths, dtype=np.uint8)
    buf = np.zeros(4096, dtype=np.uint8)
    bits = lib().ora_write_code(lengths.ctypes.data, lengths.size, buf.ctypes.data, buf.size)
    return bits, buf[: (max(bits, 0) + 7) // 8].tobytes()


def inverse_transform(ttype, argb, w, h, tdata=None, txsize=0, tbits=0):
    img = np.ascontiguousarray(argb, dtype=np.uint32).copy()
    td = None if tdata is None else np.ascontiguousarray(tdata, dtype=np.uint32)
    rc = lib().ora_inverse_transform(ttype, img.ctypes.data, w, h, None if td is None else td.ctypes.data,
                                     txsize, tbits)
    return rc, img


def case_image(desc):
    """A test image from a small descriptor (shared by the recorded fixtures and their tests):
    ["gen", mode, w, h, seed] the synthetic generators; ["pal", ncolors, w, h, seed] random palette
    indices; ["rand", w, h, seed] noise."""
    kind = desc[0]
    if kind == "gen":
        _, mode, w, h, seed = desc
        return gen_image(mode, w, h, seed, 200), w, h
    if kind == "pal":
        _, ncol, w, h, seed = desc
        rng = np.random.default_rng(seed)
        pal = rng.integers(0, 1 << 32, size=ncol, dtype=np.uint32)
        return pal[rng.integers(0, ncol, size=w * h)], w, h
    if kind == "rand":
        _, w, h, seed = desc
        return np.random.default_rng(seed).integers(0, 1 << 32, size=w * h, dtype=np.uint32), w, h
    raise ValueError(desc)
