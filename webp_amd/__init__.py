"""webp_amd -- MI355X (gfx950) VP8L codec core: thin ctypes binding over the C ABI
(include/vp8l_amd.h, built as webp_amd/csrc/libvp8l_amd.so).

The codec itself is HIP code; this module only marshals pointers.  There is no CPU fallback:
importing works anywhere, but every call fails with Vp8lError(NO_DEVICE) without a gfx950 GPU,
and loading fails loudly if the library has not been built.

Names follow the reference's exception types (webp/exception/exception.h) so parity tests read
like calls into the reference: InvalidARGBImage, TooBigARGBImage, TooBigCodeLength,
UnsupportedVP8, InvalidVP8L, InvalidHuffman.
"""
import ctypes as C
import os

import numpy as np

# PyTorch bundles its own HIP runtime (same soname as /opt/rocm's).  Two HIP runtimes in one
# process cannot both own the GPU, so torch's is loaded FIRST and this library then binds to it.
import torch  # noqa: F401  (device memory, streams and torch.distributed come from it anyway)

_HERE = os.path.dirname(os.path.abspath(__file__))
# VP8L_AMD_LIB: another build of the same library (kernel experiments); there is still no fallback of any kind
LIB_PATH = os.environ.get("VP8L_AMD_LIB") or os.path.join(_HERE, "csrc", "libvp8l_amd.so")

OK = 0
STATUS_NAMES = {
    0: "OK", -1: "InvalidARGBImage", -2: "TooBigARGBImage", -3: "TooBigCodeLength", -4: "UnsupportedVP8",
    -5: "InvalidVP8L", -6: "InvalidHuffman", -7: "NoDevice", -8: "MemoryAllocationException", -9: "HipError",
    -10: "BadArgument", -11: "Workspace",
}
STAGE_RESIDUAL, STAGE_MATCH, STAGE_HISTO, STAGE_CODELEN, STAGE_INFO, STAGE_PRED_MODES, STAGE_CROSS_ELEMS, STAGE_META_GROUPS = range(8)
DEC_OPT_THREADS_PER_STREAM = 1
DEC_OPT_WHOLE_DEVICE = 2
DEC_OPT_SCOUT_FORM = 3
ENCODE_STAGES = ("palette", "transform", "match", "parse", "histogram", "header", "bitcount_scan", "emit",
                 "match_fallback")  # the last one is the part of "match" spent in the exhaustive kernel


class Vp8lError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        self.name = STATUS_NAMES.get(status, "status %d" % status)
        detail = ""
        if status in (-7, -9):
            try:
                detail = ": " + lib().vp8l_last_error().decode()
            except Exception:  # pragma: no cover
                pass
        super().__init__("%s%s%s" % (self.name, " in " + where if where else "", detail))


_lib = None


def lib():
    """The loaded C ABI.  Raises if the HIP library was not built (never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        pp_void = C.POINTER(C.c_void_p)
        L.vp8l_init.restype = C.c_int
        L.vp8l_status_string.argtypes = [C.c_int]
        L.vp8l_status_string.restype = C.c_char_p
        L.vp8l_last_error.restype = C.c_char_p
        L.vp8l_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, pp_void, C.POINTER(C.c_size_t), C.c_uint32]
        L.vp8l_decode.argtypes = [C.c_char_p, C.c_size_t, pp_void, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vp8l_get_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vp8l_free.argtypes = [C.c_void_p]
        L.vp8l_free.restype = None
        L.vp8l_encode_batch.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vp8l_decode_batch.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vp8l_encode_batch_devices.argtypes = [C.c_void_p, C.c_size_t] + L.vp8l_encode_batch.argtypes[1:]
        L.vp8l_decode_batch_devices.argtypes = [C.c_void_p, C.c_size_t] + L.vp8l_decode_batch.argtypes[1:]
        L.vp8l_encoder_create.argtypes = [C.c_int, pp_void]
        L.vp8l_encoder_destroy.argtypes = [C.c_void_p]
        L.vp8l_encoder_destroy.restype = None
        L.vp8l_encode_bound.argtypes = [C.c_uint32, C.c_uint32]
        L.vp8l_encode_bound.restype = C.c_size_t
        L.vp8l_encoder_encode_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vp8l_encoder_fetch_stage.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.vp8l_encoder_fetch_stage.restype = C.c_long
        L.vp8l_encoder_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vp8l_encoder_last_dense_tiles.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.vp8l_encoder_last_dense_tiles.restype = C.c_long
        L.vp8l_encoder_last_tile_kinds.argtypes = [C.c_void_p, C.c_void_p]
        L.vp8l_encoder_last_tile_kinds.restype = C.c_long
        L.vp8l_encoder_last_device_seconds.argtypes = [C.c_void_p]
        L.vp8l_encoder_last_device_seconds.restype = C.c_double
        L.vp8l_encoder_last_stage_seconds.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vp8l_decoder_create.argtypes = [C.c_int, pp_void]
        L.vp8l_decoder_destroy.argtypes = [C.c_void_p]
        L.vp8l_decoder_destroy.restype = None
        L.vp8l_decoder_decode_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
        L.vp8l_decoder_last_device_seconds.argtypes = [C.c_void_p]
        L.vp8l_decoder_last_device_seconds.restype = C.c_double
        L.vp8l_decoder_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vp8l_decoder_last_whole_device_streams.argtypes = [C.c_void_p]
        L.vp8l_decoder_last_whole_device_streams.restype = C.c_long
        L.vp8l_decoder_last_scout_streams.argtypes = [C.c_void_p]
        L.vp8l_decoder_last_scout_streams.restype = C.c_long
        L.vp8l_generate_device.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.vp8l_encode_device_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = (
    "vp8l_init", "vp8l_status_string", "vp8l_last_error", "vp8l_encode", "vp8l_decode", "vp8l_get_info", "vp8l_free",
    "vp8l_encode_batch", "vp8l_decode_batch", "vp8l_encoder_create", "vp8l_encoder_destroy", "vp8l_encode_bound",
    "vp8l_encoder_encode_device", "vp8l_encoder_fetch_stage", "vp8l_decoder_create", "vp8l_decoder_destroy",
    "vp8l_decoder_decode_device", "vp8l_encoder_last_device_seconds", "vp8l_encoder_last_stage_seconds",
    "vp8l_decoder_last_device_seconds", "vp8l_generate_device", "vp8l_encoder_set_option",
    "vp8l_encoder_last_dense_tiles", "vp8l_encoder_last_tile_kinds", "vp8l_decoder_set_option",
    "vp8l_decoder_last_whole_device_streams",
    "vp8l_decoder_last_scout_streams", "vp8l_encode_device_noalloc", "vp8l_encode_batch_devices", "vp8l_decode_batch_devices",
)


def init():
    rc = lib().vp8l_init()
    if rc < 0:
        raise Vp8lError(rc, "vp8l_init")
    return rc


# ------------------------------------------------------------------ host-memory forms
def riff_container(stream):
    """The bytes of a .webp file around a VP8L stream (webp.h:167-173: "RIFF", size, "WEBP", "VP8L", stream), with the pad byte RIFF
    asks for after a chunk of odd length.  libwebp reads the result if the stream is in the standard dialect."""
    pad = len(stream) & 1
    return b"RIFF" + (8 + len(stream) + pad).to_bytes(4, "little") + b"WEBPVP8L" + stream + b"\0" * pad


def encode(argb, width, height, predictor_bits=0, cross_color_bits=0, standard=False, cache_bits=0, meta_bits=0):
    """VP8_LOSSLESS_ENCODER(argb, width, height).get_bit_writer() -> stream bytes (vp8l.h:585-615).
    predictor_bits 2..9 adds the predictor transform (not the reference encoder's bytes any more)."""
    a = None if argb is None else np.ascontiguousarray(argb, dtype=np.uint32)
    # the C ABI cannot know the buffer's size (the C++ mirror checks the same, webp.hpp): a short array would be read past its end
    if a is not None and a.size and max(int(width), int(height)) <= 16384 and a.size < int(width) * int(height):
        raise Vp8lError(-1, "vp8l_encode (array holds %d pixels, %d x %d needs %d)" % (a.size, width, height, int(width) * int(height)))
    # each option has its own field of the flags word: a value beyond its range would spill into the next field
    for name, v, hi in (("predictor_bits", predictor_bits, 9), ("cross_color_bits", cross_color_bits, 9), ("cache_bits", cache_bits, 11),
                        ("meta_bits", meta_bits, 9)):
        if not 0 <= int(v) <= hi:
            raise Vp8lError(-10, "vp8l_encode (%s = %d, allowed 0..%d)" % (name, int(v), hi))
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().vp8l_encode(a.ctypes.data if a is not None and a.size else None, width, height, C.byref(out), C.byref(n), int(predictor_bits) | (int(cross_color_bits) << 4) | (0x100 if standard else 0) | (int(cache_bits) << 12) | (int(meta_bits) << 16))
    if rc != OK:
        raise Vp8lError(rc, "vp8l_encode")
    data = C.string_at(out.value, n.value)
    lib().vp8l_free(out)
    return data


def decode(data):
    """VP8_LOSSLESS_DECODER(data, len, argb) -> (argb uint32[w*h], width, height) (vp8l.h:330-351)."""
    out = C.c_void_p()
    w, h = C.c_uint32(), C.c_uint32()
    rc = lib().vp8l_decode(data, len(data), C.byref(out), C.byref(w), C.byref(h))
    if rc != OK:
        raise Vp8lError(rc, "vp8l_decode")
    arr = np.frombuffer(C.string_at(out.value, w.value * h.value * 4), dtype=np.uint32).copy()
    lib().vp8l_free(out)
    return arr, w.value, h.value


def get_info(data):
    w, h = C.c_uint32(), C.c_uint32()
    rc = lib().vp8l_get_info(data, len(data), C.byref(w), C.byref(h))
    if rc != OK:
        raise Vp8lError(rc, "vp8l_get_info")
    return w.value, h.value


def encode_batch(images, device=-1, devices=None):
    """images: list of (argb, w, h).  Returns list of (status, bytes).  devices: a list of device numbers to deal the
    images out to (vp8l_encode_batch_devices) instead of the one `device`."""
    n = len(images)
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a, _, _ in images]
    for a, (_, w, h) in zip(arrs, images):
        if a.size and max(int(w), int(h)) <= 16384 and a.size < int(w) * int(h):  # larger sides get TooBigARGBImage from the library
            raise Vp8lError(-1, "vp8l_encode_batch (an array holds %d pixels, %d x %d needs %d)" % (a.size, w, h, int(w) * int(h)))
    ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
    ws = (C.c_uint32 * n)(*[w for _, w, _ in images])
    hs = (C.c_uint32 * n)(*[h for _, _, h in images])
    outs = (C.c_void_p * n)()
    lens = (C.c_size_t * n)()
    status = (C.c_int * n)()
    if devices is not None:
        devs = (C.c_int * len(devices))(*devices)
        rc = lib().vp8l_encode_batch_devices(devs, len(devices), n, ptrs, ws, hs, outs, lens, status)
    else:
        rc = lib().vp8l_encode_batch(device, n, ptrs, ws, hs, outs, lens, status)
    if rc != OK:
        raise Vp8lError(rc, "vp8l_encode_batch")
    res = []
    for i in range(n):
        if status[i] == OK:
            res.append((OK, C.string_at(outs[i], lens[i])))
            lib().vp8l_free(outs[i])
        else:
            res.append((status[i], b""))
    return res


def decode_batch(streams, device=-1, devices=None):
    """streams: list of bytes.  Returns list of (status, argb | None, w, h).  devices: as in encode_batch."""
    n = len(streams)
    bufs = [C.create_string_buffer(s, len(s)) for s in streams]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(s) for s in streams])
    outs = (C.c_void_p * n)()
    ws = (C.c_uint32 * n)()
    hs = (C.c_uint32 * n)()
    status = (C.c_int * n)()
    if devices is not None:
        devs = (C.c_int * len(devices))(*devices)
        rc = lib().vp8l_decode_batch_devices(devs, len(devices), n, ptrs, lens, outs, ws, hs, status)
    else:
        rc = lib().vp8l_decode_batch(device, n, ptrs, lens, outs, ws, hs, status)
    if rc != OK:
        raise Vp8lError(rc, "vp8l_decode_batch")
    res = []
    for i in range(n):
        if status[i] == OK:
            arr = np.frombuffer(C.string_at(outs[i], ws[i] * hs[i] * 4), dtype=np.uint32).copy()
            lib().vp8l_free(outs[i])
            res.append((OK, arr, ws[i], hs[i]))
        else:
            res.append((status[i], None, ws[i], hs[i]))
    return res


# ------------------------------------------------------------------ device-memory forms
def c_array(ctype, values):
    """A ctypes array of `ctype` holding `values`; an array of that type is passed through as it is."""
    if isinstance(values, C.Array) and values._type_ is ctype:
        return values
    return (ctype * len(values))(*values)


class Encoder:
    """Batch encoder context (vp8l_encoder_*).  Device pointers are plain integers, e.g.
    torch.Tensor.data_ptr() of uint8/int32 CUDA tensors."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        rc = lib().vp8l_encoder_create(device, C.byref(self._h))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_create")

    def close(self):
        if self._h:
            lib().vp8l_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    @staticmethod
    def bound(w, h):
        return lib().vp8l_encode_bound(w, h)

    def encode_device(self, d_argb_ptrs, widths, heights, d_out_ptr, out_cap):
        """Returns (offsets, lengths, status) as numpy arrays.  The three per-image arguments may be ctypes arrays made with
        c_array() (a caller that repeats a call then does not pay for converting Python lists every time)."""
        n = len(d_argb_ptrs)
        ptrs = c_array(C.c_void_p, d_argb_ptrs)
        ws = c_array(C.c_uint32, widths)
        hs = c_array(C.c_uint32, heights)
        off = (C.c_size_t * n)()
        lens = (C.c_size_t * n)()
        status = (C.c_int * n)()
        rc = lib().vp8l_encoder_encode_device(self._h, n, ptrs, ws, hs, d_out_ptr, out_cap, off, lens, status)
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_encode_device")
        return np.array(off[:], dtype=np.uint64), np.array(lens[:], dtype=np.uint64), np.array(status[:], dtype=np.int32)

    def fetch_stage(self, image, what):
        info = np.zeros(4, dtype=np.uint32)
        rc = lib().vp8l_encoder_fetch_stage(self._h, image, STAGE_INFO, info.ctypes.data, info.nbytes)
        if rc < 0:
            raise Vp8lError(int(rc), "vp8l_encoder_fetch_stage")
        if what == STAGE_INFO:
            return dict(ncolors=int(info[0]), bits=int(info[1]), xsize=int(info[2]), M=int(info[3]))
        if what in (STAGE_PRED_MODES, STAGE_CROSS_ELEMS, STAGE_META_GROUPS):
            buf = np.zeros(3 + 1024, dtype=np.uint32)
            rc = lib().vp8l_encoder_fetch_stage(self._h, image, what, buf.ctypes.data, buf.nbytes)
            if rc < 0:
                raise Vp8lError(int(rc), "vp8l_encoder_fetch_stage")
            if rc == 0:
                return None
            tiles = buf[3:3 + int(buf[1]) * int(buf[2])].copy()
            return dict(bits=int(buf[0]), tx=int(buf[1]), ty=int(buf[2]), modes=tiles, elems=tiles, groups=tiles)
        if what in (STAGE_RESIDUAL, STAGE_MATCH):
            buf = np.zeros(int(info[3]), dtype=np.uint32)
        elif what == STAGE_HISTO:
            buf = np.zeros(5 * 280, dtype=np.uint32)
        else:
            buf = np.zeros(5 * 280, dtype=np.uint8)
        rc = lib().vp8l_encoder_fetch_stage(self._h, image, what, buf.ctypes.data, buf.nbytes)
        if rc < 0:
            raise Vp8lError(int(rc), "vp8l_encoder_fetch_stage")
        return buf

    def last_device_seconds(self):
        return lib().vp8l_encoder_last_device_seconds(self._h)

    def set_exhaustive_search(self, on):
        """True: test all 1024 distances everywhere; False (default): candidate lists + fallback."""
        rc = lib().vp8l_encoder_set_option(self._h, 1, 1 if on else 0)
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_match_search(self, mode):
        """0 (default): candidate lists first; 1: exhaustive everywhere; 2: the candidate lists refuse every tile; 3: as 2, with the
        search by numbered grams in front of the exhaustive one; 4: as 0, with the search along the parse for palette images' dense
        tiles (same streams every way)."""
        rc = lib().vp8l_encoder_set_option(self._h, 1, int(mode))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_predictor_bits(self, bits):
        """0 (default): the reference encoder's output.  2..9: add the predictor transform, tiles of 2**bits."""
        rc = lib().vp8l_encoder_set_option(self._h, 2, int(bits))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_cross_color_bits(self, bits):
        """0 (default): none.  2..9: add the cross-colour transform, tiles of 2**bits (after subtract green / predictor)."""
        rc = lib().vp8l_encoder_set_option(self._h, 3, int(bits))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_standard(self, on):
        """False (default): the reference's dialect.  True: the standard VP8L format (what libwebp reads)."""
        rc = lib().vp8l_encoder_set_option(self._h, 4, 1 if on else 0)
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_cache_bits(self, bits):
        """0 (default): no colour cache.  1..11: a colour cache of 2**bits entries."""
        rc = lib().vp8l_encoder_set_option(self._h, 5, int(bits))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def set_meta(self, bits, groups=4):
        """bits 0 (default): one Huffman group.  2..9: an entropy image with tiles of 2**bits pixels and `groups` (2..8) groups."""
        rc = lib().vp8l_encoder_set_option(self._h, 6, int(bits))
        if rc == OK and bits:
            rc = lib().vp8l_encoder_set_option(self._h, 7, int(groups))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_encoder_set_option")

    def last_dense_tiles(self):
        total = C.c_long()
        dense = lib().vp8l_encoder_last_dense_tiles(self._h, C.byref(total))
        if dense < 0:
            raise Vp8lError(int(dense), "vp8l_encoder_last_dense_tiles")
        return int(dense), int(total.value)

    def last_tile_kinds(self):
        """{kind: tiles} of the last call: how each 2048-position tile was searched (same matches whichever way)."""
        buf = (C.c_long * 8)()
        total = lib().vp8l_encoder_last_tile_kinds(self._h, buf)
        if total < 0:
            raise Vp8lError(int(total), "vp8l_encoder_last_tile_kinds")
        names = {0: "lists", 1: "exhaustive", 2: "pairs", 4: "frequent_value", 5: "grams", 6: "parse_walk"}
        return {names.get(k, "flag%d" % k): int(buf[k]) for k in range(8) if buf[k]}

    def last_stage_seconds(self):
        buf = (C.c_double * 9)()
        k = lib().vp8l_encoder_last_stage_seconds(self._h, buf, 9)
        return dict(zip(ENCODE_STAGES, buf[:k]))


class Decoder:
    """Batch decoder context (vp8l_decoder_*)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        rc = lib().vp8l_decoder_create(device, C.byref(self._h))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_decoder_create")

    def close(self):
        if self._h:
            lib().vp8l_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def decode_device(self, d_data_ptrs, lengths, widths, heights, d_argb_ptrs):
        n = len(d_data_ptrs)
        dp = c_array(C.c_void_p, d_data_ptrs)
        ln = c_array(C.c_size_t, lengths)
        ws = c_array(C.c_uint32, widths)
        hs = c_array(C.c_uint32, heights)
        op = c_array(C.c_void_p, d_argb_ptrs)
        status = (C.c_int * n)()
        rc = lib().vp8l_decoder_decode_device(self._h, n, dp, ln, ws, hs, op, status)
        if rc != OK:
            raise Vp8lError(rc, "vp8l_decoder_decode_device")
        return np.array(status[:], dtype=np.int32)

    def last_device_seconds(self):
        return lib().vp8l_decoder_last_device_seconds(self._h)

    def set_whole_device(self, mode):
        """0 = chosen per call, 1 = never, 2 = always: large streams decoded by many workgroups (same pixels)."""
        rc = lib().vp8l_decoder_set_option(self._h, DEC_OPT_WHOLE_DEVICE, int(mode))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_decoder_set_option")

    def set_scout_form(self, form):
        """How the segment kernels find a segment's tokens: 0 = a lane per piece of 8192 bits (default), 1 = a token at
        every bit offset (same pixels)."""
        rc = lib().vp8l_decoder_set_option(self._h, DEC_OPT_SCOUT_FORM, int(form))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_decoder_set_option")

    def last_whole_device_streams(self):
        return int(lib().vp8l_decoder_last_whole_device_streams(self._h))

    def last_scout_streams(self):
        return int(lib().vp8l_decoder_last_scout_streams(self._h))

    def set_threads_per_stream(self, threads):
        """0 or 1024 = a workgroup per stream (default), 64 = a wave per stream (same pixels either way)."""
        rc = lib().vp8l_decoder_set_option(self._h, DEC_OPT_THREADS_PER_STREAM, int(threads))
        if rc != OK:
            raise Vp8lError(rc, "vp8l_decoder_set_option")


def generate_device(mode, w, h, seed, ncolors, d_ptr, device=-1):
    rc = lib().vp8l_generate_device(device, mode, w, h, seed, ncolors, d_ptr)
    if rc != OK:
        raise Vp8lError(rc, "vp8l_generate_device")


NOALLOC_MAX_PIXELS = 1 << 21


def encode_device_noalloc(d_argb_ptr, width, height, d_out_ptr, out_cap, device=-1, stream=None, d_result_ptr=None, wait=True):
    """vp8l_encode_device_noalloc: one image, workspace in a static device array, no allocation.  Returns the stream
    length (wait=True) or None (wait=False: read the two uint64 at d_result_ptr after synchronising yourself)."""
    n = C.c_size_t()
    rc = lib().vp8l_encode_device_noalloc(device, stream, d_argb_ptr, width, height, d_out_ptr, out_cap, d_result_ptr,
                                          C.byref(n) if wait else None)
    if rc != OK:
        raise Vp8lError(rc, "vp8l_encode_device_noalloc")
    return n.value if wait else None
