"""CPU-only: the shipped library loads and exports every function include/vp8l_amd.h declares.
No codec call is made here (there is no GPU in the build container and no CPU fallback)."""
import ctypes as C
import os
import re

import webp_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vp8l_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vp8l_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(webp_amd.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(webp_amd.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    L = C.CDLL(webp_amd.LIB_PATH)
    for name in declared_functions():
        assert hasattr(L, name), name


def test_get_info_is_host_only():
    # ReadInfo on the host: 9 header bytes
    hdr = bytes([0, 0, 0, 0, 0x2F]) + (((100 - 1) | ((50 - 1) << 14)).to_bytes(4, "little"))
    assert webp_amd.get_info(hdr) == (100, 50)
    try:
        webp_amd.get_info(b"\0\0\0\0\x2e\0\0\0\0")
        assert False
    except webp_amd.Vp8lError as e:
        assert e.name == "UnsupportedVP8"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        return
    try:
        webp_amd.encode([0xFF000000], 1, 1)
        assert False, "encode must not succeed without a GPU"
    except webp_amd.Vp8lError as e:
        assert e.name == "NoDevice"


def test_binding_refuses_short_pixel_arrays():
    # the C ABI only gets a pointer: the binding must not let a short array through (it would be read past its end)
    import numpy as np
    import pytest
    short = np.zeros(10, dtype=np.uint32)
    with pytest.raises(webp_amd.Vp8lError) as ei:
        webp_amd.encode(short, 4, 4)
    assert ei.value.name == "InvalidARGBImage"
    with pytest.raises(webp_amd.Vp8lError) as ei:
        webp_amd.encode_batch([(np.zeros(16, dtype=np.uint32), 4, 4), (short, 4, 4)])
    assert ei.value.name == "InvalidARGBImage"


def test_c_array_passes_matching_arrays_through():
    # per-image arguments of the device entry points: lists are converted, arrays made once are used as they are
    a = webp_amd.c_array(C.c_uint32, [1, 2, 3])
    assert list(a) == [1, 2, 3] and a._type_ is C.c_uint32
    assert webp_amd.c_array(C.c_uint32, a) is a
    b = webp_amd.c_array(C.c_size_t, a)  # another element type: converted
    assert b is not a and list(b) == [1, 2, 3] and b._type_ is C.c_size_t
    assert len(webp_amd.c_array(C.c_void_p, [])) == 0
