"""ctypes wrapper over the JPEG part of oracle/_build/liboracle.so (oracle/jpeg_oracle.c).

TEST INFRASTRUCTURE ONLY.  Baseline JPEG decoding as cv::imdecode / libjpeg performs it for the reference's CompressedImage
ingest (velocity_measurment_node.py:112); pinned by tests/golden/jpeg_golden.npz (libjpeg-turbo outputs via Pillow).
"""
import ctypes as C

import numpy as np

from . import image_oracle as _io

_ready = False


def _lib():
    global _ready
    L = _io.lib()
    if not _ready:
        L.orc_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _ready = True
    return L


def info(data):
    """dict(h, w, ncomp, hmax, vmax, nblocks) of a baseline JPEG; ValueError for anything outside the supported subset."""
    data = bytes(data)
    out = (C.c_int * 6)()
    rc = _lib().orc_jpeg_info(data, len(data), out)
    if rc:
        raise ValueError(f"jpeg oracle: unsupported or corrupt stream (code {rc})")
    return dict(zip(("h", "w", "ncomp", "hmax", "vmax", "nblocks"), list(out)))


def decode(data, want_coef=False):
    """BGR uint8 [h][w][3] (gray replicated over the channels, as cv2.IMREAD_COLOR does); optionally also the quantised
    coefficient blocks [nblocks][64] int16 in decode order, natural positions, DC predicted."""
    data = bytes(data)
    i = info(data)
    bgr = np.empty((i["h"], i["w"], 3), np.uint8)
    coef = np.zeros((i["nblocks"], 64), np.int16) if want_coef else None
    rc = _lib().orc_jpeg_decode(data, len(data), coef.ctypes.data if want_coef else None, bgr.ctypes.data)
    if rc:
        raise ValueError(f"jpeg oracle: decode failed (code {rc})")
    return (bgr, coef) if want_coef else bgr
