"""ctypes wrapper over oracle/_build/liboracle.so (built from oracle/image_oracle.c by `make -C oracle`).

TEST INFRASTRUCTURE ONLY — see oracle/image_oracle.c for the semantics and the reference call sites.
PARITY UNPINNED (OpenCV absent): this restatement defines the image-stage semantics.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")


def build():
    srcs = [os.path.join(_HERE, f) for f in ("image_oracle.c", "jpeg_oracle.c")]
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_gray_bgr8.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_pyr_down.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_lk_levels.argtypes = [C.c_int] * 4
        L.orc_scharr.argtypes = [u8p, C.c_int, C.c_int, i16p]
        L.orc_mineig.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_select_corners.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                         f32p, C.c_int, C.POINTER(C.c_int)]
        L.orc_good_features.argtypes = [u8p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_int, f32p, C.c_int]
        L.orc_lk_pyr.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_double, f32p, u8p, f32p]
        _lib = L
    return _lib


def gray_bgr8(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gray_bgr8(bgr, h, w, out)
    return out


def pyr_down(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(src, h, w, out)
    return out


def lk_levels(h, w, win, max_level):
    return lib().orc_lk_levels(h, w, win, max_level)


def pyramid(gray, win=15, max_level=3):
    levels = [np.ascontiguousarray(gray, np.uint8)]
    for _ in range(lk_levels(gray.shape[0], gray.shape[1], win, max_level)):
        levels.append(pyr_down(levels[-1]))
    return levels


def scharr(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    out = np.empty((h, w, 2), np.int16)
    lib().orc_scharr(src, h, w, out)
    return out


def mineig(gray, block):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    rc = lib().orc_mineig(gray, h, w, block, out)
    if rc:
        raise ValueError(f"orc_mineig rc={rc}")
    return out


def _maskptr(mask):
    if mask is None:
        return None, None
    m = np.ascontiguousarray(mask, np.uint8)
    return m, m.ctypes.data_as(C.c_void_p)


def select_corners(eig, max_corners, quality, min_distance, mask=None, cap=None):
    eig = np.ascontiguousarray(eig, np.float32)
    h, w = eig.shape
    cap = cap or (max_corners if max_corners > 0 else h * w)
    pts = np.empty((cap, 2), np.float32)
    ncand = C.c_int(0)
    keep, mp = _maskptr(mask)
    n = lib().orc_select_corners(eig, mp, h, w, max_corners, quality, min_distance, pts, cap, C.byref(ncand))
    if n < 0:
        raise ValueError(f"orc_select_corners rc={n}")
    return pts[:n].reshape(n, 1, 2).copy(), ncand.value


def good_features(gray, max_corners, quality, min_distance, block, mask=None):
    """goodFeaturesToTrack semantics -> (N,1,2) float32 (x,y)."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = max_corners if max_corners > 0 else h * w
    pts = np.empty((cap, 2), np.float32)
    keep, mp = _maskptr(mask)
    n = lib().orc_good_features(gray, mp, h, w, max_corners, quality, min_distance, block, pts, cap)
    if n < 0:
        raise ValueError(f"orc_good_features rc={n}")
    return pts[:n].reshape(n, 1, 2).copy()


def lk_pyr(prev, nxt, prev_pts, win=15, max_level=3, max_count=20, eps=0.03, min_eig_thr=1e-4):
    """calcOpticalFlowPyrLK semantics -> (next_pts (N,1,2) f32, status (N,1) u8, err (N,1) f32)."""
    prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
    h, w = prev.shape
    p = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = len(p)
    out = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8); err = np.zeros(n, np.float32)
    rc = lib().orc_lk_pyr(prev, nxt, h, w, p, n, win, max_level, max_count, eps, min_eig_thr, out, st, err)
    if rc:
        raise ValueError(f"orc_lk_pyr rc={rc}")
    return out.reshape(n, 1, 2), st.reshape(n, 1), err.reshape(n, 1)
