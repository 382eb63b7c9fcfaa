"""ctypes front-end of oracle/cspn_oracle.c (CPU restatement of
/root/reference/cspn_pytorch/models/cspn.py:42-172).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcspn_oracle.so")
_lib = None

NORM_TYPES = {"8sum": 0, "8sum_abs": 1, "none": 2, "prenorm": 3}


def build(force=False):
    """Compile libcspn_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "cspn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libcspn_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        lib.cspn2d_oracle_f32.restype = ctypes.c_int
        lib.cspn2d_oracle_f32.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 5
        lib.cspn3d_oracle_f32.restype = ctypes.c_int
        lib.cspn3d_oracle_f32.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 6
        lib.cspn2d_oracle_gate_wb_f32.restype = ctypes.c_int
        lib.cspn2d_oracle_gate_wb_f32.argtypes = [fp, fp] + [ctypes.c_int] * 4
        lib.cspn_oracle_threads.restype = ctypes.c_int
        lib.cspn_oracle_set_threads.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def oracle_threads():
    return int(_load().cspn_oracle_threads())


def set_oracle_threads(n):
    _load().cspn_oracle_set_threads(int(n))


def _f32(a):
    if a is None:
        return None
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    fp = ctypes.POINTER(ctypes.c_float)
    return a.ctypes.data_as(fp) if a is not None else ctypes.cast(None, fp)


def cspn2d_oracle(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum"):
    """guidance [B,8,H,W], blur_depth [B,1,H,W], sparse_depth [B,1,H,W]|None -> np.float32 [B,1,H,W]"""
    g, h, s = _f32(guidance), _f32(blur_depth), _f32(sparse_depth)
    B, C, H, W = g.shape
    assert C == 8 and h.shape == (B, 1, H, W) and (s is None or s.shape == h.shape)
    out = np.empty_like(h)
    rc = _load().cspn2d_oracle_f32(_ptr(g), _ptr(h), _ptr(s), _ptr(out), B, H, W, int(n_iter), NORM_TYPES[norm_type])
    if rc:
        raise RuntimeError("cspn2d_oracle_f32 failed: %d" % rc)
    return out


def cspn2d_gate_wb_oracle(guidance, norm_type="8sum"):
    """guidance [B,8,H,W] -> gate_wb [B,8,H,W] of reference cspn.py:85-144 (affinity_normalization), cropped to the image: the
    normalised, consumer-sited weights; cspn2d_oracle(gate_wb, ..., norm_type="prenorm") continues from there."""
    g = _f32(guidance)
    B, C, H, W = g.shape
    assert C == 8
    wb = np.empty_like(g)
    rc = _load().cspn2d_oracle_gate_wb_f32(_ptr(g), _ptr(wb), B, H, W, NORM_TYPES[norm_type])
    if rc:
        raise RuntimeError("cspn2d_oracle_gate_wb_f32 failed: %d" % rc)
    return wb


def cspn3d_oracle(gate, feat, sparse=None, n_iter=12, norm_type="8sum_abs"):
    """gate [B,26,D,H,W], feat [B,1,D,H,W] -> np.float32 [B,1,D,H,W]  (parity unpinned, see .c header)"""
    g, h, s = _f32(gate), _f32(feat), _f32(sparse)
    B, C, D, H, W = g.shape
    assert C == 26 and h.shape == (B, 1, D, H, W) and (s is None or s.shape == h.shape)
    out = np.empty_like(h)
    rc = _load().cspn3d_oracle_f32(_ptr(g), _ptr(h), _ptr(s), _ptr(out), B, D, H, W, int(n_iter), NORM_TYPES[norm_type])
    if rc:
        raise RuntimeError("cspn3d_oracle_f32 failed: %d" % rc)
    return out
