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


def guidance_head_oracle(x, w_guidance, w_blur=None, oheight=0, owidth=0):
    """Simple_Gudi_UpConv_Block_Last_Layer x 2 of the reference backbone (cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206; the heads
    gud_up_proj_layer6 / gud_up_proj_layer5 of :318-319, called :372-373), restated in numpy, fp64 accumulation:
        Unpool (:41-54: conv_transpose2d with a one-hot 2x2 kernel, stride 2: U[c][2i][2j] = x[c][i][j], zeros elsewhere)
        -> narrow to (oheight, owidth) if both != 0 (:196-201)  -> 3x3 conv, padding 1, no bias (:190, :203-206).
    x [B,C,h,w], w_guidance [8,C,3,3], w_blur [1,C,3,3] or None -> (guidance [B,8,H,W], blur [B,1,H,W] or None), float32."""
    x = _f32(x)
    B, C, h, w = x.shape
    U = np.zeros((B, C, 2 * h, 2 * w), np.float64)
    U[:, :, 0::2, 0::2] = x
    if oheight and owidth:
        U = U[:, :, :oheight, :owidth]
    H, W = U.shape[2], U.shape[3]
    Up = np.zeros((B, C, H + 2, W + 2), np.float64)
    Up[:, :, 1:-1, 1:-1] = U

    def conv(wt):
        wt = np.asarray(wt, np.float64)
        out = np.zeros((B, wt.shape[0], H, W), np.float64)
        for ky in range(3):
            for kx in range(3):
                out += np.einsum("oc,bcyx->boyx", wt[:, :, ky, kx], Up[:, :, ky:ky + H, kx:kx + W])
        return out.astype(np.float32)
    return conv(w_guidance), (conv(w_blur) if w_blur is not None else None)


def guidance_head_backward_oracle(x, w_guidance, w_blur, grad_guidance, grad_blur=None, oheight=0, owidth=0):
    """The gradient of guidance_head_oracle's two outputs (reference torch_resnet_cspn_nyu.py:187-206, back-propagated through at :372-373), numpy, fp64:
        out[o][Y][X] = sum_{c,ky,kx} W[o][c][ky][kx] U[c][Y + ky - 1][X + kx - 1],  U[c][2i][2j] = x[c][i][j]  (zeros elsewhere / beyond the narrowed size)
        dL/dx[c][i][j]       = sum_{o,ky,kx} W[o][c][ky][kx] g[o][2i + 1 - ky][2j + 1 - kx]
        dL/dW[o][c][ky][kx]  = sum_{b,i,j}   x[b][c][i][j]   g[b][o][2i + 1 - ky][2j + 1 - kx]              (g = dL/dout, zero outside the output)
    -> (dL/dx [B,C,h,w], dL/dw_guidance [8,C,3,3], dL/dw_blur [1,C,3,3] or None), float32.  Pinned to the unmodified reference's autograd by
    tests/golden/head_grad_golden.npz."""
    x = np.asarray(_f32(x), np.float64)
    B, C, h, w = x.shape
    H, W = (int(oheight), int(owidth)) if (oheight and owidth) else (2 * h, 2 * w)
    planes = [(np.asarray(w_guidance, np.float64), np.asarray(grad_guidance, np.float64))]
    if w_blur is not None and grad_blur is not None:
        planes.append((np.asarray(w_blur, np.float64), np.asarray(grad_blur, np.float64)))
    dx = np.zeros_like(x)
    dws = []
    for wt, g in planes:
        O = wt.shape[0]
        gp = np.zeros((B, O, 2 * h + 2, 2 * w + 2), np.float64)          # g[Y][X] at gp[Y + 1][X + 1]; zeros outside the (narrowed) output
        gp[:, :, 1:H + 1, 1:W + 1] = g
        dw = np.zeros_like(wt)
        for ky in range(3):
            for kx in range(3):
                # Y = 2i + 1 - ky -> index 2i + 2 - ky; inputs whose unpooled position lies beyond the narrowed size feed nothing
                win = gp[:, :, 2 - ky:2 - ky + 2 * h:2, 2 - kx:2 - kx + 2 * w:2]      # [B,O,h,w]
                ok = np.zeros((h, w), bool)
                ok[:(H + 1) // 2, :(W + 1) // 2] = True
                win = win * ok
                dx += np.einsum("oc,boyx->bcyx", wt[:, :, ky, kx], win)
                dw[:, :, ky, kx] = np.einsum("bcyx,boyx->oc", x, win)
        dws.append(dw.astype(np.float32))
    return dx.astype(np.float32), dws[0], (dws[1] if len(dws) > 1 else None)
