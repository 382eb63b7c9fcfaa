"""oracle/backward.py -- CPU restatement (numpy, float64 accumulation optional) of the gradient torch autograd computes
through reference cspn_pytorch/models/cspn.py:42-83 (the path reference train.py:196-198 back-propagates through).
TEST INFRASTRUCTURE ONLY; pinned by tests/golden/cspn2d_grad_golden.npz, which the unmodified reference produced
(tests/golden/make_grad_golden.py).

Forward, folded:  H_{t+1} = c' + sum_k w'_k * shift_k(H_t),  w'_k = (1-m) w_k,
c' = (1-m)(1-sigma) H_0 + m H_0,  w_k = G_k / S,  S = sum_j |G_j|,  G_k(p) = g~_k(p + off_k) (0 outside), sigma = sum_k w_k.
Adjoint: A_N = dL/dout;  dW'_k += A_{t+1} * shift_k(H_t);  dC += A_{t+1};  A_t = sum_k shift_k^T(w'_k * A_{t+1});
then the chain through the fold and the normalisation (abs: d|x|/dx = sign(x), 0 at 0, as torch.abs)."""
import numpy as np

DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]


def _shift(a, dy, dx):
    """out(p) = a(p + (dy,dx)), zero outside; a: [B,H,W]"""
    B, H, W = a.shape
    pad = np.zeros((B, H + 2, W + 2), a.dtype)
    pad[:, 1:-1, 1:-1] = a
    return pad[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]


def cspn2d_backward_oracle(guidance, blur, sparse, grad_out, n_iter, norm_type="8sum", dtype=np.float32):
    """-> (out, dL/dguidance, dL/dblur_depth).  norm_type 'prenorm': `guidance` is the reference's gate_wb (cspn.py:85-144, cropped to the
    image), used as given at the pixel itself -- the gradient w.r.t. it is dL/dw of the derivation above, the chain through the
    normalisation and the neighbour-sited gather is the producer's (pinned by tests/golden/cspn2d_grad_prenorm_golden.npz: the unmodified
    reference's autograd with the returned gate_wb as the differentiated tensor)."""
    g = np.asarray(guidance, dtype)
    h0 = np.asarray(blur, dtype)[:, 0]
    go = np.asarray(grad_out, dtype)[:, 0]
    B, _, H, W = g.shape
    gt = np.abs(g) if norm_type == "8sum_abs" else g
    with np.errstate(all="ignore"):
        if norm_type == "prenorm":
            G, S, w = None, None, g
        else:
            G = np.stack([_shift(gt[:, k], DY[k], DX[k]) for k in range(8)], 1)      # [B,8,H,W]
            S = np.abs(G).sum(1)
            w = G / S[:, None]
        sigma = w.sum(1)
        m = np.sign(np.asarray(sparse, dtype)[:, 0]) if sparse is not None else np.zeros_like(h0)
        om = 1 - m
        wp = om[:, None] * w
        cp = om * (1 - sigma) * h0 + m * h0
        # forward, keeping every H_t
        hs = [h0]
        for _ in range(n_iter):
            ht = hs[-1]
            acc = cp.copy()
            for k in range(8):
                acc = acc + wp[:, k] * _shift(ht, DY[k], DX[k])
            hs.append(acc.astype(dtype))
        A = go.copy()
        dW = np.zeros_like(wp)
        dC = np.zeros_like(h0)
        for t in range(n_iter - 1, -1, -1):
            ht = hs[t]
            dC += A
            for k in range(8):
                dW[:, k] += A * _shift(ht, DY[k], DX[k])
            An = np.zeros_like(A)
            for k in range(8):
                An += _shift(wp[:, k] * A, -DY[k], -DX[k])
            A = An.astype(dtype)
        dw = om[:, None] * (dW - (dC * h0)[:, None])
        grad_blur = A + dC * (om * (1 - sigma) + m)
        if norm_type == "prenorm":
            return hs[-1][:, None], dw.astype(np.float32), grad_blur[:, None].astype(np.float32)
        T1 = (dw * G).sum(1)
        dG = dw / S[:, None] - np.sign(G) * (T1 / (S * S))[:, None]
        gg = np.zeros_like(g)
        for k in range(8):
            gg[:, k] = _shift(dG[:, k], -DY[k], -DX[k])    # g_k(q) feeds pixel p = q - off_k only
        if norm_type == "8sum_abs":
            gg = gg * np.sign(g)
    return hs[-1][:, None], gg.astype(np.float32), grad_blur[:, None].astype(np.float32)


# ---- 3D, Paddle contract (gates used as given, no mask): adjoint of oracle/cspn_oracle.c's cspn3d_one with norm_type 2 ----
OFF3 = [(1 - f, 1 - t, 1 - l) for f in range(3) for t in range(3) for l in range(3) if (f, t, l) != (1, 1, 1)]


def _shift3(a, dz, dy, dx):
    """out(p) = a(p + (dz,dy,dx)), zero outside; a: [B,D,H,W]"""
    B, D, H, W = a.shape
    pad = np.zeros((B, D + 2, H + 2, W + 2), a.dtype)
    pad[:, 1:-1, 1:-1, 1:-1] = a
    return pad[:, 1 + dz:1 + dz + D, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]


def cspn3d_forward_levels(gate, feat, n_iter, dtype=np.float32):
    """[H_0 .. H_n] of H_{t+1}(p) = sum_k g_k(p) H_t(p + off_k) (cspn_oracle.c cspn3d_one, norm_type 2)"""
    g = np.asarray(gate, dtype)
    hs = [np.asarray(feat, dtype)[:, 0]]
    for _ in range(n_iter):
        acc = np.zeros_like(hs[-1])
        for k, (dz, dy, dx) in enumerate(OFF3):
            acc = acc + g[:, k] * _shift3(hs[-1], dz, dy, dx)
        hs.append(acc.astype(dtype))
    return hs


def cspn3d_backward_oracle(gate, feat, grad_out, n_iter, dtype=np.float32):
    """-> (grad_gate [B,26,D,H,W], grad_feat [B,1,D,H,W]) of the n_iter-step 3D propagation with the gates as given.
    The reference op's source is absent (parity unpinned); this is the adjoint of the oracle's forward, pinned
    in tests/test_oracle.py against torch autograd through the same recurrence."""
    g = np.asarray(gate, dtype)
    hs = cspn3d_forward_levels(gate, feat, n_iter, dtype)
    A = np.asarray(grad_out, dtype)[:, 0].copy()
    dG = np.zeros_like(g)
    for t in range(n_iter - 1, -1, -1):
        An = np.zeros_like(A)
        for k, (dz, dy, dx) in enumerate(OFF3):
            dG[:, k] += A * _shift3(hs[t], dz, dy, dx)
            An += _shift3(g[:, k] * A, -dz, -dy, -dx)
        A = An.astype(dtype)
    return dG, A[:, None]
