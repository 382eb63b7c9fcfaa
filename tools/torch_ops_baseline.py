"""tools/torch_ops_baseline.py -- the reference's 2D path restated as a sequence of plain torch ops, so that "what the
reference's PyTorch path costs on THIS GPU" can be measured on a box where /root/reference does not exist.
TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing in cspn_amd/ may import it (tools/bench_torch_ops_baseline.py times it,
tests/test_oracle.py pins it to the golden vectors of the unmodified reference).  It is not the oracle either: parity is
checked against oracle/cspn_oracle.c; this file exists to be TIMED.

It follows the reference's *shape of work*, not its text: per forward one normalisation (eight shifted copies of the
guidance planes stacked into [B,8,1,H+2,W+2], abs-sum, divide; cspn.py:85-144), then per iteration eight shifted copies of the
current depth (cspn.py:147-172), a product with the weights, a channel sum through a frozen 1x1x1 Conv3d with unit weights
(cspn.py:44-53,70-72), the (1 - sum w) * H0 term (cspn.py:76) and the sparse-depth pin (cspn.py:81) -- i.e. the same ~27 device
kernels and ~330 B/pixel of temporaries per iteration the reference launches.  `channel_sum="sum"` replaces the
Conv3d by torch.sum (no MIOpen involved)."""
import torch
import torch.nn.functional as F

# guidance channel k is sited at the neighbour (dy, dx): the pad tuples of cspn.py:105-132 give (dy, dx) = (1 - top, 1 - left)
# 
_PADS = [(0, 2, 0, 2), (1, 1, 0, 2), (2, 0, 0, 2), (0, 2, 1, 1), (2, 0, 1, 1), (0, 2, 2, 0), (1, 1, 2, 0), (2, 0, 2, 0)]


def _shifted_stack(planes):
    """planes: list of 8 tensors [B,1,H,W] -> [B,8,1,H+2,W+2], plane k zero-padded with its own (l, r, t, b)"""
    return torch.cat([F.pad(p, pad).unsqueeze(1) for p, pad in zip(planes, _PADS)], 1)


def _channel_sum(x, ones, mode):
    if mode == "sum":
        return x.sum(1, keepdim=True)
    return F.conv3d(x, ones)


def affinity_propagate_torch_ops(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum", channel_sum="conv3d"):
    """guidance [B,8,H,W], blur_depth [B,1,H,W], sparse_depth [B,1,H,W] or None (fp32, any device) -> [B,1,H,W]"""
    assert norm_type in ("8sum", "8sum_abs")
    with torch.no_grad():
        ones = torch.ones(1, 8, 1, 1, 1, dtype=guidance.dtype, device=guidance.device)
        g = guidance.abs() if norm_type == "8sum_abs" else guidance
        wb = _shifted_stack([g[:, k:k + 1] for k in range(8)])
        wb = wb / _channel_sum(wb.abs(), ones, channel_sum)
        wsum = _channel_sum(wb, ones, channel_sum).squeeze(1)[:, :, 1:-1, 1:-1]
        raw = blur_depth
        out = blur_depth
        mask = sparse_depth.sign() if sparse_depth is not None else None
        for _ in range(n_iter):
            nb = _shifted_stack([out] * 8)
            out = _channel_sum(wb * nb, ones, channel_sum).squeeze(1)[:, :, 1:-1, 1:-1]
            out = (1.0 - wsum) * raw + out
            if mask is not None:
                out = (1.0 - mask) * out + mask * raw
        return out
