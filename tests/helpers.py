import numpy as np
import torch


def make_inputs(B, H, W, seed=0, sparse=True, p_sparse=0.05, depth_scale=10.0, neg=False, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    g = torch.randn(B, 8, H, W, generator=gen, dtype=dtype)
    h = torch.rand(B, 1, H, W, generator=gen, dtype=dtype) * depth_scale
    s = None
    if sparse:
        m = (torch.rand(B, 1, H, W, generator=gen) < p_sparse).to(dtype)
        s = m * (torch.rand(B, 1, H, W, generator=gen, dtype=dtype) * depth_scale + 0.1)
        if neg and s.numel() > 3:
            s.view(-1)[3] = -2.5
    return g, h, s


def rel_err(a, b):
    """max|a-b| / max|b| over finite entries; NaN/Inf positions must coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    na, nb = ~np.isfinite(a), ~np.isfinite(b)
    assert np.array_equal(na, nb), "non-finite pattern differs (%d vs %d)" % (na.sum(), nb.sum())
    fin = ~nb
    if not fin.any():
        return 0.0
    denom = max(np.abs(b[fin]).max(), 1e-30)
    return float(np.abs(a[fin] - b[fin]).max() / denom)


RTOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative float tolerance"
