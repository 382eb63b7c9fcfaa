"""tools/torch_path.py -- the propagation written with stock PyTorch ops (pad, slice, sum, div), the way a framework user would
write reference cspn_pytorch/models/cspn.py:42-172 today: the "before" number on the same GPU, and an autograd-capable
cross-check for the HIP forward / backward at sizes the numpy oracle is too slow for.  Not part of the product."""
import torch
import torch.nn.functional as TF

DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]


def _gather8(x):
    """x [B,8,H,W] or [B,1,H,W] -> [B,8,H,W]: plane k read at (y + DY[k], x + DX[k]), zero outside (cspn.py:91-132,149-167)"""
    B, C, H, W = x.shape
    p = TF.pad(x, (1, 1, 1, 1))
    return torch.stack([p[:, k if C == 8 else 0, 1 + DY[k]:1 + DY[k] + H, 1 + DX[k]:1 + DX[k] + W] for k in range(8)], 1)


def cspn2d_torch(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum"):
    g = guidance.abs() if norm_type == "8sum_abs" else guidance            # cspn.py:88-89
    G = _gather8(g)
    w = G / G.abs().sum(1, keepdim=True)                                   # cspn.py:135-138
    gate_sum = w.sum(1, keepdim=True)                                      # cspn.py:139-142
    m = sparse_depth.sign() if sparse_depth is not None else None          # cspn.py:64
    h = blur_depth
    for _ in range(n_iter):                                                # cspn.py:66-81
        h_new = (w * _gather8(h)).sum(1, keepdim=True)
        h_new = (1.0 - gate_sum) * blur_depth + h_new
        if m is not None:
            h_new = (1.0 - m) * h_new + m * blur_depth
        h = h_new
    return h


if __name__ == "__main__":
    import json
    import sys
    import time
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H, W, N = 304, 1216, 24
    gen = torch.Generator(device="cuda").manual_seed(0)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    with torch.no_grad():
        for _ in range(2):
            cspn2d_torch(g, h, None, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cspn2d_torch(g, h, None, N)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
    res = {"path": "stock PyTorch ops on the same GPU (tools/torch_path.py)", "B": B, "H": H, "W": W, "n_iter": N,
           "ms_per_forward": round(ms, 3), "Mpix_iters_per_s": round(B * H * W * N / ms / 1e3, 1)}
    if B <= 16:  # forward + backward through torch autograd (it keeps ~27 temporaries per iteration)
        go = torch.randn_like(h)
        gr, hr = g.clone().requires_grad_(True), h.clone().requires_grad_(True)
        for _ in range(2):
            gr.grad = hr.grad = None
            cspn2d_torch(gr, hr, None, N).backward(go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gr.grad = hr.grad = None
            cspn2d_torch(gr, hr, None, N).backward(go)
        torch.cuda.synchronize()
        res["ms_per_forward_backward"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        res["peak_memory_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import cspn_amd
        m = cspn_amd.Affinity_Propagate(N, 3, "8sum")
        for _ in range(3):
            gr.grad = hr.grad = None
            m(gr, hr, None).backward(go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gr.grad = hr.grad = None
            m(gr, hr, None).backward(go)
        torch.cuda.synchronize()
        res["cspn_amd_ms_per_forward_backward"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    print(json.dumps(res))
