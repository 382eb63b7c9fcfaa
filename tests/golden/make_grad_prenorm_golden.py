"""Generates tests/golden/cspn2d_grad_prenorm_golden.npz: the gradient of the UNMODIFIED reference module
(/root/reference/cspn_pytorch/models/cspn.py:42-83) with respect to the tensor its affinity_normalization returns (gate_wb, :85-144) and to
blur_depth, by torch autograd on seeded CPU inputs (oracle/ref_harness.py reference_grads_wrt_gate_wb) -- the backward of the
pre-normalised input contract (CSPN_NORM_PRENORM).  Authoring container only:
    python tests/golden/make_grad_prenorm_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.ref_harness import reference_grads_wrt_gate_wb  # noqa: E402

# name, B, H, W, n_iter, norm_type of the normalisation that produced gate_wb, sparse kind
CASES = [
    ("a_8sum_sparse_neg", 2, 11, 14, 6, "8sum", "neg"),
    ("b_abs_sparse", 1, 13, 17, 24, "8sum_abs", "pos"),
    ("c_nosparse", 1, 9, 40, 12, "8sum", None),
    ("d_row_1x7", 1, 1, 7, 3, "8sum", None),
    ("e_n1", 2, 6, 5, 1, "8sum", "pos"),
    ("f_wide_band", 1, 8, 272, 24, "8sum", "pos"),
    ("g_wide_n8", 1, 12, 260, 8, "8sum_abs", None),
]


def main():
    out = {}
    for idx, (name, B, H, W, N, norm, sk) in enumerate(CASES):
        gen = torch.Generator().manual_seed(900 + idx)
        g = torch.randn(B, 8, H, W, generator=gen)
        h = torch.rand(B, 1, H, W, generator=gen) * 10
        s = None
        if sk is not None:
            m = (torch.rand(B, 1, H, W, generator=gen) < 0.08).float()
            s = m * (torch.rand(B, 1, H, W, generator=gen) * 10 + 0.1)
            if sk == "neg":
                s.view(-1)[3] = -2.5
        go = torch.randn(B, 1, H, W, generator=gen)
        wb, o, gwb, gh = reference_grads_wrt_gate_wb(g, h, s, go, N, norm)
        out[name + "/gate_wb"] = wb.numpy()
        out[name + "/blur"] = h.numpy()
        if s is not None:
            out[name + "/sparse"] = s.numpy()
        out[name + "/grad_out"] = go.numpy()
        out[name + "/out"] = o.numpy()
        out[name + "/grad_gate_wb"] = gwb.numpy()
        out[name + "/grad_blur"] = gh.numpy()
        out[name + "/meta"] = np.array([B, H, W, N], dtype=np.int64)
        print(name, float(gwb.abs().max()), float(gh.abs().max()))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "cspn2d_grad_prenorm_golden.npz"), **out)


if __name__ == "__main__":
    main()
