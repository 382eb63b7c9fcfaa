"""Generates tests/golden/cspn2d_grad_golden.npz: gradients of the UNMODIFIED reference module
(/root/reference/cspn_pytorch/models/cspn.py:42-83) by torch autograd on seeded CPU inputs -- what
reference train.py:196-198 back-propagates through.  Authoring container only:
    python tests/golden/make_grad_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.ref_harness import reference_grads  # noqa: E402

# name, B, H, W, n_iter, norm_type, sparse kind
CASES = [
    ("a_8sum_sparse_neg", 2, 11, 14, 6, "8sum", "neg"),
    ("b_abs_sparse", 1, 13, 17, 24, "8sum_abs", "pos"),
    ("c_nosparse", 1, 9, 40, 12, "8sum", None),
    ("d_row_1x7", 1, 1, 7, 3, "8sum", None),
    ("e_3x3_abs", 1, 3, 3, 5, "8sum_abs", "pos"),
    ("f_n1", 2, 6, 5, 1, "8sum", "pos"),
    ("g_wide_band", 1, 8, 272, 24, "8sum", "pos"),
]


def main():
    out = {}
    for idx, (name, B, H, W, N, norm, sk) in enumerate(CASES):
        gen = torch.Generator().manual_seed(700 + idx)
        g = torch.randn(B, 8, H, W, generator=gen)
        h = torch.rand(B, 1, H, W, generator=gen) * 10
        s = None
        if sk is not None:
            m = (torch.rand(B, 1, H, W, generator=gen) < 0.08).float()
            s = m * (torch.rand(B, 1, H, W, generator=gen) * 10 + 0.1)
            if sk == "neg":
                s.view(-1)[3] = -2.5
        go = torch.randn(B, 1, H, W, generator=gen)
        o, gg, gh = reference_grads(g, h, s, go, N, norm)
        out[name + "/guidance"] = g.numpy()
        out[name + "/blur"] = h.numpy()
        if s is not None:
            out[name + "/sparse"] = s.numpy()
        out[name + "/grad_out"] = go.numpy()
        out[name + "/out"] = o.numpy()
        out[name + "/grad_guidance"] = gg.numpy()
        out[name + "/grad_blur"] = gh.numpy()
        out[name + "/meta"] = np.array([B, H, W, N, 0 if norm == "8sum" else 1], dtype=np.int64)
        print(name, float(gg.abs().max()), float(gh.abs().max()))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "cspn2d_grad_golden.npz"), **out)


if __name__ == "__main__":
    main()
