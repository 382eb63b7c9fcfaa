"""Generates tests/golden/cspn2d_golden.npz by running the UNMODIFIED reference
module (/root/reference/cspn_pytorch/models/cspn.py:14-83) on seeded CPU inputs.

Run in the authoring container only (the reference tree is not on the GPU box):
    python tests/golden/make_golden.py
The resulting .npz is committed; tests read it, never /root/reference.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.ref_harness import reference_forward  # noqa: E402

# name, B, H, W, n_iter, norm_type, sparse kind
CASES = [
    ("a_8sum_sparse_neg", 2, 19, 27, 24, "8sum", "neg"),
    ("b_abs_sparse", 1, 19, 27, 24, "8sum_abs", "pos"),
    ("c_wide_nosparse", 1, 33, 70, 12, "8sum", None),
    ("d_row_1x7", 1, 1, 7, 3, "8sum", None),
    ("e_3x3", 1, 3, 3, 5, "8sum_abs", "pos"),
    ("f_1x1_nan", 1, 1, 1, 2, "8sum", None),
    ("g_zero_guidance_patch", 1, 16, 18, 2, "8sum", "pos"),
    ("h_identity_n0", 1, 9, 11, 0, "8sum", "pos"),
    ("i_multiband_280", 1, 26, 280, 24, "8sum", "pos"),
    ("j_col_9x1", 1, 9, 1, 4, "8sum_abs", None),
    ("k_n1", 2, 10, 13, 1, "8sum", "pos"),
    ("l_n30_abs", 1, 40, 37, 30, "8sum_abs", "pos"),
]


def make_inputs(seed, B, H, W, sparse_kind, name):
    gen = torch.Generator().manual_seed(seed)
    g = torch.randn(B, 8, H, W, generator=gen)
    h = torch.rand(B, 1, H, W, generator=gen) * 10
    s = None
    if sparse_kind is not None:
        m = (torch.rand(B, 1, H, W, generator=gen) < 0.08).float()
        s = m * (torch.rand(B, 1, H, W, generator=gen) * 10 + 0.1)
        if sparse_kind == "neg":
            s.view(-1)[3] = -2.5  # sign() -> -1 (cspn.py:64)
    if "zero_guidance" in name:
        g[:, :, 3:8, 4:9] = 0.0  # 0/0 -> NaN region (cspn.py:138)
    return g, h, s


def main():
    out = {}
    for idx, (name, B, H, W, N, norm, sk) in enumerate(CASES):
        g, h, s = make_inputs(100 + idx, B, H, W, sk, name)
        ref = reference_forward(g, h, s, N, norm)
        out[name + "/guidance"] = g.numpy()
        out[name + "/blur"] = h.numpy()
        if s is not None:
            out[name + "/sparse"] = s.numpy()
        out[name + "/out"] = ref.numpy().astype(np.float32)
        out[name + "/meta"] = np.array([B, H, W, N, 0 if norm == "8sum" else 1], dtype=np.int64)
        print(name, tuple(ref.shape), "nan:", int(torch.isnan(ref).sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cspn2d_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
