"""Generates tests/golden/head_grad_golden.npz: the gradients torch autograd computes through the UNMODIFIED reference heads
Simple_Gudi_UpConv_Block_Last_Layer (/root/reference/cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206, Unpool :41-54; called :372-373) with respect to
their input feature map and their two conv weights, for seeded inputs and output gradients (oracle/ref_harness.py reference_guidance_heads_grads).
Authoring container only:
    python tests/golden/make_head_grad_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_harness import reference_guidance_heads_grads  # noqa: E402

# name: (B, C, h, w, oheight, owidth)
CASES = {
    "a_exact_2x": (2, 64, 7, 9, 14, 18),
    "b_no_narrow": (1, 64, 5, 6, 0, 0),
    "c_narrow_odd": (2, 64, 6, 70, 11, 139),
    "d_few_channels": (1, 5, 9, 4, 18, 8),
    "e_one_pixel": (1, 64, 1, 1, 2, 2),
    "f_narrow_more": (1, 16, 8, 40, 13, 77),      # the last input row / column lies beyond the narrowed output: no gradient reaches it
}


def main():
    out = {}
    for name, (B, C, h, w, oh, ow) in CASES.items():
        gen = torch.Generator().manual_seed(sum(map(ord, name)) + 7)
        x = torch.randn(B, C, h, w, generator=gen)
        w6 = torch.randn(8, C, 3, 3, generator=gen) / (3.0 * C ** 0.5)
        w5 = torch.randn(1, C, 3, 3, generator=gen) / (3.0 * C ** 0.5)
        H, W = (oh, ow) if (oh and ow) else (2 * h, 2 * w)
        gg = torch.randn(B, 8, H, W, generator=gen)
        gb = torch.randn(B, 1, H, W, generator=gen)
        dx, dw6, dw5 = reference_guidance_heads_grads(x, w6, w5, gg, gb, oh, ow)
        for k, v in (("x", x), ("w6", w6), ("w5", w5), ("grad_guidance", gg), ("grad_blur", gb), ("grad_x", dx), ("grad_w6", dw6), ("grad_w5", dw5)):
            out[name + "/" + k] = v.numpy()
        out[name + "/meta"] = np.array([oh, ow], np.int32)
        print(name, tuple(dx.shape), float(dx.abs().max()), float(dw6.abs().max()), float(dw5.abs().max()))
    path = os.path.join(HERE, "head_grad_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
