"""Generates tests/golden/head_golden.npz: what the UNMODIFIED reference heads Simple_Gudi_UpConv_Block_Last_Layer
(/root/reference/cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206: Unpool :41-54 + 3x3 conv; instantiated :318-319 as gud_up_proj_layer6 = guidance and
gud_up_proj_layer5 = blur depth, called :372-373) return for seeded inputs and weights, and the reference's affinity_normalization
(cspn.py:85-144) of that guidance -- gate_wb, the contract of cspn_guidance_head_f32 with norm 8SUM / 8SUM_ABS.

Run in the authoring container only (the reference tree is not on the GPU box):
    python tests/golden/make_head_golden.py
The resulting .npz is committed; tests read it, never /root/reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_harness import reference_guidance_heads, reference_gate_wb  # noqa: E402

# name: (B, C, h, w, oheight, owidth)   (0, 0: no narrowing, as for the reference's own sizes where oheight = 2 h)
CASES = {
    "a_exact_2x": (2, 64, 7, 9, 14, 18),
    "b_no_narrow": (1, 64, 5, 6, 0, 0),
    "c_narrow_odd": (2, 64, 6, 70, 11, 139),     # crosses a 62-column stripe of the kernel; odd output sizes
    "d_few_channels": (1, 5, 9, 4, 18, 8),
    "e_one_pixel": (1, 64, 1, 1, 2, 2),
    "f_zero_patch": (1, 64, 8, 8, 16, 16),        # a patch of zero features: zero guidance -> 0 / 0 = NaN in gate_wb
}


def main():
    out = {}
    for name, (B, C, h, w, oh, ow) in CASES.items():
        gen = torch.Generator().manual_seed(sum(map(ord, name)))
        x = torch.randn(B, C, h, w, generator=gen)
        if name == "f_zero_patch":
            x[:, :, 2:6, 2:6] = 0
        w6 = torch.randn(8, C, 3, 3, generator=gen) / (3.0 * C ** 0.5)
        w5 = torch.randn(1, C, 3, 3, generator=gen) / (3.0 * C ** 0.5)
        g, b = reference_guidance_heads(x, w6, w5, oh, ow)
        out[name + "/x"], out[name + "/w6"], out[name + "/w5"] = x.numpy(), w6.numpy(), w5.numpy()
        out[name + "/meta"] = np.array([oh, ow], np.int32)
        out[name + "/guidance"], out[name + "/blur"] = g.numpy(), b.numpy()
        for norm in ("8sum", "8sum_abs"):
            wb, _ = reference_gate_wb(g, norm)
            out[name + "/gate_wb_" + norm] = wb.numpy().astype(np.float32)
        print(name, tuple(g.shape), tuple(b.shape), "nan in gate_wb:", int(np.isnan(out[name + "/gate_wb_8sum"]).sum()))
    path = os.path.join(HERE, "head_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
