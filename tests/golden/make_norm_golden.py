"""Generates tests/golden/cspn2d_norm_golden.npz: `gate_wb` (and `gate_sum`) as the UNMODIFIED reference's
affinity_normalization (/root/reference/cspn_pytorch/models/cspn.py:85-144) returns them, for the guidance tensors of
tests/golden/cspn2d_golden.npz (same cases, same inputs: only the new arrays are stored here).  gate_wb is cropped to the image
(the reference crops after the product, cspn.py:72): [B,8,H,W], normalised and consumer-sited -- the contract of
CSPN_NORM_PRENORM / cspn2d_normalize_f32 (include/cspn_amd.h).

Run in the authoring container only (the reference tree is not on the GPU box):
    python tests/golden/make_norm_golden.py
The resulting .npz is committed; tests read it, never /root/reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_harness import reference_gate_wb  # noqa: E402

CASES = ["a_8sum_sparse_neg", "b_abs_sparse", "d_row_1x7", "e_3x3", "f_1x1_nan", "g_zero_guidance_patch", "i_multiband_280", "j_col_9x1",
         "l_n30_abs"]


def main():
    src = np.load(os.path.join(HERE, "cspn2d_golden.npz"))
    out = {}
    for name in CASES:
        norm = "8sum_abs" if int(src[name + "/meta"][4]) else "8sum"
        wb, gs = reference_gate_wb(torch.from_numpy(src[name + "/guidance"]), norm)
        out[name + "/gate_wb"] = wb.numpy().astype(np.float32)
        out[name + "/gate_sum"] = gs.numpy().astype(np.float32)
        print(name, norm, tuple(wb.shape), "nan:", int(torch.isnan(wb).sum()))
    path = os.path.join(HERE, "cspn2d_norm_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
