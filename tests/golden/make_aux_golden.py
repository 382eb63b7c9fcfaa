"""tests/golden/make_aux_golden.py -- golden vectors for the steps next to the path, produced by the UNMODIFIED
reference files: cspn_pytorch/loss.py (Wighted_L1_Loss, its autograd gradient) and cspn_pytorch/utils.py (evaluate_error).
utils.py imports plotting / dataset modules at module scope that this container does not have (torchvision, PIL,
matplotlib, the repo's data_transform); the harness puts empty stand-ins into sys.modules -- the reference files themselves
are imported as they are.  Run in the build container (needs /root/reference):
    python tests/golden/make_aux_golden.py   ->  tests/golden/aux_golden.npz"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/cspn_pytorch"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux_golden.npz")
KEYS = ['MSE', 'RMSE', 'ABS_REL', 'LG10', 'MAE', 'DELTA1.02', 'DELTA1.05', 'DELTA1.10', 'DELTA1.25', 'DELTA1.25^2', 'DELTA1.25^3']


def _load(name, path, stubs=()):
    for s in stubs:
        if s not in sys.modules:
            m = types.ModuleType(s)
            m.__path__ = []   # so that "from torchvision import transforms" / "import matplotlib.pyplot" resolve
            sys.modules[s] = m
    for parent, child in (("torchvision", "transforms"), ("PIL", "Image"), ("PIL", "ImageOps"), ("matplotlib", "pyplot")):
        if parent in stubs:
            full = parent + "." + child
            if full not in sys.modules:
                sys.modules[full] = types.ModuleType(full)
            setattr(sys.modules[parent], child, sys.modules[full])
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    gen = torch.Generator().manual_seed(2024)
    for name, shape, frac, scale in (("nyu_batch", (2, 1, 114, 152), 0.9, 10.0), ("kitti_sparse_gt", (1, 1, 152, 608), 0.05, 80.0),
                                     ("tiny", (1, 1, 5, 7), 0.5, 10.0), ("nothing_valid", (1, 1, 6, 6), 0.0, 10.0)):
        gt = torch.rand(shape, generator=gen) * scale + 0.2
        gt = gt * (torch.rand(shape, generator=gen) < frac).float()
        pred = (gt + torch.randn(shape, generator=gen) * 0.3 * scale / 10).clamp_min(0.05) + (gt == 0).float() * torch.rand(shape, generator=gen)
        yield name, gt, pred


def main():
    loss_mod = _load("ref_loss", os.path.join(REF, "loss.py"))
    utils_mod = _load("ref_utils", os.path.join(REF, "utils.py"), stubs=("data_transform", "torchvision", "PIL", "matplotlib"))
    out = {}
    for name, gt, pred in cases():
        out[name + "/gt"] = gt.numpy()
        out[name + "/pred"] = pred.numpy()
        err = utils_mod.evaluate_error(gt, pred)
        out[name + "/metrics"] = np.array([float(err[k]) for k in KEYS], np.float64)
        if float((gt > 0.0001).sum()) > 0:
            p = pred.clone().requires_grad_(True)
            loss = loss_mod.Wighted_L1_Loss()(p, gt)
            loss.backward()
            out[name + "/loss"] = np.array([float(loss.detach())], np.float64)
            out[name + "/grad_pred"] = p.grad.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("metrics") or k.endswith("loss")})


if __name__ == "__main__":
    main()
