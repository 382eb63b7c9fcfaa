"""on-device metrics + masked L1 loss (reference utils.py:19-47, loss.py:16-23) and Unpool
(reference torch_resnet_cspn_nyu.py:41-54) through the C ABI, against plain-torch restatements of those reference lines
(floating-point kernels: torch fp32 reference; tolerance 1e-5 relative for the means, exact for counts and Unpool)."""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_evaluate_error(gt, pred):  # utils.py:19-47, line by line
    mask = gt > 0.0001
    err = {k: 0 for k in ['MSE', 'RMSE', 'ABS_REL', 'LG10', 'MAE', 'DELTA1.02', 'DELTA1.05', 'DELTA1.10', 'DELTA1.25',
                          'DELTA1.25^2', 'DELTA1.25^3']}
    p, g = pred[mask].double(), gt[mask].double()
    n = g.numel()
    if n > 0:
        d = (g - p).abs()
        err['MSE'] = float((d ** 2).sum() / n)
        err['RMSE'] = math.sqrt(err['MSE'])
        err['MAE'] = float(d.sum() / n)
        err['ABS_REL'] = float((d / g).sum() / n)
        r = torch.max(gt[mask] / pred[mask], pred[mask] / gt[mask])  # fp32 ratios decide the thresholds (utils.py:38-40)
        for k, t in (('DELTA1.02', 1.02), ('DELTA1.05', 1.05), ('DELTA1.10', 1.10), ('DELTA1.25', 1.25),
                     ('DELTA1.25^2', 1.25 ** 2), ('DELTA1.25^3', 1.25 ** 3)):
            err[k] = float((r < t).sum()) / n
    return err


@pytest.mark.parametrize("shape,frac", [((4, 1, 228, 304), 0.9), ((2, 1, 37, 53), 0.3), ((1, 1, 5, 7), 0.0),
                                        ((8, 1, 304, 1216), 0.05)])
def test_evaluate_error_matches_reference_formulas(shape, frac):
    from cspn_amd.train_utils import evaluate_error
    gen = torch.Generator().manual_seed(shape[2])
    gt = torch.rand(shape, generator=gen) * 10 * (torch.rand(shape, generator=gen) < frac).float()
    pred = (gt + torch.randn(shape, generator=gen) * 0.3).abs() + 0.05
    ref = ref_evaluate_error(gt, pred)
    got = evaluate_error(gt.to(DEV), pred.to(DEV))
    assert set(got) == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (k, got[k], ref[k])


def test_weighted_l1_loss_value_and_gradient():
    from cspn_amd.train_utils import Wighted_L1_Loss
    gen = torch.Generator().manual_seed(4)
    label = torch.rand(3, 1, 60, 80, generator=gen) * 10 * (torch.rand(3, 1, 60, 80, generator=gen) < 0.4).float()
    pred = torch.rand(3, 1, 60, 80, generator=gen) * 10
    # reference loss.py:16-23
    p0 = pred.clone().requires_grad_(True)
    m = label > 0.0001
    ref = (p0[m] - label[m]).abs().sum() / m.sum()
    (3.0 * ref).backward()
    p1 = pred.to(DEV).requires_grad_(True)
    loss = Wighted_L1_Loss()(p1, label.to(DEV))
    (3.0 * loss).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref)
    assert torch.allclose(p1.grad.cpu(), p0.grad, rtol=1e-5, atol=1e-9)
    # nothing valid: the reference divides 0 by 0
    assert math.isnan(float(Wighted_L1_Loss()(pred.to(DEV), torch.zeros_like(label).to(DEV))))


@pytest.mark.parametrize("N,C,H,W,S", [(2, 3, 5, 7, 2), (1, 64, 57, 76, 2), (1, 1, 4, 4, 3)])
def test_unpool_matches_conv_transpose(N, C, H, W, S):
    from cspn_amd.train_utils import Unpool
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(N + C))
    w = torch.zeros(C, 1, S, S)
    w[:, :, 0, 0] = 1  # torch_resnet_cspn_nyu.py:49-51
    x0 = x.clone().requires_grad_(True)
    ref = TF.conv_transpose2d(x0, w, stride=S, groups=C)  # :53-54
    go = torch.randn_like(ref)
    ref.backward(go)
    x1 = x.to(DEV).requires_grad_(True)
    out = Unpool(C, S)(x1)
    out.backward(go.to(DEV))
    assert out.shape == ref.shape
    assert torch.equal(out.cpu(), ref.detach())
    assert torch.equal(x1.grad.cpu(), x0.grad)


@pytest.mark.gpu
def test_sparse_depth_sampling_on_device_matches_loader_statistics():
    """reference createSparseDepthImage (nyu_dataset_loader.py:135-144: p = n_sample / n_pixels; kitti_dataset_loader.py:
    138-148: p = n_sample / n_valid): sparse = depth * bernoulli(p).  Different generator than torch.bernoulli, so the check
    is statistical: counts within 5 sigma of n * p per image, values are the depth's, masks independent across images and
    seeds, reproducible per seed, spatially uniform."""
    import cspn_amd.train_utils as tu
    torch.manual_seed(0)
    B, H, W, n_sample = 12, 228, 304, 500
    depth = (torch.rand(B, 1, H, W, device="cuda") * 10 + 0.5)
    depth[:, :, :40] = 0.0                                   # invalid rows (KITTI-like): depth <= 1e-4
    n_pix, n_valid = H * W, (H - 40) * W
    for mode, p in (("nyu", n_sample / n_pix), ("kitti", n_sample / n_valid)):
        s = tu.createSparseDepthImage(depth, n_sample, mode=mode, seed=123)
        assert s.shape == depth.shape and s.is_cuda
        kept = s != 0
        assert torch.equal(s[kept], depth[kept])             # values are the depth's, everything else exactly 0
        assert not kept[:, :, :40].any()                     # zero depth stays zero
        cnt = kept.flatten(1).sum(1).double().cpu()
        exp = n_valid * p                                    # only valid pixels can show up in `kept`
        sig = (exp * (1 - p)) ** 0.5
        assert float((cnt - exp).abs().max()) <= 5 * sig, (mode, cnt.tolist(), exp)
        assert abs(float(cnt.mean()) - exp) <= 5 * sig / B ** 0.5
        # reproducible per seed, different across seeds and across images
        assert torch.equal(s, tu.createSparseDepthImage(depth, n_sample, mode=mode, seed=123))
        s2 = tu.createSparseDepthImage(depth, n_sample, mode=mode, seed=124)
        both = ((s != 0) & (s2 != 0)).flatten(1).sum(1).double().cpu()
        assert float(both.max()) <= exp * p + 5 * (exp * p) ** 0.5 + 3   # overlap of two independent masks ~ n p^2
        m0, m1 = kept[0], kept[1]
        assert float((m0 & m1).sum()) <= exp * p + 5 * (exp * p) ** 0.5 + 3
        # spatial uniformity over the valid area: 4 x 4 cells, chi-square with 15 dof (99.99 % quantile: 44.3)
        cells = kept[:, 0, 40:40 + 188, :304].reshape(B, 4, 47, 4, 76).sum((0, 2, 4)).double().cpu().flatten()
        e = cells.sum() / 16
        assert float(((cells - e) ** 2 / e).sum()) <= 44.3
    # the NYU setting of BASELINE config 2/4: about 500 points per image
    s = tu.createSparseDepthImage(torch.rand(4, 1, 304, 1216, device="cuda") + 1.0, 500, seed=7)
    c = (s != 0).flatten(1).sum(1)
    assert int(c.min()) > 380 and int(c.max()) < 620
    # odd sizes (hw % 4 != 0) and a single image without channel dim
    s = tu.createSparseDepthImage(torch.ones(7, 9, device="cuda"), 63)
    assert s.shape == (7, 9) and float(s.sum()) == 63.0      # p = 1: keeps everything


def _aux_golden():
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aux_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def test_metrics_and_loss_vs_vectors_of_the_reference_files():
    """tests/golden/aux_golden.npz was produced by the UNMODIFIED reference cspn_pytorch/utils.py (evaluate_error) and loss.py
    (Wighted_L1_Loss + its autograd gradient) via tests/golden/make_aux_golden.py: the device mirrors must reproduce them"""
    import numpy as np
    import cspn_amd.train_utils as tu
    keys = ['MSE', 'RMSE', 'ABS_REL', 'LG10', 'MAE', 'DELTA1.02', 'DELTA1.05', 'DELTA1.10', 'DELTA1.25', 'DELTA1.25^2', 'DELTA1.25^3']
    for name, c in _aux_golden().items():
        gt, pred = torch.from_numpy(c["gt"]).to(DEV), torch.from_numpy(c["pred"]).to(DEV)
        err = tu.evaluate_error(gt, pred)
        for i, k in enumerate(keys):
            ref = float(c["metrics"][i])
            assert abs(err[k] - ref) <= 2e-5 * max(abs(ref), 1e-3), (name, k, err[k], ref)
        if "loss" in c:
            p = pred.clone().requires_grad_(True)
            loss = tu.Wighted_L1_Loss()(p, gt)
            loss.backward()
            assert abs(float(loss) - float(c["loss"][0])) <= 1e-5 * float(c["loss"][0]), name
            assert np.allclose(p.grad.cpu().numpy(), c["grad_pred"], rtol=1e-5, atol=1e-9), name
