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
