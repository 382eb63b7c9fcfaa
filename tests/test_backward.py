"""Backward of the 2D op: the gradient torch autograd computes through the reference forward
(cspn.py:42-83), which reference train.py:196-198 back-propagates through.
CPU: the numpy restatement (oracle/backward.py) against gradients the UNMODIFIED reference produced
(tests/golden/cspn2d_grad_golden.npz, made by tests/golden/make_grad_golden.py).
GPU: the HIP kernels through the C ABI / the autograd Function against those vectors and against the restatement."""
import os

import numpy as np
import pytest
import torch

from helpers import make_inputs
from oracle.backward import cspn2d_backward_oracle

GFLOOR = 5e-6
GTOL = 2e-4  # relative to max|grad|: 1e-4 forward tolerance with headroom for the longer accumulation chains
NORMS = {0: "8sum", 1: "8sum_abs"}


def _golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cspn2d_grad_golden.npz"))
    for n in sorted({k.split("/")[0] for k in z.files}):
        c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}
        yield n, c


def _check(a, b, what=""):
    """gradient parity, element-wise: |a - b| <= GFLOOR * max|ref| + GTOL * |ref| (the form of the forward's
    helpers.assert_close_tight: dL/dguidance carries 1 / sum|g| tails, so a max-norm check alone leaves everything small
    unchecked) AND max-norm <= GTOL.  GFLOOR = 5e-6: with the forward's 1e-6 the 3 x 70 x 512 '8sum_abs' case exceeds the bound
    2.1x at elements below 1 % of max|grad| (max-norm error 3.5e-6) -- fp32 cancellation in
    dG_k = dw_k / S - sign(G_k) sum_j dw_j G_j / S^2 summed over 24 levels, an absolute error of ~2e-6 max|grad| wherever the two
    terms nearly cancel; the float64 oracle has none of it, the reference's own fp32 autograd has the same."""
    from helpers import assert_close
    assert_close(a, b, what, rtol=GTOL, atol_frac=GFLOOR)
    return True


def _err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.array_equal(np.isfinite(a), np.isfinite(b))
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_backward_oracle_vs_reference_autograd():
    for name, c in _golden():
        B, H, W, N, norm = (int(v) for v in c["meta"])
        o, gg, gh = cspn2d_backward_oracle(c["guidance"], c["blur"], c.get("sparse"), c["grad_out"], N, NORMS[norm])
        assert _err(o, c["out"]) <= 1e-5, name
        assert _err(gg, c["grad_guidance"]) <= 1e-5, name
        assert _err(gh, c["grad_blur"]) <= 1e-5, name


@pytest.mark.gpu
def test_hip_backward_vs_reference_autograd_goldens():
    import cspn_amd
    for name, c in _golden():
        B, H, W, N, norm = (int(v) for v in c["meta"])
        t = {k: torch.from_numpy(v).cuda() for k, v in c.items() if k != "meta"}
        gg, gh = cspn_amd.cspn2d_backward(t["guidance"], t["blur"], t.get("sparse"), t["grad_out"], N, NORMS[norm])
        torch.cuda.synchronize()
        assert _check(gg.cpu().numpy(), c["grad_guidance"]), name
        assert _check(gh.cpu().numpy(), c["grad_blur"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,N,norm,sp", [(2, 37, 53, 24, "8sum", True), (1, 64, 320, 24, "8sum_abs", True),
                                             (3, 20, 256, 12, "8sum", False), (1, 5, 9, 1, "8sum_abs", False),
                                             (1, 30, 40, 30, "8sum", True),
                                             # the checkpointed final pass at awkward tile geometry: one tile row and a bit, a last
                                             # tile column of 8 pixels, image borders inside every block
                                             (2, 41, 260, 24, "8sum", True), (1, 89, 300, 24, "8sum_abs", False),
                                             (1, 3, 256, 24, "8sum", False),    # an image lower than the recompute halo
                                             # round 5: n_iter = 4, 8 .. 20 on the checkpointed ring path too (the sweeps run their 24 levels,
                                             # H_n / A_0 are checkpoint planes, the final pass runs n_iter / 4 segments)
                                             (2, 41, 260, 4, "8sum", True), (1, 70, 304, 8, "8sum_abs", False), (2, 33, 516, 16, "8sum", True),
                                             (1, 50, 256, 20, "8sum_abs", True)])
def test_hip_backward_vs_oracle_and_through_autograd(B, H, W, N, norm, sp):
    import cspn_amd
    g, h, s = make_inputs(B, H, W, seed=B + H + W + N, sparse=sp, neg=sp)
    go = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(5))
    _, rgg, rgh = cspn2d_backward_oracle(g.numpy(), h.numpy(), None if s is None else s.numpy(), go.numpy(), N, norm)
    gd, hd = g.cuda().requires_grad_(True), h.cuda().requires_grad_(True)
    m = cspn_amd.Affinity_Propagate(N, 3, norm)
    out = m(gd, hd, None if s is None else s.cuda())
    out.backward(go.cuda())
    assert _check(gd.grad.cpu().numpy(), rgg)
    assert _check(hd.grad.cpu().numpy(), rgh)
    # only one input needs a gradient: the other output is skipped
    hd2 = h.cuda().requires_grad_(True)
    m(g.cuda(), hd2, None if s is None else s.cuda()).backward(go.cuda())
    assert _check(hd2.grad.cpu().numpy(), rgh)


@pytest.mark.gpu
@pytest.mark.parametrize("norm,sp", [("8sum", True), ("8sum_abs", False)])
@pytest.mark.parametrize("N", [24, 12])
def test_training_mode_history_matches_recompute_path(norm, sp, N):
    """the forward that keeps its checkpoints + the backward that starts from them == plain forward + recomputing backward"""
    import cspn_amd
    B, H, W = 3, 70, 512
    assert cspn_amd.cspn2d_history_bytes(B, H, W, N) > 0 and cspn_amd.cspn2d_history_bytes(B, H, 64, N) == 0
    g, h, s = make_inputs(B, H, W, seed=21, sparse=sp, neg=sp)
    go = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(6)).cuda()
    gd, hd, sd = g.cuda(), h.cuda(), None if s is None else s.cuda()
    out_ref = cspn_amd.cspn2d_forward(gd, hd, sd, N, norm)
    gg_ref, gh_ref = cspn_amd.cspn2d_backward(gd, hd, sd, go, N, norm)
    out, hist = cspn_amd.cspn2d_forward_with_history(gd, hd, sd, N, norm)
    gg, gh = cspn_amd.cspn2d_backward_from_history(gd, hd, sd, go, hist, N, norm)
    # (the plain forward streams the pieces of the linear plan, the history forward the band groups: a row lands in another ring
    # slot, i.e. the same arithmetic in another summation order)
    assert float((out - out_ref).abs().max()) <= 4e-6 * float(out_ref.abs().max())
    assert torch.equal(gg, gg_ref) and torch.equal(gh, gh_ref)
    # through the module: training keeps the history, and the gradients agree with the oracle
    _, rgg, rgh = cspn2d_backward_oracle(g.numpy(), h.numpy(), None if s is None else s.numpy(), go.cpu().numpy(), N, norm)
    g1, h1 = g.cuda().requires_grad_(True), h.cuda().requires_grad_(True)
    m = cspn_amd.Affinity_Propagate(N, 3, norm)
    o = m(g1, h1, sd)
    assert o.grad_fn is not None and len(o.grad_fn.saved_tensors) == 4 and o.grad_fn.saved_tensors[3] is not None
    o.backward(go)
    assert _check(g1.grad.cpu().numpy(), rgg) and _check(h1.grad.cpu().numpy(), rgh)
    m.keep_history = False
    g2, h2 = g.cuda().requires_grad_(True), h.cuda().requires_grad_(True)
    m(g2, h2, sd).backward(go)
    assert torch.equal(g2.grad, g1.grad) and torch.equal(h2.grad, h1.grad)


@pytest.mark.gpu
def test_hip_backward_full_size_is_linear_in_grad_out():
    import cspn_amd
    B, H, W = 4, 304, 1216
    g, h, s = make_inputs(B, H, W, seed=11, sparse=True)
    gd, hd, sd = g.cuda(), h.cuda(), s.cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    a, b = torch.randn(B, 1, H, W, generator=gen, device="cuda"), torch.randn(B, 1, H, W, generator=gen, device="cuda")
    ga, ha = cspn_amd.cspn2d_backward(gd, hd, sd, a, 24, "8sum")
    gb, hb = cspn_amd.cspn2d_backward(gd, hd, sd, b, 24, "8sum")
    gc, hc = cspn_amd.cspn2d_backward(gd, hd, sd, 2.0 * a - 0.5 * b, 24, "8sum")
    assert torch.isfinite(gc).all() and torch.isfinite(hc).all()
    assert float((gc - (2.0 * ga - 0.5 * gb)).abs().max() / gc.abs().max()) <= GTOL
    assert float((hc - (2.0 * ha - 0.5 * hb)).abs().max() / hc.abs().max()) <= GTOL
    # masked pixels: their output is H_0 itself, so d out / d blur carries the full grad_out there
    one = cspn2d_backward_check_mask(cspn_amd, gd, hd, sd)
    assert one


def cspn2d_backward_check_mask(cspn_amd, gd, hd, sd):
    # with grad_out = indicator of the masked pixels only, every masked pixel's own blur gradient contains that 1
    mask = (sd != 0).float()
    _, hb = cspn_amd.cspn2d_backward(gd, hd, sd, mask, 24, "8sum")
    return bool(((hb - 1.0)[mask.bool()] >= -1e-4).all())


@pytest.mark.gpu
def test_forward_and_backward_vs_stock_torch_autograd_at_full_width():
    """a plain-torch restatement of the reference ops (tools/torch_path.py) with torch autograd on the GPU: forward and both
    gradients at KITTI width, larger than the numpy restatement is practical for"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.torch_path import cspn2d_torch
    import cspn_amd
    B, H, W, N = 2, 96, 1216, 24
    g, h, s = make_inputs(B, H, W, seed=31, sparse=True, neg=True, depth_scale=80.0)
    go = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(8)).cuda()
    g0, h0 = g.cuda().requires_grad_(True), h.cuda().requires_grad_(True)
    ref = cspn2d_torch(g0, h0, s.cuda(), N, "8sum")
    ref.backward(go)
    g1, h1 = g.cuda().requires_grad_(True), h.cuda().requires_grad_(True)
    out = cspn_amd.Affinity_Propagate(N, 3, "8sum")(g1, h1, s.cuda())
    out.backward(go)
    assert _err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-4
    assert _check(g1.grad.cpu().numpy(), g0.grad.cpu().numpy())
    assert _check(h1.grad.cpu().numpy(), h0.grad.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("sp", [False, True])
def test_backward_norm_none_on_the_checkpointed_path_vs_torch_autograd(sp):
    """norm NONE (gates used as given, centre-sited: the Paddle-style 2D contract; c' = m H_0 in the recomputing final pass) at a
    size the assembly sweeps take, against torch autograd through the same recurrence"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.torch_path import _gather8
    import cspn_amd
    B, H, W, N = 2, 50, 272, 24
    gen = torch.Generator().manual_seed(17)
    g = torch.rand(B, 8, H, W, generator=gen)
    g = (g / g.sum(1, keepdim=True)).cuda()
    h = (torch.rand(B, 1, H, W, generator=gen) * 10).cuda()
    s = ((torch.rand(B, 1, H, W, generator=gen) < 0.02).float() * 3.0).cuda() if sp else None
    go = torch.randn(B, 1, H, W, generator=gen).cuda()
    g0, h0 = g.clone().requires_grad_(True), h.clone().requires_grad_(True)
    cur = h0
    for _ in range(N):
        cur = (g0 * _gather8(cur)).sum(1, keepdim=True)
        if s is not None:
            m = s.sign()
            cur = (1 - m) * cur + m * h0
    cur.backward(go)
    gg, gh = cspn_amd.cspn2d_backward(g, h, s, go, N, "none")
    assert _check(gg.cpu().numpy(), g0.grad.cpu().numpy())
    assert _check(gh.cpu().numpy(), h0.grad.cpu().numpy())


# ---- the pre-normalised input contract (CSPN_NORM_PRENORM, round 6): gradient w.r.t. the reference's gate_wb ------------------------------
def _golden_prenorm():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cspn2d_grad_prenorm_golden.npz"))
    for n in sorted({k.split("/")[0] for k in z.files}):
        yield n, {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}


def test_prenorm_backward_oracle_vs_reference_autograd():
    """oracle/backward.py with norm 'prenorm' against the unmodified reference's autograd gradient w.r.t. the gate_wb its
    affinity_normalization returned (tests/golden/make_grad_prenorm_golden.py)"""
    n = 0
    for name, c in _golden_prenorm():
        B, H, W, N = (int(v) for v in c["meta"])
        o, gg, gh = cspn2d_backward_oracle(c["gate_wb"], c["blur"], c.get("sparse"), c["grad_out"], N, "prenorm")
        assert _err(o, c["out"]) <= 1e-5 and _err(gg, c["grad_gate_wb"]) <= 1e-5 and _err(gh, c["grad_blur"]) <= 1e-5, name
        n += 1
    assert n == 7


def test_prenorm_backward_oracle_vs_live_reference():
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip("reference tree not present")
    gen = torch.Generator().manual_seed(31)
    g = torch.randn(1, 8, 7, 9, generator=gen)
    h = torch.rand(1, 1, 7, 9, generator=gen) * 5
    s = (torch.rand(1, 1, 7, 9, generator=gen) < 0.2).float() * 2.0
    go = torch.randn(1, 1, 7, 9, generator=gen)
    wb, o, gwb, gh = ref_harness.reference_grads_wrt_gate_wb(g, h, s, go, 5, "8sum_abs")
    ro, rgg, rgh = cspn2d_backward_oracle(wb.numpy(), h.numpy(), s.numpy(), go.numpy(), 5, "prenorm")
    assert _err(ro, o.numpy()) <= 1e-5 and _err(rgg, gwb.numpy()) <= 1e-5 and _err(rgh, gh.numpy()) <= 1e-5


@pytest.mark.gpu
def test_hip_prenorm_backward_vs_reference_autograd_goldens():
    import cspn_amd
    for name, c in _golden_prenorm():
        B, H, W, N = (int(v) for v in c["meta"])
        t = {k: torch.from_numpy(v).cuda() for k, v in c.items() if k != "meta"}
        gg, gh = cspn_amd.cspn2d_backward(t["gate_wb"], t["blur"], t.get("sparse"), t["grad_out"], N, "prenorm")
        torch.cuda.synchronize()
        assert _check(gg.cpu().numpy(), c["grad_gate_wb"]), name
        assert _check(gh.cpu().numpy(), c["grad_blur"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,N,sp", [(2, 37, 53, 24, True), (3, 20, 256, 12, False), (2, 41, 260, 24, True), (1, 89, 300, 24, False),
                                        (1, 3, 256, 24, False), (2, 33, 516, 16, True), (1, 30, 40, 30, True), (2, 70, 512, 24, True)])
def test_hip_prenorm_backward_vs_oracle_and_through_autograd(B, H, W, N, sp):
    """every backward path (one launch per step; the two ring sweeps + recomputing final pass, n_iter = 4 .. 24; training mode with kept
    checkpoints) with the reference's gate_wb as the input, against the numpy restatement -- and the chain rule closed: the gradient of the
    raw contract = this gradient pushed through the stand-alone normalisation by torch autograd"""
    import cspn_amd
    g, h, s = make_inputs(B, H, W, seed=B + H + W + N, sparse=sp, neg=sp)
    go = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(5))
    wb = cspn_amd.cspn2d_normalize(g.cuda(), "8sum")
    sd = None if s is None else s.cuda()
    _, rgg, rgh = cspn2d_backward_oracle(wb.cpu().numpy(), h.numpy(), None if s is None else s.numpy(), go.numpy(), N, "prenorm")
    gg, gh = cspn_amd.cspn2d_backward(wb, h.cuda(), sd, go.cuda(), N, "prenorm")
    assert _check(gg.cpu().numpy(), rgg) and _check(gh.cpu().numpy(), rgh)
    wbd, hd = wb.clone().requires_grad_(True), h.cuda().requires_grad_(True)
    out = cspn_amd.propagate_prenorm(wbd, hd, sd, N)
    ref = cspn_amd.cspn2d_forward(wb, h.cuda(), sd, N, "prenorm")
    assert float((out.detach() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    out.backward(go.cuda())
    assert _check(wbd.grad.cpu().numpy(), rgg) and _check(hd.grad.cpu().numpy(), rgh)
    if cspn_amd.cspn2d_history_bytes(B, H, W, N) > 0:   # training mode == recomputing path, bit for bit
        assert torch.equal(wbd.grad, gg) and torch.equal(hd.grad, gh)
    # only one input needs a gradient
    hd2 = h.cuda().requires_grad_(True)
    cspn_amd.propagate_prenorm(wb, hd2, sd, N).backward(go.cuda())
    assert _check(hd2.grad.cpu().numpy(), rgh)
