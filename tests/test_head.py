"""The producer of the propagation's inputs (producer half): cspn_guidance_head_f32 = both reference heads Simple_Gudi_UpConv_Block_Last_Layer
(cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206: Unpool :41-54 + bias-free 3x3 conv; gud_up_proj_layer6 / gud_up_proj_layer5 of :318-319, called :372-373) as
one kernel, optionally with affinity_normalization (cspn.py:85-144) fused behind it (gate_wb).  Golden vectors: tests/golden/head_golden.npz, produced by the
UNMODIFIED reference classes (tests/golden/make_head_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import guidance_head_oracle, guidance_head_backward_oracle, cspn2d_gate_wb_oracle, cspn2d_oracle  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "head_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def _case(name):
    g = {k.split("/")[1]: GOLD[k] for k in GOLD.files if k.startswith(name + "/")}
    return g, int(g["meta"][0]), int(g["meta"][1])


def _rel(a, b):
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    return float(np.abs(a[fin] - b[fin]).max() / max(1e-30, np.abs(b[fin]).max())) if fin.any() else 0.0


@pytest.mark.parametrize("name", NAMES)
def test_head_oracle_vs_reference_golden(name):
    """the numpy restatement (oracle/oracle.py guidance_head_oracle) is pinned to what the unmodified reference layers returned"""
    g, oh, ow = _case(name)
    guid, blur = guidance_head_oracle(g["x"], g["w6"], g["w5"], oh, ow)
    assert guid.shape == g["guidance"].shape and blur.shape == g["blur"].shape
    assert _rel(guid, g["guidance"]) <= 2e-6 and _rel(blur, g["blur"]) <= 2e-6
    # and the normalisation behind it: the oracle's gate_wb of the oracle's guidance against the reference's gate_wb of the reference's guidance
    for norm in ("8sum", "8sum_abs"):
        wb = cspn2d_gate_wb_oracle(g["guidance"], norm)
        assert _rel(wb, g["gate_wb_" + norm]) <= 2e-6


def test_head_oracle_vs_live_reference():
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip("/root/reference not present (GPU box)")
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 6, 5, generator=gen)
    w6, w5 = torch.randn(8, 64, 3, 3, generator=gen) * 0.05, torch.randn(1, 64, 3, 3, generator=gen) * 0.05
    rg, rb = ref_harness.reference_guidance_heads(x, w6, w5, 12, 10)
    og, ob = guidance_head_oracle(x.numpy(), w6.numpy(), w5.numpy(), 12, 10)
    assert _rel(og, rg.numpy()) <= 2e-6 and _rel(ob, rb.numpy()) <= 2e-6


# ---------------------------------------------------------------------------------------------------------------- GPU
def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_head_kernel_vs_reference_golden(name):
    import cspn_amd
    from cspn_amd.train_utils import guidance_heads
    g, oh, ow = _case(name)
    x, w6, w5 = _dev(g["x"]), _dev(g["w6"]), _dev(g["w5"])
    guid, blur = guidance_heads(x, w6, w5, oh, ow)
    torch.cuda.synchronize()
    assert _rel(guid.cpu().numpy(), g["guidance"]) <= 1e-5 and _rel(blur.cpu().numpy(), g["blur"]) <= 1e-5
    g2, none = guidance_heads(x, w6, None, oh, ow)                      # guidance only
    assert none is None and torch.equal(g2, guid)
    for norm in ("8sum", "8sum_abs"):
        wb, blur2 = guidance_heads(x, w6, w5, oh, ow, norm_type=norm)
        torch.cuda.synchronize()
        assert torch.equal(blur2, blur)      # (one kernel, two store patterns)
        ref = g["gate_wb_" + norm]
        got = wb.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name      # 0 / 0 = NaN exactly where the reference has it (cspn.py:138)
        # (a raw guidance value at 1e-7 of the others carries 1e-5 x its own size of summation-order noise: absolute floor at the weights' scale)
        fin = np.isfinite(ref)
        assert np.abs(got[fin] - ref[fin]).max() <= 2e-5, (name, norm)


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,w,oh,ow", [(2, 114, 152, 228, 304), (1, 152, 608, 304, 1216), (3, 40, 125, 79, 249)])
def test_head_kernel_vs_torch_conv_and_through_the_forward(B, h, w, oh, ow):
    """the reference's own sizes (NYU 114x152 -> 228x304, :318-319) and KITTI: against torch's conv_transpose2d + conv2d on the GPU (the reference's op
    sequence), and END TO END: head (gate_wb) -> forward with 'prenorm' == head (raw) -> forward with '8sum' == oracle on the raw guidance"""
    import cspn_amd
    import torch.nn.functional as F
    from cspn_amd.train_utils import guidance_heads
    gen = torch.Generator(device="cuda").manual_seed(B + h + w)
    C = 64
    x = torch.randn(B, C, h, w, generator=gen, device="cuda")
    w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / (3.0 * C ** 0.5)
    w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / (3.0 * C ** 0.5) + 0.02
    up = torch.zeros(C, 1, 2, 2, device="cuda")
    up[:, :, 0, 0] = 1
    U = F.conv_transpose2d(x, up, stride=2, groups=C)[:, :, :oh, :ow]
    rg = F.conv2d(U.double(), w6.double(), padding=1).float()
    rb = F.conv2d(U.double(), w5.double(), padding=1).float()
    guid, blur = guidance_heads(x, w6, w5, oh, ow)
    assert float((guid - rg).abs().max() / rg.abs().max()) <= 1e-5
    assert float((blur - rb).abs().max() / rb.abs().max()) <= 1e-5
    wb, blur2 = guidance_heads(x, w6, w5, oh, ow, norm_type="8sum")
    assert torch.equal(blur2, blur)
    ref_wb = cspn_amd.cspn2d_normalize(guid, "8sum")                      # the stand-alone normalisation of the engine (pinned to the reference's gate_wb)
    assert torch.equal(torch.isnan(wb), torch.isnan(ref_wb))
    assert float((wb - ref_wb).abs().nan_to_num().max()) <= 1e-5
    if ow % 4 == 0 and ow >= 256:
        depth = blur.abs() * 10 + 1.0
        a = cspn_amd.cspn2d_forward(wb, depth, None, 24, "prenorm")
        b_ = cspn_amd.cspn2d_forward(guid, depth, None, 24, "8sum")
        torch.cuda.synchronize()
        assert float((a - b_).abs().max() / b_.abs().max()) <= 1e-5
        ref = cspn2d_oracle(guid[:1].cpu(), depth[:1].cpu(), None, 24, "8sum")
        assert _rel(a[:1].cpu().numpy(), ref) <= 1e-4


@pytest.mark.gpu
def test_head_argument_checks():
    import cspn_amd
    from cspn_amd import _lib
    lib = cspn_amd.load()
    x = torch.zeros(1, 4, 3, 3, device="cuda")
    w6 = torch.zeros(8, 4, 3, 3, device="cuda")
    out = torch.zeros(1, 8, 6, 6, device="cuda")
    ws = torch.zeros(lib.cspn_guidance_head_workspace_bytes(4), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.cspn_guidance_head_f32(x.data_ptr(), w6.data_ptr(), None, out.data_ptr(), None, 1, 4, 3, 3, 7, 6, 2, ws.data_ptr(), ws.numel(), st) == -1   # H > 2 h
    assert lib.cspn_guidance_head_f32(x.data_ptr(), w6.data_ptr(), None, out.data_ptr(), None, 1, 4, 3, 3, 6, 6, _lib.NORM_TYPES["prenorm"], ws.data_ptr(), ws.numel(), st) == -1
    assert lib.cspn_guidance_head_f32(x.data_ptr(), w6.data_ptr(), w6.data_ptr(), out.data_ptr(), None, 1, 4, 3, 3, 6, 6, 2, ws.data_ptr(), ws.numel(), st) == -1   # w_blur without blur_out
    assert lib.cspn_guidance_head_f32(x.data_ptr(), w6.data_ptr(), None, out.data_ptr(), None, 1, 4, 3, 3, 6, 6, 2, ws.data_ptr(), 8, st) == -1          # workspace too small
    assert lib.cspn_guidance_head_f32(x.data_ptr(), w6.data_ptr(), None, out.data_ptr(), None, 1, 4, 3, 3, 6, 6, 2, ws.data_ptr(), ws.numel(), st) == 0


@pytest.mark.gpu
def test_head_fuzzed_shapes_vs_oracle():
    """60 seeded random shapes (1 .. 3 images, 1 .. 70 channels, heights 1 .. 13, widths 1 .. 150: below, at and above the 63-column segments and the
    two-row pairs; exact x2 and narrowed outputs incl. odd sizes): raw guidance + blur against the numpy oracle, gate_wb ('8sum' / '8sum_abs':
    consumer-sited stores + the in-place normalisation) against the engine's stand-alone normalisation of the oracle's raw guidance"""
    import cspn_amd
    from cspn_amd.train_utils import guidance_heads
    rng = np.random.default_rng(2026)
    for case in range(60):
        B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3, 8, 17, 64, 70]))
        h = int(rng.integers(1, 14))
        w = int(rng.choice([1, 2, 5, 31, 62, 63, 64, 65, 126, 127, 150]))
        oh, ow = 0, 0
        if rng.random() < 0.5:
            oh, ow = int(rng.integers(max(1, 2 * h - 3), 2 * h + 1)), int(rng.integers(max(1, 2 * w - 3), 2 * w + 1))
        gen = torch.Generator().manual_seed(case)
        x = torch.randn(B, C, h, w, generator=gen)
        w6 = torch.randn(8, C, 3, 3, generator=gen) / 3
        w5 = torch.randn(1, C, 3, 3, generator=gen) / 3
        rg, rb = guidance_head_oracle(x.numpy(), w6.numpy(), w5.numpy(), oh, ow)
        guid, blur = guidance_heads(x.cuda(), w6.cuda(), w5.cuda(), oh, ow)
        what = "case %d: B%d C%d h%d w%d -> %dx%d" % (case, B, C, h, w, oh, ow)
        assert guid.shape == rg.shape and _rel(guid.cpu().numpy(), rg) <= 1e-5 and _rel(blur.cpu().numpy(), rb) <= 1e-5, what
        for norm in ("8sum", "8sum_abs"):
            wb, blur2 = guidance_heads(x.cuda(), w6.cuda(), w5.cuda(), oh, ow, norm_type=norm)
            ref = cspn_amd.cspn2d_normalize(torch.from_numpy(rg).cuda(), norm)
            assert torch.equal(blur2, blur), what
            assert torch.equal(torch.isnan(wb), torch.isnan(ref)), what
            assert float((wb - ref).abs().nan_to_num().max()) <= 2e-5, what     # (weights are <= 1 in magnitude)


# ---------------------------------------------------------------------------------------------------------------- the heads' gradient (round 6)
def _grad_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_grad_golden.npz"))
    for n in sorted({k.split("/")[0] for k in z.files}):
        yield n, {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}


def test_head_backward_oracle_vs_reference_autograd_golden():
    """oracle/oracle.py guidance_head_backward_oracle against the gradients the unmodified reference layers' autograd produced
    (tests/golden/make_head_grad_golden.py)"""
    n = 0
    for name, c in _grad_golden():
        oh, ow = (int(v) for v in c["meta"])
        dx, d6, d5 = guidance_head_backward_oracle(c["x"], c["w6"], c["w5"], c["grad_guidance"], c["grad_blur"], oh, ow)
        assert _rel(dx, c["grad_x"]) <= 1e-5 and _rel(d6, c["grad_w6"]) <= 1e-5 and _rel(d5, c["grad_w5"]) <= 1e-5, name
        n += 1
    assert n == 6


def test_head_backward_oracle_vs_live_reference():
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip("reference tree not present")
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(2, 6, 5, 7, generator=gen)
    w6 = torch.randn(8, 6, 3, 3, generator=gen)
    w5 = torch.randn(1, 6, 3, 3, generator=gen)
    gg, gb = torch.randn(2, 8, 9, 13, generator=gen), torch.randn(2, 1, 9, 13, generator=gen)
    rdx, rd6, rd5 = ref_harness.reference_guidance_heads_grads(x, w6, w5, gg, gb, 9, 13)
    dx, d6, d5 = guidance_head_backward_oracle(x.numpy(), w6.numpy(), w5.numpy(), gg.numpy(), gb.numpy(), 9, 13)
    assert _rel(dx, rdx.numpy()) <= 1e-5 and _rel(d6, rd6.numpy()) <= 1e-5 and _rel(d5, rd5.numpy()) <= 1e-5


@pytest.mark.gpu
def test_head_backward_kernels_vs_reference_autograd_golden():
    from cspn_amd.train_utils import guidance_heads_backward
    for name, c in _grad_golden():
        t = {k: _dev(v) for k, v in c.items() if k != "meta"}
        dx, d6, d5 = guidance_heads_backward(t["x"], t["w6"], t["w5"], t["grad_guidance"], t["grad_blur"])
        torch.cuda.synchronize()
        assert _rel(dx.cpu().numpy(), c["grad_x"]) <= 1e-5, name
        assert _rel(d6.cpu().numpy(), c["grad_w6"]) <= 2e-5 and _rel(d5.cpu().numpy(), c["grad_w5"]) <= 2e-5, name
        # guidance head only; one of the two results only
        dx2, d62, none = guidance_heads_backward(t["x"], t["w6"], None, t["grad_guidance"], None)
        r = guidance_head_backward_oracle(c["x"], c["w6"], None, c["grad_guidance"], None, *(int(v) for v in c["meta"]))
        assert none is None and _rel(dx2.cpu().numpy(), r[0]) <= 1e-5 and _rel(d62.cpu().numpy(), r[1]) <= 2e-5, name
        dx3, a, b_ = guidance_heads_backward(t["x"], t["w6"], t["w5"], t["grad_guidance"], t["grad_blur"], need_w=False)
        assert a is None and b_ is None and torch.equal(dx3, dx)
        n_, d63, d53 = guidance_heads_backward(t["x"], t["w6"], t["w5"], t["grad_guidance"], t["grad_blur"], need_x=False)
        assert n_ is None and torch.equal(d63, d6) and torch.equal(d53, d5)       # (deterministic: the partial sums are added in a fixed order)


@pytest.mark.gpu
def test_head_backward_fuzzed_shapes_and_autograd():
    """40 seeded random shapes against the numpy oracle (channel counts around the matrix core's 32-column blocks, widths around the 8-pixel tiles and the
    63-column segments, narrowed and odd outputs), and the autograd Function: guidance_heads(...) -> loss.backward() == the same through torch's conv"""
    import torch.nn.functional as F
    from cspn_amd.train_utils import guidance_heads, guidance_heads_backward
    rng = np.random.default_rng(606)
    for case in range(40):
        B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3, 31, 32, 33, 64, 65, 96]))
        h = int(rng.integers(1, 12))
        w = int(rng.choice([1, 3, 7, 8, 9, 62, 63, 64, 100]))
        oh, ow = 0, 0
        if rng.random() < 0.5:
            oh, ow = int(rng.integers(max(1, 2 * h - 3), 2 * h + 1)), int(rng.integers(max(1, 2 * w - 3), 2 * w + 1))
        H, W = (oh, ow) if oh else (2 * h, 2 * w)
        gen = torch.Generator().manual_seed(1000 + case)
        x = torch.randn(B, C, h, w, generator=gen)
        w6 = torch.randn(8, C, 3, 3, generator=gen) / 3
        w5 = torch.randn(1, C, 3, 3, generator=gen) / 3
        gg, gb = torch.randn(B, 8, H, W, generator=gen), torch.randn(B, 1, H, W, generator=gen)
        rdx, rd6, rd5 = guidance_head_backward_oracle(x.numpy(), w6.numpy(), w5.numpy(), gg.numpy(), gb.numpy(), oh, ow)
        dx, d6, d5 = guidance_heads_backward(x.cuda(), w6.cuda(), w5.cuda(), gg.cuda(), gb.cuda())
        what = "case %d: B%d C%d h%d w%d -> %dx%d" % (case, B, C, h, w, H, W)
        assert _rel(dx.cpu().numpy(), rdx) <= 1e-5, what
        assert _rel(d6.cpu().numpy(), rd6) <= 2e-5 and _rel(d5.cpu().numpy(), rd5) <= 2e-5, what
    # through autograd, against torch's own conv on the unpooled map
    B, C, h, w, oh, ow = 2, 64, 9, 70, 17, 139
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, C, h, w, generator=gen, device="cuda")
    w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 24
    w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 24
    gg, gb = torch.randn(B, 8, oh, ow, generator=gen, device="cuda"), torch.randn(B, 1, oh, ow, generator=gen, device="cuda")
    xa, w6a, w5a = (t.clone().requires_grad_(True) for t in (x, w6, w5))
    g, b = guidance_heads(xa, w6a, w5a, oh, ow)
    ((g * gg).sum() + (b * gb).sum()).backward()
    xb, w6b, w5b = (t.double().clone().requires_grad_(True) for t in (x, w6, w5))
    up = torch.zeros(C, 1, 2, 2, device="cuda", dtype=torch.float64)
    up[:, :, 0, 0] = 1
    U = F.conv_transpose2d(xb, up, stride=2, groups=C)[:, :, :oh, :ow]
    ((F.conv2d(U, w6b, padding=1) * gg.double()).sum() + (F.conv2d(U, w5b, padding=1) * gb.double()).sum()).backward()
    for a, r in ((xa.grad, xb.grad), (w6a.grad, w6b.grad), (w5a.grad, w5b.grad)):
        assert float((a.double() - r).abs().max() / r.abs().max()) <= 2e-5
    # frozen weights: only dL/dx is computed
    xc = x.clone().requires_grad_(True)
    g, b = guidance_heads(xc, w6, w5, oh, ow)
    ((g * gg).sum() + (b * gb).sum()).backward()
    assert torch.equal(xc.grad, xa.grad)


@pytest.mark.gpu
def test_heads_plus_propagation_train_step_vs_torch_fp64():
    """the producer and the path as one differentiable pipeline: guidance_heads (raw) -> Affinity_Propagate (training mode: kept checkpoints) -> loss.backward();
    dL/dx, dL/dweight_guidance, dL/dweight_blur against torch autograd in float64 through the reference's op sequence (conv on the unpooled map + the plain-torch
    restatement of cspn.py:42-83)"""
    import torch.nn.functional as F
    import cspn_amd
    from cspn_amd.train_utils import guidance_heads
    sys.path.insert(0, ROOT)
    from tools.torch_path import cspn2d_torch
    B, C, h, w, N = 2, 16, 20, 140, 24
    gen = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn(B, C, h, w, generator=gen, device="cuda")
    w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 12
    w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 12 + 0.05
    sp = (torch.rand(B, 1, 2 * h, 2 * w, generator=gen, device="cuda") < 0.03).float() * 2.0
    go = torch.randn(B, 1, 2 * h, 2 * w, generator=gen, device="cuda")
    xa, w6a, w5a = (t.clone().requires_grad_(True) for t in (x, w6, w5))
    g, b = guidance_heads(xa, w6a, w5a)
    out = cspn_amd.Affinity_Propagate(N, 3, "8sum")(g, b, sp)
    (out * go).sum().backward()
    xb, w6b, w5b = (t.double().clone().requires_grad_(True) for t in (x, w6, w5))
    up = torch.zeros(C, 1, 2, 2, device="cuda", dtype=torch.float64)
    up[:, :, 0, 0] = 1
    U = F.conv_transpose2d(xb, up, stride=2, groups=C)
    ref = cspn2d_torch(F.conv2d(U, w6b, padding=1), F.conv2d(U, w5b, padding=1), sp.double(), N, "8sum")
    (ref * go.double()).sum().backward()
    assert float((out.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max()) <= 1e-5
    for a, r, what in ((xa.grad, xb.grad, "x"), (w6a.grad, w6b.grad, "w6"), (w5a.grad, w5b.grad, "w5")):
        assert float((a.double() - r).abs().max() / r.abs().max()) <= 2e-4, what


@pytest.mark.gpu
def test_head_backward_argument_checks():
    import cspn_amd
    lib = cspn_amd.load()
    x = torch.zeros(1, 4, 3, 3, device="cuda")
    w6, w5 = torch.zeros(8, 4, 3, 3, device="cuda"), torch.zeros(1, 4, 3, 3, device="cuda")
    gg, gb = torch.zeros(1, 8, 6, 6, device="cuda"), torch.zeros(1, 1, 6, 6, device="cuda")
    dx, d6, d5 = torch.empty_like(x), torch.empty_like(w6), torch.empty_like(w5)
    n = lib.cspn_guidance_head_backward_workspace_bytes(1, 4, 3, 3)
    assert n > 0 and lib.cspn_guidance_head_backward_workspace_bytes(0, 4, 3, 3) == 0
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda *a: lib.cspn_guidance_head_backward_f32(*a)   # noqa: E731
    P = lambda t: t.data_ptr()   # noqa: E731
    assert call(P(x), P(w6), P(w5), P(gg), P(gb), P(dx), P(d6), P(d5), 1, 4, 3, 3, 6, 6, P(ws), n, st) == 0
    assert call(P(x), P(w6), P(w5), P(gg), None, P(dx), P(d6), P(d5), 1, 4, 3, 3, 6, 6, P(ws), n, st) == -1        # a blur head without its gradient
    assert call(P(x), P(w6), None, P(gg), None, P(dx), P(d6), P(d5), 1, 4, 3, 3, 6, 6, P(ws), n, st) == -1         # grad_w_blur without a blur head
    assert call(P(x), P(w6), P(w5), P(gg), P(gb), P(dx), P(d6), P(d5), 1, 4, 3, 3, 7, 6, P(ws), n, st) == -1       # H > 2 h
    assert call(P(x), P(w6), P(w5), P(gg), P(gb), P(dx), P(d6), P(d5), 1, 4, 3, 3, 6, 6, P(ws), 64, st) == -2      # workspace too small
    assert call(None, P(w6), P(w5), P(gg), P(gb), P(dx), P(d6), P(d5), 1, 4, 3, 3, 6, 6, P(ws), n, st) == -1
    assert call(P(x), P(w6), P(w5), P(gg), P(gb), None, None, None, 1, 4, 3, 3, 6, 6, None, 0, st) == 0            # nothing asked for: nothing needed
    assert call(P(x), P(w6), P(w5), P(gg), P(gb), P(dx), P(d6), P(d5), 0, 4, 3, 3, 6, 6, None, 0, st) == 0         # empty batch
