"""Device-side mirrors of the small steps next to the propagation path in the reference training / evaluation loops
(SURVEY.md §8f-3, §8f-4), same names and call signatures:

    reference                                                   here
    utils.evaluate_error(gt_depth, pred_depth)   utils.py:19-47   evaluate_error(gt_depth, pred_depth) -> same dict, one
                                                                  fused masked reduction on the GPU, one 48-byte copy back
    loss.Wighted_L1_Loss()(pred, label)          loss.py:16-23    Wighted_L1_Loss()(pred, label) -> 0-d tensor, differentiable
    Unpool(num_channels, stride=2)(x)            torch_resnet_cspn_nyu.py:41-54   Unpool(num_channels, stride)(x), differentiable

    createSparseDepthImage(depth_image, n_sample)                 createSparseDepthImage(depth, n_sample, mode='nyu'|'kitti',
      nyu_dataset_loader.py:135-144, kitti_dataset_loader.py:138-148   seed): the Bernoulli mask drawn on the GPU, batched
    gud_up_proj_layer6(x), gud_up_proj_layer5(x)                  guidance_heads(x, layer6.conv1.weight, layer5.conv1.weight, oheight, owidth
      torch_resnet_cspn_nyu.py:187-206, :318-319, :372-373          [, norm_type]): both Simple_Gudi_UpConv_Block_Last_Layer heads (Unpool + 3x3 conv) as
                                                                  ONE kernel; with norm_type the guidance comes back as gate_wb (forward only)

The reference moves every prediction to the host before reducing it (train.py:204-206, eval.py:146-150)."""
import torch
import torch.nn as nn

from . import _lib
from .functional import _prep, _workspace

_KEYS = ['MSE', 'RMSE', 'ABS_REL', 'LG10', 'MAE', 'DELTA1.02', 'DELTA1.05', 'DELTA1.10', 'DELTA1.25', 'DELTA1.25^2',
         'DELTA1.25^3']


def _metrics(gt, pred):
    """-> device float32[12]: n_valid, then the 11 values of _KEYS"""
    lib = _lib.load()
    g = _prep(gt, "gt_depth")
    p = _prep(pred, "pred_depth", tuple(g.shape))
    out = torch.empty(12, dtype=torch.float32, device=g.device)
    n = g.numel()
    with torch.cuda.device(g.device):
        wsb = lib.cspn_metrics_workspace_bytes(n)
        ws = _workspace(wsb, g.device)
        rc = lib.cspn_metrics_f32(g.data_ptr(), p.data_ptr(), n, out.data_ptr(), ws.data_ptr(), wsb,
                                  torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn_metrics_f32")
    return out


def evaluate_error(gt_depth, pred_depth):
    """reference utils.py:19-47: dict of python floats (0 everywhere when no pixel has gt > 1e-4)"""
    v = _metrics(gt_depth, pred_depth).tolist()
    return {k: v[i + 1] for i, k in enumerate(_KEYS)}


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label):
        stats = _metrics(label, pred)
        ctx.save_for_backward(pred, label, stats)
        # MAE over label > 1e-4 == loss.py:18-22; nothing valid: loss.py:21-22 computes 0/0 = nan
        return torch.where(stats[0] > 0, stats[5], stats[5] + float("nan"))

    @staticmethod
    def backward(ctx, grad):
        pred, label, stats = ctx.saved_tensors
        lib = _lib.load()
        p, l = pred.contiguous(), label.contiguous()
        gp = torch.empty_like(p)
        gs = grad.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(p.device):
            rc = lib.cspn_l1_backward_f32(p.data_ptr(), l.data_ptr(), stats.data_ptr(), gs.data_ptr(), gp.data_ptr(), p.numel(),
                                          torch.cuda.current_stream(p.device).cuda_stream)
        _lib.check(rc, "cspn_l1_backward_f32")
        return gp.view_as(pred), None


class Wighted_L1_Loss(nn.Module):
    """reference loss.py:12-23 (spelling as there)"""

    def forward(self, pred, label):
        return _L1.apply(pred, label)


class _Unpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        lib = _lib.load()
        xc = _prep(x, "x")
        N, C, H, W = xc.shape
        out = torch.empty(N, C, H * stride, W * stride, dtype=torch.float32, device=xc.device)
        ctx.shape, ctx.stride = (N, C, H, W), stride
        with torch.cuda.device(xc.device):
            rc = lib.cspn_unpool_f32(xc.data_ptr(), out.data_ptr(), N * C, H, W, stride,
                                     torch.cuda.current_stream(xc.device).cuda_stream)
        _lib.check(rc, "cspn_unpool_f32")
        return out

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        N, C, H, W = ctx.shape
        g = go.contiguous()
        gx = torch.empty(N, C, H, W, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.cspn_unpool_backward_f32(g.data_ptr(), gx.data_ptr(), N * C, H, W, ctx.stride,
                                              torch.cuda.current_stream(g.device).cuda_stream)
        _lib.check(rc, "cspn_unpool_backward_f32")
        return gx, None


class Unpool(nn.Module):
    """reference torch_resnet_cspn_nyu.py:41-54: stride x stride unpooling with zero padding (no parameters)"""

    def __init__(self, num_channels, stride=2):
        super(Unpool, self).__init__()
        self.num_channels = num_channels
        self.stride = stride

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.num_channels:
            raise ValueError("expected [N,%d,H,W], got %s" % (self.num_channels, tuple(x.shape)))
        return _Unpool.apply(x, self.stride)


def createSparseDepthImage(depth_image, n_sample, mode="nyu", seed=0):
    """reference nyu_dataset_loader.py:135-144 (mode 'nyu': keep probability n_sample / n_pixels) and
    kitti_dataset_loader.py:138-148 (mode 'kitti': n_sample / n_valid_pixels, valid = depth > 1e-4), on the GPU:
    sparse_depth = depth_image * bernoulli(p), independently per pixel.  depth_image [..., H, W] on the device (any number
    of leading dims; every [H, W] slice is one image).  `seed` keys a counter-based generator (a fixed seed reproduces the
    mask; the reference draws from torch's global CPU generator)."""
    lib = _lib.load()
    d = _prep(depth_image, "depth_image")
    if d.dim() < 2:
        raise ValueError("depth_image must be [..., H, W]")
    hw = d.shape[-1] * d.shape[-2]
    n_images = d.numel() // hw if hw else 0
    out = torch.empty_like(d)
    if d.numel() == 0:
        return out
    m = {"nyu": 0, "kitti": 1}[mode]
    with torch.cuda.device(d.device):
        wsb = lib.cspn_sparse_sample_workspace_bytes(n_images)
        ws = _workspace(wsb, d.device)
        rc = lib.cspn_sparse_sample_f32(d.data_ptr(), out.data_ptr(), n_images, hw, int(n_sample), m, int(seed) & (2 ** 64 - 1),
                                        ws.data_ptr(), wsb, torch.cuda.current_stream(d.device).cuda_stream)
    _lib.check(rc, "cspn_sparse_sample_f32")
    return out


def _heads_forward(xx, wg, wb, H, W, norm):
    lib = _lib.load()
    B, C, h, w = xx.shape
    g = torch.empty(B, 8, H, W, dtype=torch.float32, device=xx.device)
    b = torch.empty(B, 1, H, W, dtype=torch.float32, device=xx.device) if wb is not None else None
    with torch.cuda.device(xx.device):
        wsb = lib.cspn_guidance_head_workspace_bytes(C)
        ws = _workspace(wsb, xx.device)
        rc = lib.cspn_guidance_head_f32(xx.data_ptr(), wg.data_ptr(), wb.data_ptr() if wb is not None else None, g.data_ptr(),
                                        b.data_ptr() if b is not None else None, B, C, h, w, H, W, norm, ws.data_ptr(), wsb,
                                        torch.cuda.current_stream(xx.device).cuda_stream)
    _lib.check(rc, "cspn_guidance_head_f32")
    return g, b


def guidance_heads_backward(x, weight_guidance, weight_blur, grad_guidance, grad_blur, need_x=True, need_w=True):
    """cspn_guidance_head_backward_f32: (dL/dx, dL/dweight_guidance, dL/dweight_blur) of the RAW heads -- what torch autograd computes through the two reference
    layers (torch_resnet_cspn_nyu.py:187-206) -- from dL/dguidance [B,8,H,W] and dL/dblur [B,1,H,W] (None without a blur head); skipped outputs are None."""
    lib = _lib.load()
    xx = _prep(x, "x")
    B, C, h, w = xx.shape
    wg = _prep(weight_guidance, "weight_guidance", (8, C, 3, 3))
    wb = _prep(weight_blur, "weight_blur", (1, C, 3, 3)) if weight_blur is not None else None
    H, W = int(grad_guidance.shape[2]), int(grad_guidance.shape[3])
    gg = _prep(grad_guidance, "grad_guidance", (B, 8, H, W))
    gb = _prep(grad_blur, "grad_blur", (B, 1, H, W)) if wb is not None else None
    dx = torch.empty_like(xx) if need_x else None
    dwg = torch.empty_like(wg) if need_w else None
    dwb = torch.empty_like(wb) if (need_w and wb is not None) else None
    with torch.cuda.device(xx.device):
        wsb = lib.cspn_guidance_head_backward_workspace_bytes(B, C, h, w)
        ws = _workspace(wsb, xx.device)
        rc = lib.cspn_guidance_head_backward_f32(xx.data_ptr(), wg.data_ptr(), wb.data_ptr() if wb is not None else None, gg.data_ptr(),
                                                 gb.data_ptr() if gb is not None else None, dx.data_ptr() if dx is not None else None,
                                                 dwg.data_ptr() if dwg is not None else None, dwb.data_ptr() if dwb is not None else None,
                                                 B, C, h, w, H, W, ws.data_ptr(), wsb, torch.cuda.current_stream(xx.device).cuda_stream)
    _lib.check(rc, "cspn_guidance_head_backward_f32")
    return dx, dwg, dwb


class _GuidanceHeadsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wg, wb, H, W):
        ctx.save_for_backward(x, wg, wb)
        g, b = _heads_forward(x, wg, wb, H, W, _lib.NORM_TYPES["none"])
        return g, b

    @staticmethod
    def backward(ctx, grad_g, grad_b):
        x, wg, wb = ctx.saved_tensors
        if grad_g is None:
            grad_g = torch.zeros(x.shape[0], 8, *((grad_b.shape[2:]) if grad_b is not None else (2 * x.shape[2], 2 * x.shape[3])), device=x.device)
        if wb is not None and grad_b is None:
            grad_b = torch.zeros(x.shape[0], 1, grad_g.shape[2], grad_g.shape[3], device=x.device)
        dx, dwg, dwb = guidance_heads_backward(x, wg, wb, grad_g.contiguous(), grad_b.contiguous() if grad_b is not None else None,
                                               need_x=ctx.needs_input_grad[0], need_w=ctx.needs_input_grad[1] or (wb is not None and ctx.needs_input_grad[2]))
        return dx, dwg if ctx.needs_input_grad[1] else None, dwb if (wb is not None and ctx.needs_input_grad[2]) else None, None, None


def guidance_heads(x, weight_guidance, weight_blur=None, oheight=0, owidth=0, norm_type=None):
    """The producer of the propagation's inputs: what the reference computes as
        guidance = self.gud_up_proj_layer6(x); x = self.gud_up_proj_layer5(x)          (torch_resnet_cspn_nyu.py:372-373)
    with both heads Simple_Gudi_UpConv_Block_Last_Layer (:187-206: Unpool + narrow to (oheight, owidth) + bias-free 3x3 conv), in ONE kernel that never
    multiplies the structurally zero taps.  x [B,C,h,w]; weight_guidance = layer6.conv1.weight [8,C,3,3]; weight_blur = layer5.conv1.weight [1,C,3,3] or None.
    norm_type None: -> (guidance [B,8,H,W], blur [B,1,H,W] | None), bit-compatible inputs of Affinity_Propagate(..., norm_type)(guidance, blur, sparse);
    differentiable w.r.t. x and both weights (cspn_guidance_head_backward_f32: the gradients torch autograd computes through the reference layers).
    norm_type '8sum' | '8sum_abs': the guidance comes back normalised -- gate_wb of affinity_normalization (cspn.py:85-144) -- for
    cspn2d_forward(gate_wb, blur, sparse, n_iter, 'prenorm') / cspn_amd.propagate_prenorm; these two modes are forward only (inference / frozen heads)."""
    xx = _prep(x, "x")
    B, C, h, w = xx.shape
    wg = _prep(weight_guidance, "weight_guidance", (8, C, 3, 3))
    wb = _prep(weight_blur, "weight_blur", (1, C, 3, 3)) if weight_blur is not None else None
    H, W = (int(oheight), int(owidth)) if (oheight and owidth) else (2 * h, 2 * w)
    if norm_type not in (None, "8sum", "8sum_abs"):
        raise ValueError("norm_type must be None (raw guidance), '8sum' or '8sum_abs' (gate_wb)")
    if norm_type is None and torch.is_grad_enabled() and (xx.requires_grad or wg.requires_grad or (wb is not None and wb.requires_grad)):
        return _GuidanceHeadsFunction.apply(xx, wg, wb, H, W)
    return _heads_forward(xx, wg, wb, H, W, _lib.NORM_TYPES["none" if norm_type is None else norm_type])
