"""Drop-in for reference cspn_pytorch/models/cspn.py: same class name, same
constructor, same forward call -- the arithmetic runs in hand-written HIP kernels
(libcspn_amd.so) instead of ZeroPad2d + cat + Conv3d.

    reference                                   here
    Affinity_Propagate(prop_time, prop_kernel,  identical signature (cspn.py:16-19)
                       norm_type='8sum')
    forward(guidance, blur_depth,               identical, plus an optional n_iter that
            sparse_depth=None)                  overrides prop_time (BASELINE north_star)

Differences a caller can observe, all deliberate:
  * no `sum_conv` sub-module ever appears in state_dict() (the reference registers one
    during its first forward, cspn.py:44-53; checkpoints are key-filtered on load,
    update_model.py:16-23, so both directions keep working);
  * inputs must already be on the GPU (the reference calls .cuda() itself, cspn.py:50);
  * differentiable w.r.t. guidance and blur_depth (HIP backward kernels, cspn_amd/csrc/cspn2d_backward.hip: the gradient
    torch autograd computes through the reference forward, which reference train.py:196-198 back-propagates through);
    sparse_depth gets no gradient (only its sign is used, cspn.py:64)."""
import torch
import torch.nn as nn

from . import functional as F


class _CSPN2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, guidance, blur_depth, sparse_depth, n_iter, norm_type, algo, keep_history):
        ctx.n_iter, ctx.norm_type = n_iter, norm_type
        needs_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        B, _, H, W = guidance.shape
        if (keep_history and needs_grad and algo in ("auto", "fused") and guidance.is_cuda
                and F.cspn2d_history_bytes(B, H, W, n_iter) > 0):
            # training: the forward keeps every FOURTH intermediate level (H_4 .. H_20) and the folded coefficients -- 13 planes,
            # where autograd keeps ~27 temporaries per iteration for the reference --, the backward starts from them and
            # recomputes the three levels in between
            out, hist = F.cspn2d_forward_with_history(guidance, blur_depth, sparse_depth, n_iter, norm_type)
            ctx.save_for_backward(guidance, blur_depth, sparse_depth, hist)
            return out
        ctx.save_for_backward(guidance, blur_depth, sparse_depth, None)
        return F.cspn2d_forward(guidance, blur_depth, sparse_depth, n_iter, norm_type, algo)

    @staticmethod
    def backward(ctx, grad_out):
        guidance, blur_depth, sparse_depth, hist = ctx.saved_tensors
        if hist is not None:
            gg, gh = F.cspn2d_backward_from_history(guidance, blur_depth, sparse_depth, grad_out, hist, ctx.n_iter, ctx.norm_type,
                                                    need_guidance=ctx.needs_input_grad[0], need_blur=ctx.needs_input_grad[1])
        else:
            gg, gh = F.cspn2d_backward(guidance, blur_depth, sparse_depth, grad_out, ctx.n_iter, ctx.norm_type,
                                       need_guidance=ctx.needs_input_grad[0], need_blur=ctx.needs_input_grad[1])
        return gg, gh, None, None, None, None, None


def propagate_prenorm(gate_wb, blur_depth, sparse_depth=None, n_iter=24, algo="auto", keep_history=True):
    """The loop of reference cspn.py:66-81 started from the tensor its affinity_normalization returns (gate_wb [B,8,H,W], cropped to the
    image: what cspn_amd.cspn2d_normalize or the guidance head with norm_type='8sum' emit) -- the pre-normalised input contract
    (CSPN_NORM_PRENORM), differentiable: the gradients w.r.t. gate_wb and blur_depth are what torch autograd computes through the
    reference forward for those two tensors (tests/golden/cspn2d_grad_prenorm_golden.npz); chaining dL/dgate_wb into whatever produced it
    is the caller's autograd."""
    if n_iter == 0:
        return blur_depth
    return _CSPN2dFunction.apply(gate_wb, blur_depth, sparse_depth, int(n_iter), "prenorm", algo, keep_history)


class Affinity_Propagate(nn.Module):

    def __init__(self, prop_time, prop_kernel, norm_type='8sum'):
        super(Affinity_Propagate, self).__init__()
        self.prop_time = prop_time
        self.prop_kernel = prop_kernel
        assert prop_kernel == 3, 'this version only support 8 (3x3 - 1) neighborhood'  # cspn.py:33
        self.norm_type = norm_type
        assert norm_type in ['8sum', '8sum_abs']  # cspn.py:36
        self.in_feature = 1
        self.out_feature = 1
        self.algo = "auto"
        self.keep_history = True   # training: keep the forward's checkpoints (every fourth level + folded coefficients) for the backward

    def forward(self, guidance, blur_depth, sparse_depth=None, n_iter=None):
        n = self.prop_time if n_iter is None else int(n_iter)
        if '8sum' not in self.norm_type:  # cspn.py:75-78
            raise ValueError('unknown norm %s' % self.norm_type)
        if n == 0:
            return blur_depth  # cspn.py:61,66,83: the very same tensor object
        return _CSPN2dFunction.apply(guidance, blur_depth, sparse_depth, n, self.norm_type, self.algo, self.keep_history)

    def extra_repr(self):
        return "prop_time=%d, prop_kernel=%d, norm_type=%r" % (self.prop_time, self.prop_kernel, self.norm_type)
