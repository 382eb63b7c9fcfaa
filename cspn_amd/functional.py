"""Functional front-end over the C ABI: tensors in, tensor out, work enqueued on
torch's current HIP stream.  PyTorch is plumbing here (device memory + streams);
all arithmetic happens in libcspn_amd.so."""
import torch

from . import _lib


def _prep(t, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.CspnError(
            "cspn_amd: %s is on %s; the engine is GPU-only (hand-written HIP for gfx950) and has no CPU path"
            % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.contiguous()


def _workspace(nbytes, device):
    # torch's caching allocator: stream-ordered reuse is safe, base is >=512-B aligned
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


def cspn2d_forward(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum", algo="auto"):
    """All n_iter steps of reference cspn_pytorch/models/cspn.py:42-83 in the HIP engine.

    guidance [B,8,H,W], blur_depth [B,1,H,W], sparse_depth [B,1,H,W] or None -> [B,1,H,W]."""
    lib = _lib.load()
    if guidance.dim() != 4 or guidance.shape[1] != 8:
        raise ValueError("guidance must be [B,8,H,W], got %s" % (tuple(guidance.shape),))
    B, _, H, W = guidance.shape
    g = _prep(guidance, "guidance")
    h = _prep(blur_depth, "blur_depth", (B, 1, H, W))
    s = _prep(sparse_depth, "sparse_depth", (B, 1, H, W)) if sparse_depth is not None else None
    if h.device != g.device or (s is not None and s.device != g.device):
        raise ValueError("all tensors must live on the same device")
    out = torch.empty_like(h)
    if B == 0:
        return out
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn2d_workspace_bytes(B, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        rc = lib.cspn2d_forward_f32_algo(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None,
                                         out.data_ptr(), B, H, W, int(n_iter), _lib.NORM_TYPES[norm_type],
                                         _lib.ALGOS[algo], ws.data_ptr(), ws_bytes, stream)
    _lib.check(rc, "cspn2d_forward_f32")
    return out


def cspn2d_normalize(guidance, norm_type="8sum"):
    """reference affinity_normalization (cspn_pytorch/models/cspn.py:85-144) as a stand-alone HIP kernel: guidance [B,8,H,W] ->
    gate_wb [B,8,H,W] (normalised, consumer-sited): what a producer head with a fused epilogue would emit, and what
    cspn2d_forward(..., norm_type="prenorm") takes in place of the raw guidance."""
    lib = _lib.load()
    if guidance.dim() != 4 or guidance.shape[1] != 8:
        raise ValueError("guidance must be [B,8,H,W], got %s" % (tuple(guidance.shape),))
    g = _prep(guidance, "guidance")
    B, _, H, W = g.shape
    out = torch.empty_like(g)
    if B == 0:
        return out
    with torch.cuda.device(g.device):
        rc = lib.cspn2d_normalize_f32(g.data_ptr(), out.data_ptr(), B, H, W, _lib.NORM_TYPES[norm_type],
                                      torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn2d_normalize_f32")
    return out


def guidance_to_sited8(guidance, norm_type="8sum"):
    """(closed experiment, experiment builds only: libcspn_amd_hooks.so) [B,8,H,W] -> the pre-sited pair-interleaved layout
    [B,H,W/2,8,2] of DESIGN.md 3.6"""
    hooks = _lib.load_hooks()
    g = _prep(guidance, "guidance")
    B, _, H, W = g.shape
    out = torch.empty(B, H, W // 2, 8, 2, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = hooks.cspn_debug_guidance_to_sited8(g.data_ptr(), out.data_ptr(), B, H, W, _lib.NORM_TYPES[norm_type],
                                                 torch.cuda.current_stream(g.device).cuda_stream)
    if rc != 0:
        raise _lib.CspnError("cspn_debug_guidance_to_sited8 failed (code %d): the sited8 experiment is only in experiment builds "
                             "(make -C cspn_amd/csrc EXPERIMENTS=1)" % rc)
    return out


def cspn2d_forward_sited8(guidance_s8, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum"):
    """(closed experiment, experiment builds only) cspn2d_forward with the guidance in the sited8 layout (24 iterations, W >= 256)."""
    hooks = _lib.load_hooks()
    B, H, W2 = guidance_s8.shape[:3]
    W = 2 * W2
    g = _prep(guidance_s8, "guidance_s8", (B, H, W2, 8, 2))
    h = _prep(blur_depth, "blur_depth", (B, 1, H, W))
    s = _prep(sparse_depth, "sparse_depth", (B, 1, H, W)) if sparse_depth is not None else None
    out = torch.empty_like(h)
    with torch.cuda.device(g.device):
        rc = hooks.cspn_debug_forward_sited8(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, out.data_ptr(),
                                             B, H, W, int(n_iter), _lib.NORM_TYPES[norm_type],
                                             torch.cuda.current_stream(g.device).cuda_stream)
    if rc != 0:
        raise _lib.CspnError("cspn_debug_forward_sited8 failed (code %d): experiment builds only, passes of exactly 24 iterations, "
                             "W >= 256, W %% 4 == 0" % rc)
    return out


def cspn2d_backward(guidance, blur_depth, sparse_depth, grad_out, n_iter=24, norm_type="8sum",
                    need_guidance=True, need_blur=True):
    """Gradient of cspn2d_forward w.r.t. guidance and blur_depth (what autograd computes through reference
    cspn_pytorch/models/cspn.py:42-83, back-propagated by reference train.py:196-198) in the HIP engine.
    -> (grad_guidance [B,8,H,W] or None, grad_blur [B,1,H,W] or None)"""
    lib = _lib.load()
    B, _, H, W = guidance.shape
    g = _prep(guidance, "guidance", (B, 8, H, W))
    h = _prep(blur_depth, "blur_depth", (B, 1, H, W))
    s = _prep(sparse_depth, "sparse_depth", (B, 1, H, W)) if sparse_depth is not None else None
    go = _prep(grad_out, "grad_out", (B, 1, H, W))
    gg = torch.empty_like(g) if need_guidance else None
    gh = torch.empty_like(h) if need_blur else None
    if B == 0 or not (need_guidance or need_blur):
        return gg, gh
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn2d_backward_workspace_bytes(B, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        rc = lib.cspn2d_backward_f32(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, go.data_ptr(),
                                     gg.data_ptr() if gg is not None else None, gh.data_ptr() if gh is not None else None,
                                     B, H, W, int(n_iter), _lib.NORM_TYPES[norm_type], ws.data_ptr(), ws_bytes, stream)
    _lib.check(rc, "cspn2d_backward_f32")
    return gg, gh


def cspn2d_history_bytes(B, H, W, n_iter):
    """bytes of what the training-mode forward keeps for its backward: every fourth level + the folded coefficients (0: not available for this shape)"""
    return int(_lib.load().cspn2d_history_bytes(int(B), int(H), int(W), int(n_iter)))


def cspn2d_forward_with_history(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum"):
    """Training-mode forward: same output as cspn2d_forward, plus an opaque `history` tensor for cspn2d_backward_from_history:
    the checkpoints H_4, H_8 .. H_20 (every fourth level, register order per 4-column group) followed by the 8 folded coefficient
    planes -- 13 planes of B*H*W floats (the backward recomputes the levels in between; a tensor in the round-2 format, all 23
    levels, is NOT accepted: its size differs and the size is checked).  Only where cspn2d_history_bytes(...) > 0."""
    lib = _lib.load()
    B, _, H, W = guidance.shape
    g = _prep(guidance, "guidance", (B, 8, H, W))
    h = _prep(blur_depth, "blur_depth", (B, 1, H, W))
    s = _prep(sparse_depth, "sparse_depth", (B, 1, H, W)) if sparse_depth is not None else None
    out = torch.empty_like(h)
    with torch.cuda.device(g.device):
        hb = lib.cspn2d_history_bytes(B, H, W, int(n_iter))
        if hb == 0:
            raise _lib.CspnError("cspn_amd: no history mode for shape %s, n_iter %d" % (tuple(guidance.shape), n_iter))
        hist = torch.empty(hb, dtype=torch.uint8, device=g.device)
        ws_bytes = lib.cspn2d_workspace_bytes(B, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        rc = lib.cspn2d_forward_history_f32(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, out.data_ptr(),
                                            hist.data_ptr(), hb, B, H, W, int(n_iter), _lib.NORM_TYPES[norm_type],
                                            ws.data_ptr(), ws_bytes, torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn2d_forward_history_f32")
    return out, hist


def cspn2d_backward_from_history(guidance, blur_depth, sparse_depth, grad_out, history, n_iter=24, norm_type="8sum",
                                 need_guidance=True, need_blur=True):
    """Gradients as cspn2d_backward, starting from the history a training-mode forward kept."""
    lib = _lib.load()
    B, _, H, W = guidance.shape
    g = _prep(guidance, "guidance", (B, 8, H, W))
    h = _prep(blur_depth, "blur_depth", (B, 1, H, W))
    s = _prep(sparse_depth, "sparse_depth", (B, 1, H, W)) if sparse_depth is not None else None
    go = _prep(grad_out, "grad_out", (B, 1, H, W))
    gg = torch.empty_like(g) if need_guidance else None
    gh = torch.empty_like(h) if need_blur else None
    if not (need_guidance or need_blur):
        return gg, gh
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn2d_backward_history_workspace_bytes(B, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        rc = lib.cspn2d_backward_history_f32(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, go.data_ptr(),
                                             history.data_ptr(), history.numel(), gg.data_ptr() if gg is not None else None,
                                             gh.data_ptr() if gh is not None else None, B, H, W, int(n_iter),
                                             _lib.NORM_TYPES[norm_type], ws.data_ptr(), ws_bytes,
                                             torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn2d_backward_history_f32")
    return gg, gh


def cspn3d_forward(gate, feat, sparse=None, n_iter=12, norm_type="8sum_abs", algo="auto", _return_ws=False):
    """gate [B,26,D,H,W], feat [B,1,D,H,W] -> [B,1,D,H,W]; n_iter 3x3x3 propagation steps.  algo: 'auto' | 'stepwise'
    (one launch per step) | 'persistent' (gates resident in registers across the steps; norm_type 'none' without a mask)."""
    lib = _lib.load()
    if gate.dim() != 5 or gate.shape[1] != 26:
        raise ValueError("gate must be [B,26,D,H,W], got %s" % (tuple(gate.shape),))
    B, _, D, H, W = gate.shape
    g = _prep(gate, "gate")
    h = _prep(feat, "feat", (B, 1, D, H, W))
    s = _prep(sparse, "sparse", (B, 1, D, H, W)) if sparse is not None else None
    out = torch.empty_like(h)
    if B == 0:
        return out
    with torch.cuda.device(g.device):
        if any(t.data_ptr() % 16 for t in (g, h, out)):   # misaligned views take the folding path: the full workspace
            ws_bytes = lib.cspn3d_workspace_bytes(B, D, H, W, int(n_iter))
        else:
            ws_bytes = lib.cspn3d_workspace_bytes_ex(B, D, H, W, int(n_iter), _lib.NORM_TYPES[norm_type], int(s is not None))
        ws = _workspace(ws_bytes, g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        rc = lib.cspn3d_forward_f32_algo(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None,
                                         out.data_ptr(), B, D, H, W, int(n_iter), _lib.NORM_TYPES[norm_type],
                                         _lib.ALGOS_3D[algo], ws.data_ptr(), ws_bytes, stream)
    _lib.check(rc, "cspn3d_forward_f32")
    return (out, ws) if _return_ws else out


def cspn3d_forward_multi(gate, feat, n_iter=12):
    """gate [B,26,D,H,W] (used as given: the Paddle contract), feat [B,C,D,H,W] -> [B,C,D,H,W]: the C channels share the gates
    (reference cspn_paddle/README.md:56), which are read once per forward and stay in the registers while the n_iter steps run for
    one channel after the other.  Raises CspnError where the persistent kernel does not take the call (see cspn3d_multi_supported)."""
    lib = _lib.load()
    if gate.dim() != 5 or gate.shape[1] != 26:
        raise ValueError("gate must be [B,26,D,H,W], got %s" % (tuple(gate.shape),))
    B, _, D, H, W = gate.shape
    C = feat.shape[1]
    g = _prep(gate, "gate")
    h = _prep(feat, "feat", (B, C, D, H, W))
    if h.device != g.device:
        raise ValueError("all tensors must live on the same device")
    out = torch.empty_like(h)
    if B == 0:
        return out
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn3d_workspace_bytes_ex(B, D, H, W, int(n_iter), _lib.NORM_TYPES["none"], 0)
        ws = _workspace(ws_bytes, g.device)
        rc = lib.cspn3d_forward_multi_f32(g.data_ptr(), h.data_ptr(), out.data_ptr(), B, C, D, H, W, int(n_iter), ws.data_ptr(), ws_bytes,
                                          torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn3d_forward_multi_f32")
    return out


def cspn3d_check_status(device=None):
    """Synchronises the current stream of `device` and raises CspnError if a persistent 3D launch gave up on it (its outputs are
    NaN-filled): the failure the C ABI can only report after the call has returned.  Every later cspn3d_* call raises it too
    (once) without a synchronisation; call this where a result is about to be trusted without another 3D call in between."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        rc = lib.cspn3d_check_status(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "cspn3d_check_status")


def cspn3d_backward(gate, feat, grad_out, n_iter=1, need_gate=True, need_feat=True):
    """Gradient of cspn3d_forward(gate, feat, None, n_iter, 'none') -- the Paddle contract, the op the reference demo's
    optimiser differentiates (cspn_paddle/demo.py:65-75) -- w.r.t. gate and feat, in the HIP engine.
    -> (grad_gate [B,26,D,H,W] or None, grad_feat [B,1,D,H,W] or None)"""
    lib = _lib.load()
    if gate.dim() != 5 or gate.shape[1] != 26:
        raise ValueError("gate must be [B,26,D,H,W], got %s" % (tuple(gate.shape),))
    B, _, D, H, W = gate.shape
    g = _prep(gate, "gate")
    h = _prep(feat, "feat", (B, 1, D, H, W))
    go = _prep(grad_out, "grad_out", (B, 1, D, H, W))
    gg = torch.empty_like(g) if need_gate else None
    gf = torch.empty_like(h) if need_feat else None
    if B == 0 or not (need_gate or need_feat):
        return gg, gf
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn3d_backward_workspace_bytes(B, D, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        rc = lib.cspn3d_backward_f32(g.data_ptr(), h.data_ptr(), go.data_ptr(), gg.data_ptr() if gg is not None else None,
                                     gf.data_ptr() if gf is not None else None, B, D, H, W, int(n_iter),
                                     _lib.NORM_TYPES["none"], ws.data_ptr(), ws_bytes, stream)
    _lib.check(rc, "cspn3d_backward_f32")
    return gg, gf


def cspn3d_backward_multi(gate, feat, grad_out, n_iter=1, need_gate=True, need_feat=True):
    """Gradient of the n_iter-step 3D propagation of C channels on SHARED gates (feat, grad_out [B,C,D,H,W]; reference
    cspn_paddle/README.md:56, differentiated at demo.py:65-75) -> (grad_gate [B,26,D,H,W] summed over the channels or None,
    grad_feat [B,C,D,H,W] or None); one call of the HIP engine (cspn3d_backward_multi_f32)."""
    lib = _lib.load()
    if gate.dim() != 5 or gate.shape[1] != 26:
        raise ValueError("gate must be [B,26,D,H,W], got %s" % (tuple(gate.shape),))
    B, _, D, H, W = gate.shape
    C = feat.shape[1]
    g = _prep(gate, "gate")
    h = _prep(feat, "feat", (B, C, D, H, W))
    go = _prep(grad_out, "grad_out", (B, C, D, H, W))
    if h.device != g.device or go.device != g.device:
        raise ValueError("all tensors must live on the same device")
    gg = torch.empty_like(g) if need_gate else None
    gf = torch.empty_like(h) if need_feat else None
    if B == 0 or not (need_gate or need_feat):
        return gg, gf
    with torch.cuda.device(g.device):
        ws_bytes = lib.cspn3d_backward_multi_workspace_bytes(B, C, D, H, W, int(n_iter))
        ws = _workspace(ws_bytes, g.device)
        rc = lib.cspn3d_backward_multi_f32(g.data_ptr(), h.data_ptr(), go.data_ptr(), gg.data_ptr() if gg is not None else None,
                                           gf.data_ptr() if gf is not None else None, B, C, D, H, W, int(n_iter),
                                           ws.data_ptr(), ws_bytes, torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(rc, "cspn3d_backward_multi_f32")
    return gg, gf


class _AffinityPropagateMultiFunction(torch.autograd.Function):
    """3D, C > 1 input channels on shared gates: forward and backward are one engine call each for all channels"""

    @staticmethod
    def forward(ctx, x, gate_weight, n_iter):
        ctx.n_iter = int(n_iter)
        ctx.save_for_backward(x, gate_weight)
        if _lib.load().cspn3d_multi_supported(x.shape[0], x.shape[1], *x.shape[2:], int(n_iter)) and x.data_ptr() % 16 == 0 \
                and gate_weight.data_ptr() % 16 == 0:
            return cspn3d_forward_multi(gate_weight, x, n_iter)
        return torch.cat([cspn3d_forward(gate_weight, x[:, c:c + 1].contiguous(), None, n_iter, "none") for c in range(x.shape[1])], 1)

    @staticmethod
    def backward(ctx, grad_out):
        x, gate_weight = ctx.saved_tensors
        gg, gx = cspn3d_backward_multi(gate_weight, x, grad_out, ctx.n_iter, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return gx, gg, None


class _AffinityPropagateFunction(torch.autograd.Function):
    """n_iter chained propagation steps with the same gates, one input channel; differentiable w.r.t. both arguments."""

    @staticmethod
    def forward(ctx, x, gate_weight, n_iter):
        ctx.n_iter = int(n_iter)
        ctx.save_for_backward(x, gate_weight)
        if x.dim() == 4:
            return cspn2d_forward(gate_weight, x, None, n_iter, "none")
        return cspn3d_forward(gate_weight, x, None, n_iter, "none")

    @staticmethod
    def backward(ctx, grad_out):
        x, gate_weight = ctx.saved_tensors
        need_x, need_g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if x.dim() == 4:
            gg, gx = cspn2d_backward(gate_weight, x, None, grad_out, ctx.n_iter, "none", need_g, need_x)
        else:
            gg, gx = cspn3d_backward(gate_weight, x, grad_out, ctx.n_iter, need_g, need_x)
        return gx, gg, None


def affinity_propagate(input, gate_weight, kernel_size=3, n_iter=1):
    """Mirror of fluid.layers.affinity_propagate (reference cspn_paddle/demo.py:41-43,50-52;
    contract cspn_paddle/README.md:54-56): input [N,C,...], gate_weight [N,3**d-1,...] already
    normalised over the channel dim by the caller, shared across the C input channels.
    d = 2 or 3.  n_iter > 1 fuses that many chained calls (demo.py:39,50).  Differentiable w.r.t. input and
    gate_weight like the reference op (the demo trains through it, demo.py:65-75)."""
    if kernel_size != 3:
        raise ValueError("only kernel_size == 3 is supported (reference cspn_paddle/demo.py:90)")
    d = input.dim() - 2
    if d not in (2, 3):
        raise ValueError("input must be [N,C,H,W] or [N,C,D,H,W]")
    if gate_weight.shape[1] != 3 ** d - 1:
        raise ValueError("gate_weight must have %d channels" % (3 ** d - 1))
    N, C = input.shape[:2]
    needs_grad = torch.is_grad_enabled() and (input.requires_grad or gate_weight.requires_grad)
    if input.device != gate_weight.device:
        raise ValueError("all tensors must live on the same device")
    if d == 3 and C > 1 and not needs_grad and input.is_cuda and _lib.load().cspn3d_multi_supported(N, C, *input.shape[2:], int(n_iter)) \
            and input.is_contiguous() and gate_weight.is_contiguous() and input.data_ptr() % 16 == 0 and gate_weight.data_ptr() % 16 == 0:
        return cspn3d_forward_multi(gate_weight, input, n_iter)   # the gates are read once for all C channels
    if d == 3 and C > 1 and needs_grad and input.is_cuda:
        # training through C channels on shared gates (demo.py:65-75): one forward and one backward call for all of them
        return _AffinityPropagateMultiFunction.apply(input.contiguous(), gate_weight.contiguous(), n_iter)
    outs = []
    for c in range(C):  # gates shared across channels (README.md:56)
        x = input[:, c:c + 1].contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or gate_weight.requires_grad):
            outs.append(_AffinityPropagateFunction.apply(x, gate_weight, n_iter))
        elif d == 2:
            outs.append(cspn2d_forward(gate_weight, x, None, n_iter, "none"))
        else:
            outs.append(cspn3d_forward(gate_weight, x, None, n_iter, "none"))
    return outs[0] if C == 1 else torch.cat(outs, 1)
