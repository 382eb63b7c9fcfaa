"""Runs the UNMODIFIED reference module on CPU tensors.  Authoring-container only:
/root/reference does not exist on the GPU box, so nothing that runs there may
import this file.  Used by tests/golden/make_golden.py (fixture generation) and
by tests/test_oracle.py::test_oracle_vs_live_reference (skipped when absent).

The reference hard-codes `.cuda()` at cspn.py:50; we replace torch.Tensor.cuda
with identity *around the call* (the reference file itself is untouched)."""
import contextlib
import importlib.util
import os

import torch

REF_CSPN = "/root/reference/cspn_pytorch/models/cspn.py"


def available():
    return os.path.exists(REF_CSPN)


def load_reference_module():
    spec = importlib.util.spec_from_file_location("_reference_cspn", REF_CSPN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def cuda_is_identity():
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def reference_forward(guidance, blur_depth, sparse_depth=None, n_iter=24, norm_type="8sum"):
    """-> torch.float32 [B,1,H,W] computed by /root/reference/cspn_pytorch/models/cspn.py:42-83."""
    ref = load_reference_module()
    m = ref.Affinity_Propagate(n_iter, 3, norm_type)
    with torch.no_grad(), cuda_is_identity():
        return m(guidance, blur_depth, sparse_depth)


def reference_grads(guidance, blur_depth, sparse_depth, grad_out, n_iter=24, norm_type="8sum"):
    """-> (out, dL/dguidance, dL/dblur_depth) of L = sum(out * grad_out), by torch autograd through the UNMODIFIED
    reference forward (what reference cspn_pytorch/train.py:196-198 back-propagates through)."""
    ref = load_reference_module()
    m = ref.Affinity_Propagate(n_iter, 3, norm_type)
    g = guidance.clone().requires_grad_(True)
    h = blur_depth.clone().requires_grad_(True)
    with cuda_is_identity():
        out = m(g, h, sparse_depth)
        gg, gh = torch.autograd.grad(out, [g, h], grad_out)
    return out.detach(), gg, gh


def reference_grads_wrt_gate_wb(guidance, blur_depth, sparse_depth, grad_out, n_iter=24, norm_type="8sum"):
    """-> (gate_wb [B,8,H,W], out, dL/dgate_wb [B,8,H,W], dL/dblur_depth) of L = sum(out * grad_out) by torch autograd through the UNMODIFIED
    reference forward, differentiated with respect to the tensor its affinity_normalization RETURNS (cspn.py:85-144; also the source of
    gate_sum, :139): the method is wrapped on the instance so that the returned gate_wb keeps its gradient -- the reference file is
    untouched.  Cropped to the image like the reference crops the product (cspn.py:72): nothing outside it reaches the output."""
    ref = load_reference_module()
    m = ref.Affinity_Propagate(n_iter, 3, norm_type)
    g = guidance.clone().requires_grad_(True)
    h = blur_depth.clone().requires_grad_(True)
    orig, cap = m.affinity_normalization, {}

    def keeping(gd):
        wb, gs = orig(gd)
        wb.retain_grad()
        cap["wb"] = wb
        return wb, gs
    m.affinity_normalization = keeping
    with cuda_is_identity():
        out = m(g, h, sparse_depth)
        out.backward(grad_out)
    wb = cap["wb"]
    return (wb.detach()[:, :, 0, 1:-1, 1:-1].contiguous(), out.detach(), wb.grad[:, :, 0, 1:-1, 1:-1].contiguous(), h.grad.detach())


def reference_gate_wb(guidance, norm_type="8sum"):
    """-> (gate_wb [B,8,H,W], gate_sum [B,1,H,W]) of /root/reference/cspn_pytorch/models/cspn.py:85-144 (affinity_normalization),
    gate_wb cropped to the image like the reference crops the product at cspn.py:72: the consumer-sited, normalised weights the
    loop of cspn.py:66-81 multiplies with.  The module's sum_conv only exists after a forward (cspn.py:42-51): one is run first."""
    ref = load_reference_module()
    m = ref.Affinity_Propagate(1, 3, norm_type)
    B, _, H, W = guidance.shape
    with torch.no_grad(), cuda_is_identity():
        m(guidance, torch.zeros(B, 1, H, W), None)           # creates m.sum_conv exactly as the reference does
        wb, gs = m.affinity_normalization(guidance)
    return wb[:, :, 0, 1:-1, 1:-1].contiguous(), gs.contiguous()


REF_MODEL = "/root/reference/cspn_pytorch/models/torch_resnet_cspn_nyu.py"


def reference_guidance_heads(x, w_guidance, w_blur, oheight=0, owidth=0):
    """-> (guidance [B,8,H,W], blur [B,1,H,W]) computed by the UNMODIFIED reference classes Simple_Gudi_UpConv_Block_Last_Layer
    (/root/reference/cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206; Unpool :41-54), instantiated as the backbone does (:318-319)
    with the given conv weights.  The file does `import cspn as post_process` (:12): its directory goes on sys.path for the import."""
    import sys
    d = os.path.dirname(REF_MODEL)
    sys.path.insert(0, d)
    try:
        with cuda_is_identity():
            spec = importlib.util.spec_from_file_location("_reference_resnet_cspn", REF_MODEL)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            C = x.shape[1]
            l6 = mod.Simple_Gudi_UpConv_Block_Last_Layer(C, 8, oheight, owidth)
            l5 = mod.Simple_Gudi_UpConv_Block_Last_Layer(C, 1, oheight, owidth)
            with torch.no_grad():
                l6.conv1.weight.copy_(w_guidance)
                l5.conv1.weight.copy_(w_blur)
                return l6(x), l5(x)
    finally:
        sys.path.remove(d)


def reference_guidance_heads_grads(x, w_guidance, w_blur, grad_guidance, grad_blur, oheight=0, owidth=0):
    """-> (dL/dx, dL/dw_guidance, dL/dw_blur) of L = sum(guidance * grad_guidance) + sum(blur * grad_blur) by torch autograd through the UNMODIFIED
    reference heads (torch_resnet_cspn_nyu.py:187-206, Unpool :41-54): what back-propagating through :372-373 gives the backbone and the two convs."""
    import sys
    d = os.path.dirname(REF_MODEL)
    sys.path.insert(0, d)
    try:
        with cuda_is_identity():
            spec = importlib.util.spec_from_file_location("_reference_resnet_cspn", REF_MODEL)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            C = x.shape[1]
            l6 = mod.Simple_Gudi_UpConv_Block_Last_Layer(C, 8, oheight, owidth)
            l5 = mod.Simple_Gudi_UpConv_Block_Last_Layer(C, 1, oheight, owidth)
            with torch.no_grad():
                l6.conv1.weight.copy_(w_guidance)
                l5.conv1.weight.copy_(w_blur)
            xr = x.clone().requires_grad_(True)
            g, b = l6(xr), l5(xr)
            (g * grad_guidance).sum().add((b * grad_blur).sum()).backward()
            return xr.grad.detach(), l6.conv1.weight.grad.detach(), l5.conv1.weight.grad.detach()
    finally:
        sys.path.remove(d)
