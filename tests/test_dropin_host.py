"""The drop-in boundary end to end (SURVEY §8b/§8c "full model as host"): a host network that does
`import cspn as post_process` like reference cspn_pytorch/models/torch_resnet_cspn_nyu.py:12 picks up the HIP engine when
cspn_amd/dropin is first on sys.path, trains through it, and loads a reference-style checkpoint that carries the
`post_process_layer.sum_conv.weight` key the reference module registers during its first forward (cspn.py:44-53)."""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "cspn_amd", "dropin")
FIX = os.path.join(ROOT, "tests", "fixtures")


def _import_host():
    for m in ("cspn", "host_model"):
        sys.modules.pop(m, None)
    sys.path.insert(0, FIX)
    sys.path.insert(0, DROPIN)       # ahead of everything, like putting it ahead of cspn_pytorch/models
    try:
        return importlib.import_module("host_model")
    finally:
        sys.path.remove(DROPIN)
        sys.path.remove(FIX)


def test_host_model_binds_to_the_engine_and_filters_checkpoint_keys():
    import cspn_amd
    hm = _import_host()
    assert hm.post_process.Affinity_Propagate is cspn_amd.Affinity_Propagate
    net = hm.HostNet(cspn_config={'step': 24, 'kernel': 3, 'norm_type': '8sum'})
    assert isinstance(net.post_process_layer, cspn_amd.Affinity_Propagate)
    assert not any(k.startswith("post_process_layer") for k in net.state_dict())
    # a checkpoint written by the REFERENCE model after a forward carries sum_conv.weight (cspn.py:44-53)
    ckpt = {k: v.clone() + 1.0 for k, v in net.state_dict().items()}
    ckpt["post_process_layer.sum_conv.weight"] = torch.ones(1, 8, 1, 1, 1)
    ckpt = {"module." + k: v for k, v in ckpt.items()}                     # saved from nn.DataParallel
    ckpt = {k[7:]: v for k, v in ckpt.items()}                              # update_model.remove_moudle
    net.load_state_dict(hm.update_model(net, ckpt))                         # train.py:150-156 / eval.py:105-110
    assert torch.equal(net.conv1_1.weight, ckpt["conv1_1.weight"])
    with pytest.raises(AssertionError):
        hm.HostNet(cspn_config={'kernel': 5})
    with pytest.raises(AssertionError):
        hm.HostNet(cspn_config={'norm_type': 'bogus'})


@pytest.mark.gpu
def test_host_model_forward_backward_on_gpu_matches_oracle():
    from helpers import assert_close_tight
    from oracle import cspn2d_oracle
    hm = _import_host()
    torch.manual_seed(3)
    net = hm.HostNet().to("cuda:0")
    x = torch.rand(2, 4, 64, 256, device="cuda:0")
    x[:, 3] = (torch.rand(2, 64, 256, device="cuda:0") < 0.02).float() * (x[:, 3] * 10 + 0.1)   # sparse depth channel
    x[:, :3] = x[:, :3] * 2 - 1
    target = torch.rand(2, 1, 64, 256, device="cuda:0")
    out = net(x)
    assert out.shape == (2, 1, 64, 256) and out.is_cuda
    loss = (out - target).abs().mean()
    loss.backward()                                                        # train.py:196-198
    torch.cuda.synchronize()
    for p in (net.conv1_1.weight, net.gud_up_proj_layer5.weight, net.gud_up_proj_layer6.weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0
    # the module's output == the oracle on the very tensors the backbone handed it
    with torch.no_grad():
        f = net.relu(net.conv1_1(x))
        g, b, s = net.gud_up_proj_layer6(f), net.gud_up_proj_layer5(f), x.narrow(1, 3, 1).clone()
        assert_close_tight(net(x).cpu().numpy(), cspn2d_oracle(g.cpu(), b.cpu(), s.cpu(), 24, "8sum"), "host model")
    # one optimizer step changes the output (the gradient really reached the heads)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    opt.step()
    with torch.no_grad():
        assert float((net(x) - out).abs().max()) > 0
