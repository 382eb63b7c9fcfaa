"""A tiny host network that uses the CSPN module exactly the way the reference's ResNet does
(reference cspn_pytorch/models/torch_resnet_cspn_nyu.py): `import cspn as post_process` at module scope (:12),
`post_process.Affinity_Propagate(cfg['step'], cfg['kernel'], norm_type=cfg['norm_type'])` in
`_make_post_process_layer` (:344-347), `self.post_process_layer(guidance, x, sparse_depth)` with the sparse depth
narrowed out of the 4-channel RGB-D input (:351,372-375).  /root/reference does not exist on the GPU box, so this stands
in for torch_resnet_cspn_nyu.resnet50 in the drop-in test; the caller decides which `cspn` module is first on sys.path.
"""
import torch.nn as nn

import cspn as post_process  # noqa: E402  (the drop-in point)


def update_model(my_model, pretrained_dict):
    """the key filter of reference cspn_pytorch/models/update_model.py:16-23"""
    my_model_dict = my_model.state_dict()
    pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in my_model_dict}
    my_model_dict.update(pretrained_dict)
    return my_model_dict


class HostNet(nn.Module):
    def __init__(self, cspn_config=None):
        super(HostNet, self).__init__()
        cspn_config_default = {'step': 24, 'kernel': 3, 'norm_type': '8sum'}   # torch_resnet_cspn_nyu.py:281-283
        if cspn_config is not None:
            cspn_config_default.update(cspn_config)
        self.conv1_1 = nn.Conv2d(4, 16, kernel_size=3, padding=1, bias=False)
        self.relu = nn.ReLU(inplace=True)
        # the two bias-free heads (gud_up_proj_layer6: 64 -> 8, gud_up_proj_layer5: 64 -> 1, :318-319)
        self.gud_up_proj_layer6 = nn.Conv2d(16, 8, kernel_size=3, padding=1, bias=False)
        self.gud_up_proj_layer5 = nn.Conv2d(16, 1, kernel_size=3, padding=1, bias=False)
        self.post_process_layer = self._make_post_process_layer(cspn_config_default)

    def _make_post_process_layer(self, cspn_config=None):
        return post_process.Affinity_Propagate(cspn_config['step'],
                                               cspn_config['kernel'],
                                               norm_type=cspn_config['norm_type'])

    def forward(self, x):
        sparse_depth = x.narrow(1, 3, 1).clone()
        x = self.relu(self.conv1_1(x))
        guidance = self.gud_up_proj_layer6(x)
        x = self.gud_up_proj_layer5(x)
        x = self.post_process_layer(guidance, x, sparse_depth)
        return x
