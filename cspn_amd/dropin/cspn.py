"""Put this directory on sys.path ahead of cspn_pytorch/models and the reference model's
`import cspn as post_process` (reference cspn_pytorch/models/torch_resnet_cspn_nyu.py:12)
binds to the HIP engine instead of the ZeroPad2d/cat/Conv3d module."""
from cspn_amd.cspn import Affinity_Propagate  # noqa: F401
