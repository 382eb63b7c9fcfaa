"""cspn_amd -- MI355X-native CSPN propagation engine (hand-written HIP for gfx950).

Scope: the one hot path of XinJCheng/CSPN -- Affinity_Propagate
(reference cspn_pytorch/models/cspn.py:14-83) and its 3D call site
(reference cspn_paddle/demo.py:41-52).  See DESIGN.md / INTEGRATION.md."""
from ._lib import CspnError, build, load  # noqa: F401
from .cspn import Affinity_Propagate, propagate_prenorm  # noqa: F401
from .functional import (affinity_propagate, cspn2d_backward, cspn2d_forward_sited8, guidance_to_sited8, cspn2d_normalize, cspn2d_backward_from_history, cspn2d_forward,  # noqa: F401
                         cspn2d_forward_with_history, cspn2d_history_bytes, cspn3d_forward, cspn3d_forward_multi, cspn3d_backward, cspn3d_backward_multi, cspn3d_check_status)

__all__ = ["Affinity_Propagate", "propagate_prenorm", "cspn2d_forward", "cspn2d_normalize", "cspn2d_backward", "cspn3d_forward", "cspn3d_forward_multi", "cspn3d_backward", "cspn3d_backward_multi", "cspn3d_check_status", "affinity_propagate", "build", "load",
           "CspnError"]
