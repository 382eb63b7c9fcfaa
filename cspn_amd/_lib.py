"""ctypes binding of libcspn_amd.so (the C ABI in include/cspn_amd.h).

There is NO CPU fallback and no PyTorch re-implementation behind this module:
if the HIP library is missing or a call fails, it raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSPN_AMD_LIB") or os.path.join(_HERE, "libcspn_amd.so")
CSRC = os.path.join(_HERE, "csrc")

NORM_TYPES = {"8sum": 0, "8sum_abs": 1, "none": 2, "prenorm": 3}
ALGOS = {"auto": 0, "stepwise": 1, "fused": 2, "fused_cxx": 3, "fused_padded": 4}
ALGOS_3D = {"auto": 0, "stepwise": 1, "persistent": 2}
ABI_VERSION = 5

HOOKS_PATH = os.path.join(os.path.dirname(LIB_PATH), "libcspn_amd_hooks.so")

_lib = None
_hooks = None


class CspnError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into cspn_amd/libcspn_amd.so (in-tree)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    if not verbose:
        args.append("-s")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CspnError(
            "cspn_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cspn_amd/csrc`). There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_size_t, vp = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p
    lib.cspn_abi_version.restype = c_int
    v = lib.cspn_abi_version()
    if v != ABI_VERSION:   # (before any symbol lookup: a stale library must fail with this message, not an AttributeError)
        raise CspnError("cspn_amd: ABI version mismatch: library %d, binding %d -- rebuild with `make -C cspn_amd/csrc`" % (v, ABI_VERSION))
    lib.cspn_last_error.restype = ctypes.c_char_p
    lib.cspn2d_workspace_bytes.restype = c_size_t
    lib.cspn2d_workspace_bytes.argtypes = [c_int] * 4
    lib.cspn2d_auto_algo.restype = c_int
    lib.cspn2d_auto_algo.argtypes = [c_int] * 4
    lib.cspn2d_forward_f32.restype = c_int
    lib.cspn2d_forward_f32.argtypes = [vp, vp, vp, vp] + [c_int] * 5 + [vp, c_size_t, vp]
    lib.cspn2d_forward_f32_algo.restype = c_int
    lib.cspn2d_forward_f32_algo.argtypes = [vp, vp, vp, vp] + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn2d_normalize_f32.restype = c_int
    lib.cspn2d_normalize_f32.argtypes = [vp, vp] + [c_int] * 4 + [vp]
    lib.cspn2d_forward_prenorm_f32.restype = c_int
    lib.cspn2d_forward_prenorm_f32.argtypes = [vp, vp, vp, vp] + [c_int] * 4 + [vp, c_size_t, vp]
    lib.cspn2d_backward_workspace_bytes.restype = c_size_t
    lib.cspn2d_backward_workspace_bytes.argtypes = [c_int] * 4
    lib.cspn2d_backward_f32.restype = c_int
    lib.cspn2d_backward_f32.argtypes = [vp] * 6 + [c_int] * 5 + [vp, c_size_t, vp]
    lib.cspn2d_history_bytes.restype = c_size_t
    lib.cspn2d_history_bytes.argtypes = [c_int] * 4
    lib.cspn2d_forward_history_f32.restype = c_int
    lib.cspn2d_forward_history_f32.argtypes = [vp] * 5 + [c_size_t] + [c_int] * 5 + [vp, c_size_t, vp]
    lib.cspn2d_backward_history_workspace_bytes.restype = c_size_t
    lib.cspn2d_backward_history_workspace_bytes.argtypes = [c_int] * 4
    lib.cspn2d_backward_history_f32.restype = c_int
    lib.cspn2d_backward_history_f32.argtypes = [vp] * 5 + [c_size_t, vp, vp] + [c_int] * 5 + [vp, c_size_t, vp]
    lib.cspn_metrics_workspace_bytes.restype = c_size_t
    lib.cspn_metrics_workspace_bytes.argtypes = [c_size_t]
    lib.cspn_metrics_f32.restype = c_int
    lib.cspn_metrics_f32.argtypes = [vp, vp, c_size_t, vp, vp, c_size_t, vp]
    lib.cspn_l1_backward_f32.restype = c_int
    lib.cspn_l1_backward_f32.argtypes = [vp, vp, vp, vp, vp, c_size_t, vp]
    lib.cspn3d_backward_workspace_bytes.restype = c_size_t
    lib.cspn3d_backward_workspace_bytes.argtypes = [c_int] * 5
    lib.cspn3d_backward_f32.restype = c_int
    lib.cspn3d_backward_f32.argtypes = [vp] * 5 + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn_unpool_f32.restype = c_int
    lib.cspn_unpool_f32.argtypes = [vp, vp, c_size_t, c_int, c_int, c_int, vp]
    lib.cspn_guidance_head_workspace_bytes.restype = c_size_t
    lib.cspn_guidance_head_workspace_bytes.argtypes = [c_int]
    lib.cspn_guidance_head_f32.restype = c_int
    lib.cspn_guidance_head_f32.argtypes = [vp] * 5 + [c_int] * 7 + [vp, c_size_t, vp]
    lib.cspn_guidance_head_backward_workspace_bytes.restype = c_size_t
    lib.cspn_guidance_head_backward_workspace_bytes.argtypes = [c_int] * 4
    lib.cspn_guidance_head_backward_f32.restype = c_int
    lib.cspn_guidance_head_backward_f32.argtypes = [vp] * 8 + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn_unpool_backward_f32.restype = c_int
    lib.cspn_unpool_backward_f32.argtypes = [vp, vp, c_size_t, c_int, c_int, c_int, vp]
    lib.cspn_sparse_sample_workspace_bytes.restype = c_size_t
    lib.cspn_sparse_sample_workspace_bytes.argtypes = [c_size_t]
    lib.cspn_sparse_sample_f32.restype = c_int
    lib.cspn_sparse_sample_f32.argtypes = [vp, vp, c_size_t, c_size_t, c_int, c_int, ctypes.c_ulonglong, vp, c_size_t, vp]
    lib.cspn3d_workspace_bytes.restype = c_size_t
    lib.cspn3d_workspace_bytes.argtypes = [c_int] * 5
    lib.cspn3d_forward_f32.restype = c_int
    lib.cspn3d_forward_f32.argtypes = [vp, vp, vp, vp] + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn3d_workspace_bytes_ex.restype = c_size_t
    lib.cspn3d_workspace_bytes_ex.argtypes = [c_int] * 7
    lib.cspn3d_forward_f32_algo.restype = c_int
    lib.cspn3d_forward_f32_algo.argtypes = [vp, vp, vp, vp] + [c_int] * 7 + [vp, c_size_t, vp]
    lib.cspn3d_multi_supported.restype = c_int
    lib.cspn3d_multi_supported.argtypes = [c_int] * 6
    lib.cspn3d_forward_multi_f32.restype = c_int
    lib.cspn3d_forward_multi_f32.argtypes = [vp, vp, vp] + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn3d_backward_multi_workspace_bytes.restype = c_size_t
    lib.cspn3d_backward_multi_workspace_bytes.argtypes = [c_int] * 6
    lib.cspn3d_backward_multi_f32.restype = c_int
    lib.cspn3d_backward_multi_f32.argtypes = [vp] * 5 + [c_int] * 6 + [vp, c_size_t, vp]
    lib.cspn3d_check_status.restype = c_int
    lib.cspn3d_check_status.argtypes = [vp]
    _lib = lib
    return lib


def load_hooks():
    """libcspn_amd_hooks.so: what tests and measuring tools need beyond the ABI (plan dumps, plan A/B, the muted-workgroup launch
    of the persistent 3D kernel ...).  It links against libcspn_amd.so and calls the same code with the test's choice as an
    argument; the product library itself exports no cspn_debug_* symbol and keeps no test state."""
    global _hooks
    if _hooks is not None:
        return _hooks
    load()   # the product library first (the hooks library resolves its symbols against it)
    if not os.path.exists(HOOKS_PATH):
        raise CspnError("cspn_amd: %s not found -- `make -C cspn_amd/csrc` builds it next to the product library" % HOOKS_PATH)
    h = ctypes.CDLL(HOOKS_PATH)
    c_int, vp = ctypes.c_int, ctypes.c_void_p
    ip = ctypes.POINTER(c_int)
    h.cspn_debug_tsw_plan_geo.restype = c_int
    h.cspn_debug_tsw_plan_geo.argtypes = [c_int] * 5 + [ip]
    h.cspn_debug_tsw_plan_cuts.restype = c_int
    h.cspn_debug_tsw_plan_cuts.argtypes = [c_int] * 5 + [ip, ip, ip]
    h.cspn_debug_tsw_dump_plan.restype = c_int
    h.cspn_debug_tsw_dump_plan.argtypes = [c_int] * 5 + [vp, vp, vp]
    h.cspn_debug_forward2d_plan.restype = c_int
    h.cspn_debug_forward2d_plan.argtypes = [vp] * 4 + [c_int] * 6 + [vp, vp]
    h.cspn_debug_3d_geo.restype = c_int
    h.cspn_debug_3d_geo.argtypes = [c_int] * 5 + [ip]
    h.cspn_debug_3d_persistent_error.restype = c_int
    h.cspn_debug_3d_persistent_error.argtypes = [vp] + [c_int] * 4
    h.cspn_debug_3d_persistent_forward.restype = c_int
    h.cspn_debug_3d_persistent_forward.argtypes = [vp] * 3 + [c_int] * 7 + [vp, vp]
    h.cspn_debug_3d_backward_stepwise.restype = c_int
    h.cspn_debug_3d_backward_stepwise.argtypes = [vp] * 5 + [c_int] * 5 + [vp, vp]
    h.cspn_debug_sited8_supported.restype = c_int
    h.cspn_debug_sited8_supported.argtypes = [c_int] * 4
    h.cspn_debug_guidance_to_sited8.restype = c_int
    h.cspn_debug_guidance_to_sited8.argtypes = [vp, vp] + [c_int] * 4 + [vp]
    h.cspn_debug_forward_sited8.restype = c_int
    h.cspn_debug_forward_sited8.argtypes = [vp, vp, vp, vp] + [c_int] * 5 + [vp]
    _hooks = h
    return h


def check(rc, what):
    if rc != 0:
        msg = load().cspn_last_error().decode("utf-8", "replace")
        raise CspnError("%s failed (code %d): %s" % (what, rc, msg))
