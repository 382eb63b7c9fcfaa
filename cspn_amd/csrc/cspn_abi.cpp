// cspn_abi.cpp -- extern "C" entry points declared in include/cspn_amd.h.
// Replaces the call boundary of Affinity_Propagate.forward
// (reference cspn_pytorch/models/cspn.py:42-83) and of the chained
// fluid.layers.affinity_propagate calls (reference cspn_paddle/demo.py:41-52).
#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "cspn_common.h"

namespace cspn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

int num_cus() {
    static std::atomic<int> cache[64];
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        v = cache[dev].load(std::memory_order_relaxed);
        if (v > 0) return v;
    }
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

static int check_common(const void* a, const void* b, const void* out, int n_iter, int norm, const void* ws,
                        size_t ws_bytes, size_t need, int norm_max = CSPN_NORM_NONE) {
    if (!a || !b || !out) { set_error("null tensor pointer"); return CSPN_E_BADARG; }
    if (n_iter < 0) { set_error("n_iter must be >= 0 (got %d)", n_iter); return CSPN_E_BADARG; }
    if (norm < CSPN_NORM_8SUM || norm > CSPN_NORM_PRENORM) { set_error("unknown norm_type %d", norm); return CSPN_E_BADARG; }
    if (norm > norm_max) { set_error("norm_type CSPN_NORM_PRENORM is taken by the 2D entry points only"); return CSPN_E_UNSUPPORTED; }
    if (need && (!ws || ws_bytes < need)) {
        set_error("workspace too small: need %zu bytes, got %zu", need, ws_bytes);
        return CSPN_E_WORKSPACE;
    }
    if (need && ((uintptr_t)ws & 255u)) { set_error("workspace must be 256-byte aligned"); return CSPN_E_WORKSPACE; }
    return 0;
}

}  // namespace cspn

using namespace cspn;

extern "C" {

int cspn_abi_version(void) { return CSPN_ABI_VERSION; }
const char* cspn_last_error(void) { return g_err; }

int cspn2d_auto_algo(int B, int H, int W, int n_iter) {
    if (fused2d_supported(B, H, W, n_iter)) return CSPN_ALGO_FUSED;
    return padded2d_supported(B, H, W, n_iter) ? CSPN_ALGO_FUSED_PADDED : CSPN_ALGO_STEPWISE;
}

size_t cspn2d_workspace_bytes(int B, int H, int W, int n_iter) {
    if (B <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    size_t a = stepwise2d_workspace(B, H, W, n_iter);
    size_t b = fused2d_supported(B, H, W, n_iter) ? fused2d_workspace(B, H, W, n_iter) : 0;
    if (padded2d_supported(B, H, W, n_iter)) b = padded2d_workspace(B, H, W, n_iter);
    return a > b ? a : b;  // large enough for either algo so callers can A/B
}

int cspn2d_forward_f32_algo(const float* guidance, const float* blur, const float* sparse, float* out, int B,
                            int H, int W, int n_iter, int norm_type, int algo, void* ws, size_t ws_bytes,
                            cspn_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d H=%d W=%d", B, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if ((long long)B * H * W > 0x7fffffffLL / 9) { set_error("tensor too large for 32-bit plane indexing"); return CSPN_E_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    if (algo == CSPN_ALGO_AUTO) {
        algo = cspn2d_auto_algo(B, H, W, n_iter);
        if (algo == CSPN_ALGO_FUSED && ((uintptr_t)out & 15u) != 0) algo = CSPN_ALGO_STEPWISE;   // (the fused kernels store aligned float4)
        // the padded path runs the fused kernels on planes inside the caller's workspace: those need the same alignment
        if (algo == CSPN_ALGO_FUSED_PADDED && ((uintptr_t)ws & 15u) != 0) algo = CSPN_ALGO_STEPWISE;
    }
    if (algo == CSPN_ALGO_FUSED_PADDED && n_iter == 0) algo = CSPN_ALGO_STEPWISE;   // n_iter == 0 is the identity copy below whatever the algo (reference cspn.py:61,66,83)
    if (algo == CSPN_ALGO_FUSED_PADDED) {
        if (!padded2d_supported(B, H, W, n_iter)) { set_error("FUSED_PADDED needs W %% 4 != 0 and a shape the fused kernels take (B=%d H=%d W=%d n_iter=%d)", B, H, W, n_iter); return CSPN_E_UNSUPPORTED; }
        if (((uintptr_t)ws & 15u) != 0) { set_error("FUSED_PADDED needs a 16-byte aligned workspace (its padded planes live there)"); return CSPN_E_UNSUPPORTED; }
        if (int e = check_common(guidance, blur, out, n_iter, norm_type, ws, ws_bytes, padded2d_workspace(B, H, W, n_iter), CSPN_NORM_PRENORM)) return e;
        return padded2d_forward(guidance, blur, sparse, out, B, H, W, n_iter, norm_type, ws, st);
    }
    if (algo != CSPN_ALGO_STEPWISE && algo != CSPN_ALGO_FUSED && algo != CSPN_ALGO_FUSED_CXX) { set_error("unknown algo %d", algo); return CSPN_E_BADARG; }
    const bool fused = algo == CSPN_ALGO_FUSED || algo == CSPN_ALGO_FUSED_CXX;
    if (fused && !fused2d_supported(B, H, W, n_iter)) {
        set_error("fused kernel does not support B=%d H=%d W=%d n_iter=%d", B, H, W, n_iter);
        return CSPN_E_UNSUPPORTED;
    }
    size_t need = n_iter == 0 ? 0
                  : (fused ? fused2d_workspace(B, H, W, n_iter)
                                             : stepwise2d_workspace(B, H, W, n_iter));
    if (int e = check_common(guidance, blur, out, n_iter, norm_type, ws, ws_bytes, need, CSPN_NORM_PRENORM)) return e;
    if (n_iter == 0) {  // reference cspn.py:61,66,83: the loop body never runs
        hipError_t e = hipMemcpyAsync(out, blur, sizeof(float) * (size_t)B * H * W, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    if (fused) return fused2d_forward(guidance, blur, sparse, out, B, H, W, n_iter, norm_type, ws, st, algo == CSPN_ALGO_FUSED);
    return stepwise2d_forward(guidance, blur, sparse, out, B, H, W, n_iter, norm_type, ws, st);
}

int cspn2d_forward_f32(const float* guidance, const float* blur, const float* sparse, float* out, int B, int H,
                       int W, int n_iter, int norm_type, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    return cspn2d_forward_f32_algo(guidance, blur, sparse, out, B, H, W, n_iter, norm_type, CSPN_ALGO_AUTO, ws,
                                   ws_bytes, stream);
}

// ---- SURVEY 8f-2, second alternative: the normalisation done by the producer of the guidance (include/cspn_amd.h) ----
int cspn2d_normalize_f32(const float* guidance, float* wb, int B, int H, int W, int norm_type, cspn_stream_t stream) {
    if (!guidance || !wb || B <= 0 || H <= 0 || W <= 0) { set_error("bad argument"); return CSPN_E_BADARG; }
    if (norm_type != CSPN_NORM_8SUM && norm_type != CSPN_NORM_8SUM_ABS) { set_error("cspn2d_normalize_f32: norm_type must be 8SUM or 8SUM_ABS (got %d)", norm_type); return CSPN_E_BADARG; }
    if ((long long)B * H * W > 0x7fffffffLL / 9) { set_error("tensor too large for 32-bit plane indexing"); return CSPN_E_UNSUPPORTED; }
    return normalize2d(guidance, wb, B, H, W, norm_type, (hipStream_t)stream);
}

int cspn2d_forward_prenorm_f32(const float* wb, const float* blur, const float* sparse, float* out, int B, int H, int W, int n_iter,
                               void* ws, size_t ws_bytes, cspn_stream_t stream) {
    return cspn2d_forward_f32_algo(wb, blur, sparse, out, B, H, W, n_iter, CSPN_NORM_PRENORM, CSPN_ALGO_AUTO, ws, ws_bytes, stream);
}

size_t cspn2d_backward_workspace_bytes(int B, int H, int W, int n_iter) {
    if (B <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    return backward2d_workspace(B, H, W, n_iter);
}

int cspn2d_backward_f32(const float* guidance, const float* blur, const float* sparse, const float* grad_out,
                        float* grad_guidance, float* grad_blur, int B, int H, int W, int n_iter, int norm_type, void* ws,
                        size_t ws_bytes, cspn_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d H=%d W=%d", B, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if (n_iter < 1) { set_error("backward needs n_iter >= 1 (got %d)", n_iter); return CSPN_E_BADARG; }
    if ((long long)B * H * W > 0x7fffffffLL / 9) { set_error("tensor too large for 32-bit plane indexing"); return CSPN_E_UNSUPPORTED; }
    if (!grad_out) { set_error("null grad_out"); return CSPN_E_BADARG; }
    if (int e = check_common(guidance, blur, grad_out, n_iter, norm_type, ws, ws_bytes, backward2d_workspace(B, H, W, n_iter), CSPN_NORM_PRENORM)) return e;
    if (!grad_guidance && !grad_blur) return 0;
    return backward2d(guidance, blur, sparse, grad_out, grad_guidance, grad_blur, B, H, W, n_iter, norm_type, ws, (hipStream_t)stream);
}

size_t cspn2d_history_bytes(int B, int H, int W, int n_iter) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return history2d_bytes(B, H, W, n_iter);
}

int cspn2d_forward_history_f32(const float* guidance, const float* blur, const float* sparse, float* out, void* history,
                               size_t history_bytes, int B, int H, int W, int n_iter, int norm_type, void* ws, size_t ws_bytes,
                               cspn_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d H=%d W=%d", B, H, W); return CSPN_E_BADARG; }
    const size_t hb = history2d_bytes(B, H, W, n_iter);
    if (hb == 0) { set_error("no history mode for B=%d H=%d W=%d n_iter=%d", B, H, W, n_iter); return CSPN_E_UNSUPPORTED; }
    if (!history || history_bytes < hb || ((uintptr_t)history & 255u)) { set_error("history buffer too small or misaligned: need %zu bytes", hb); return CSPN_E_WORKSPACE; }
    if (int e = check_common(guidance, blur, out, n_iter, norm_type, ws, ws_bytes, fused2d_workspace(B, H, W, n_iter), CSPN_NORM_PRENORM)) return e;
    if (((uintptr_t)out & 15u) != 0) { set_error("output must be 16-byte aligned"); return CSPN_E_UNSUPPORTED; }
    return forward2d_history(guidance, blur, sparse, out, history, B, H, W, n_iter, norm_type, ws, (hipStream_t)stream);
}

size_t cspn2d_backward_history_workspace_bytes(int B, int H, int W, int n_iter) {
    if (B <= 0 || H <= 0 || W <= 0 || history2d_bytes(B, H, W, n_iter) == 0) return 0;
    return backward2d_history_workspace(B, H, W);
}

int cspn2d_backward_history_f32(const float* guidance, const float* blur, const float* sparse, const float* grad_out,
                                const void* history, size_t history_bytes, float* grad_guidance, float* grad_blur, int B, int H,
                                int W, int n_iter, int norm_type, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d H=%d W=%d", B, H, W); return CSPN_E_BADARG; }
    const size_t hb = history2d_bytes(B, H, W, n_iter);
    if (hb == 0) { set_error("no history mode for B=%d H=%d W=%d n_iter=%d", B, H, W, n_iter); return CSPN_E_UNSUPPORTED; }
    if (!grad_out) { set_error("null grad_out"); return CSPN_E_BADARG; }
    if (!history || history_bytes < hb) { set_error("history buffer too small: need %zu bytes", hb); return CSPN_E_WORKSPACE; }
    if (int e = check_common(guidance, blur, grad_out, n_iter, norm_type, ws, ws_bytes, backward2d_history_workspace(B, H, W), CSPN_NORM_PRENORM)) return e;
    if (!grad_guidance && !grad_blur) return 0;
    return backward2d_history(guidance, blur, sparse, grad_out, history, grad_guidance, grad_blur, B, H, W, n_iter, norm_type, ws,
                              (hipStream_t)stream);
}

// a persistent 3D launch of an EARLIER call gave up (cspn3d_persistent.hip): report it once, through whichever 3D call comes next
static int async_failure_of_earlier_call() {
    if (persistent3d_take_status() == 0) return 0;
    set_error("an earlier cspn3d call's persistent kernel gave up waiting for a neighbouring workgroup (not all of its workgroups were "
              "resident: the device is shared with other work that holds compute units); that call's outputs are NaN-filled / invalid. "
              "Re-run it, or request CSPN_ALGO3D_STEPWISE");
    return CSPN_E_ASYNC;
}

int cspn3d_check_status(cspn_stream_t stream) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { set_error("hipStreamSynchronize: %s", hipGetErrorString(e)); return (int)e; }
    return async_failure_of_earlier_call();
}

size_t cspn3d_workspace_bytes(int B, int D, int H, int W, int n_iter) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    return stepwise3d_workspace(B, D, H, W, n_iter);
}

size_t cspn3d_workspace_bytes_ex(int B, int D, int H, int W, int n_iter, int norm_type, int has_sparse) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    return forward3d_workspace(B, D, H, W, n_iter, norm_type, has_sparse != 0);
}

int cspn3d_forward_f32(const float* gate, const float* feat, const float* sparse, float* out, int B, int D, int H,
                       int W, int n_iter, int norm_type, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    return cspn3d_forward_f32_algo(gate, feat, sparse, out, B, D, H, W, n_iter, norm_type, CSPN_ALGO3D_AUTO, ws, ws_bytes, stream);
}

int cspn3d_forward_f32_algo(const float* gate, const float* feat, const float* sparse, float* out, int B, int D, int H,
                            int W, int n_iter, int norm_type, int algo, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d D=%d H=%d W=%d", B, D, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if ((long long)B * D * H * W > 0x7fffffffLL / 27) { set_error("tensor too large for 32-bit plane indexing"); return CSPN_E_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    if (algo < CSPN_ALGO3D_AUTO || algo > CSPN_ALGO3D_PERSISTENT) { set_error("unknown 3D algo %d", algo); return CSPN_E_BADARG; }
    if (int e = async_failure_of_earlier_call()) return e;
    // misaligned tensors cannot take the 16-byte paths: they fold like the normalising modes (cspn3d_workspace_bytes())
    const bool aligned = ((((uintptr_t)gate | (uintptr_t)feat | (uintptr_t)out | (uintptr_t)ws) & 15u) == 0);
    size_t need = n_iter == 0 ? 0 : (aligned ? forward3d_workspace(B, D, H, W, n_iter, norm_type, sparse != nullptr)
                                             : stepwise3d_workspace(B, D, H, W, n_iter));
    if (int e = check_common(gate, feat, out, n_iter, norm_type, ws, ws_bytes, need)) return e;
    if (n_iter == 0) {
        hipError_t e = hipMemcpyAsync(out, feat, sizeof(float) * (size_t)B * D * H * W, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    return stepwise3d_forward(gate, feat, sparse, out, B, D, H, W, n_iter, norm_type, ws, st, algo);
}

int cspn3d_multi_supported(int B, int C, int D, int H, int W, int n_iter) {
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return persistent3d_multi_supported(B, C, D, H, W, n_iter) ? 1 : 0;
}

int cspn3d_forward_multi_f32(const float* gate, const float* feat, float* out, int B, int C, int D, int H, int W, int n_iter,
                             void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (B < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if (int e = async_failure_of_earlier_call()) return e;
    if (int e = check_common(gate, feat, out, n_iter, CSPN_NORM_NONE, ws, ws_bytes,
                             n_iter == 0 ? 0 : forward3d_workspace(B, D, H, W, n_iter, CSPN_NORM_NONE, false))) return e;
    const bool aligned = ((((uintptr_t)gate | (uintptr_t)feat | (uintptr_t)out | (uintptr_t)ws) & 15u) == 0);
    if (!aligned || !persistent3d_multi_supported(B, C, D, H, W, n_iter)) {
        set_error("cspn3d_forward_multi_f32 runs the persistent kernel only (W %% 4 == 0, 2 <= n_iter <= 60, 16-byte aligned tensors, "
                  "volume resident on the device): loop over the channels with cspn3d_forward_f32 for B=%d C=%d D=%d H=%d W=%d n_iter=%d",
                  B, C, D, H, W, n_iter);
        return CSPN_E_UNSUPPORTED;
    }
    return persistent3d_forward_multi(gate, feat, out, B, C, D, H, W, n_iter, ws, (hipStream_t)stream);
}

size_t cspn3d_backward_workspace_bytes(int B, int D, int H, int W, int n_iter) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    return backward3d_workspace(B, D, H, W, n_iter);
}

int cspn3d_backward_f32(const float* gate, const float* feat, const float* grad_out, float* grad_gate, float* grad_feat, int B,
                        int D, int H, int W, int n_iter, int norm_type, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d D=%d H=%d W=%d", B, D, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if ((long long)B * D * H * W > 0x7fffffffLL / 27) { set_error("tensor too large for 32-bit plane indexing"); return CSPN_E_UNSUPPORTED; }
    if (norm_type != CSPN_NORM_NONE) {
        set_error("the 3D backward covers the Paddle contract only (norm_type NONE: gates used as given, no mask)");
        return CSPN_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int e = async_failure_of_earlier_call()) return e;
    if (int e = check_common(gate, feat, grad_out, n_iter, norm_type, ws, ws_bytes, n_iter == 0 ? 0 : backward3d_workspace(B, D, H, W, n_iter))) return e;
    if (!grad_gate && !grad_feat) return 0;
    const size_t bytes = sizeof(float) * (size_t)B * D * H * W;
    if (n_iter == 0) {   // identity: dL/dfeat = dL/dout, the gates are not used
        hipError_t e = hipSuccess;
        if (grad_feat) e = hipMemcpyAsync(grad_feat, grad_out, bytes, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && grad_gate) e = hipMemsetAsync(grad_gate, 0, 26 * bytes, st);
        if (e != hipSuccess) { set_error("hipMemcpyAsync / hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    return backward3d(gate, feat, grad_out, grad_gate, grad_feat, B, D, H, W, n_iter, ws, st);
}

size_t cspn3d_backward_multi_workspace_bytes(int B, int C, int D, int H, int W, int n_iter) {
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || n_iter <= 0) return 0;
    return backward3d_workspace(B, D, H, W, n_iter, C);
}

int cspn3d_backward_multi_f32(const float* gate, const float* feat, const float* grad_out, float* grad_gate, float* grad_feat, int B,
                              int C, int D, int H, int W, int n_iter, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (B < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) { set_error("bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W); return CSPN_E_BADARG; }
    if (B == 0) return 0;
    if ((long long)B * D * H * W > 0x7fffffffLL / 27 || (long long)B * C * D * H * W > 0x7fffffffLL / 2) {
        set_error("tensor too large for 32-bit plane indexing");
        return CSPN_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int e = async_failure_of_earlier_call()) return e;
    if (int e = check_common(gate, feat, grad_out, n_iter, CSPN_NORM_NONE, ws, ws_bytes, n_iter == 0 ? 0 : backward3d_workspace(B, D, H, W, n_iter, C))) return e;
    if (!grad_gate && !grad_feat) return 0;
    if (n_iter == 0) {   // identity: dL/dfeat = dL/dout, the gates are not used
        hipError_t e = hipSuccess;
        if (grad_feat) e = hipMemcpyAsync(grad_feat, grad_out, sizeof(float) * (size_t)B * C * D * H * W, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && grad_gate) e = hipMemsetAsync(grad_gate, 0, 26 * sizeof(float) * (size_t)B * D * H * W, st);
        if (e != hipSuccess) { set_error("hipMemcpyAsync / hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    return backward3d(gate, feat, grad_out, grad_gate, grad_feat, B, D, H, W, n_iter, ws, st, false, C);
}

}  // extern "C"
