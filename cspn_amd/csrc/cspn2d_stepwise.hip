// cspn2d_stepwise.hip -- general-shape 2D path: one "fold" launch + one launch per
// iteration.  Used for shapes the fused kernel does not take, and as an on-device
// cross-check of it.  Arithmetic: reference cspn_pytorch/models/cspn.py:42-172,
// restated as
//     H_{t+1}(p) = c'(p) + sum_k w'_k(p) * H_t(p + off_k)
// with w'_k = (1-m) w_k, c' = (1-m)(1-sigma) H_0 + m H_0, m = sign(sparse).
#include "cspn_common.h"

namespace cspn {

// Per pixel: gather the eight neighbour-sited affinities (cspn.py:105-132),
// normalise by their abs-sum (cspn.py:135-138), fold centre term (cspn.py:76)
// and sparse pinning (cspn.py:81) into 9 coefficients.  wf: [9][B*H*W].
__global__ __launch_bounds__(256) void fold2d_kernel(const float* __restrict__ g, const float* __restrict__ blur,
                                                      const float* __restrict__ sparse, float* __restrict__ wf,
                                                      int B, int H, int W, int norm) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    const float* gb = g + (size_t)b * 8 * HW;
    float G[8], S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v;
        if (norm == CSPN_NORM_NONE || norm == CSPN_NORM_PRENORM) {   // used as given, centre-sited
            v = gb[k * HW + r];
        } else {
            const int yy = y + dy2(k), xx = x + dx2(k);
            v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gb[k * HW + (size_t)yy * W + xx];
            if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
        }
        G[k] = v;
        S += fabsf(v);
    }
    float sigma = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (norm != CSPN_NORM_NONE && norm != CSPN_NORM_PRENORM) G[k] = G[k] / S;  // IEEE: 0/0 -> NaN like torch.div (cspn.py:138)
        sigma += G[k];
    }
    const float h0 = blur[idx];
    const float m = sparse ? signf(sparse[idx]) : 0.f;
    const float om = 1.f - m;
    float c = (norm == CSPN_NORM_NONE) ? 0.f : (1.f - sigma) * h0;   // (PRENORM: the centre term stays, cspn.py:76)
    if (sparse) {
        c = om * c + m * h0;
#pragma unroll
        for (int k = 0; k < 8; ++k) G[k] *= om;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) wf[k * total + idx] = G[k];
    wf[8 * total + idx] = c;
}

__global__ __launch_bounds__(256) void step2d_kernel(const float* __restrict__ wf, const float* __restrict__ hin,
                                                      float* __restrict__ hout, int B, int H, int W) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    const float* hb = hin + (size_t)b * HW;
    float acc = wf[8 * total + idx];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y + dy2(k), xx = x + dx2(k);
        float hv = 0.f;  // ZeroPad2d (cspn.py:149-167)
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) hv = hb[(size_t)yy * W + xx];
        acc = fmaf(wf[k * total + idx], hv, acc);
    }
    hout[idx] = acc;
}

// reference affinity_normalization (cspn.py:85-144) as a stand-alone kernel: wb[B,8,H,W] = gate_wb, i.e. w_k(p) = G_k(p) / sum_j |G_j(p)|
// with G_k(p) = g~_k(p + off_k), zero outside the image -- what a producer head with a fused epilogue would emit and what
// norm PRENORM takes (second alternative).  One thread per pixel; 36 B read (L2 serves the shifted re-reads),
// 32 B written per pixel.
__global__ __launch_bounds__(256) void normalize2d_kernel(const float* __restrict__ g, float* __restrict__ wb, int B, int H, int W,
                                                           int norm) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    const float* gb = g + (size_t)b * 8 * HW;
    float G[8], S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y + dy2(k), xx = x + dx2(k);
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gb[k * HW + (size_t)yy * W + xx];
        if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
        G[k] = v;
        S += fabsf(v);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) wb[((size_t)b * 8 + k) * HW + r] = G[k] / S;   // IEEE division: 0/0 = NaN (cspn.py:138)
}

int normalize2d(const float* g, float* wb, int B, int H, int W, int norm, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(normalize2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g, wb, B, H, W, norm);
    return check_launch("normalize2d_kernel");
}

size_t stepwise2d_workspace(int B, int H, int W, int n_iter) {
    (void)n_iter;
    const size_t total = (size_t)B * H * W;
    return (9 + 2) * total * sizeof(float);
}

int stepwise2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W,
                       int n_iter, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    float* wf = (float*)ws;
    float* ping[2] = {wf + 9 * total, wf + 10 * total};
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fold2d_kernel, dim3(blocks), dim3(256), 0, st, g, blur, sparse, wf, B, H, W, norm);
    if (int e = check_launch("fold2d_kernel")) return e;
    const float* src = blur;
    for (int it = 0; it < n_iter; ++it) {
        float* dst = (it == n_iter - 1) ? out : ping[it & 1];
        hipLaunchKernelGGL(step2d_kernel, dim3(blocks), dim3(256), 0, st, wf, src, dst, B, H, W);
        src = dst;
    }
    return check_launch("step2d_kernel");
}

}  // namespace cspn
