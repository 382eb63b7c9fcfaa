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

// ---- W % 4 != 0 (round 5).  The fused kernels move rows as whole 16-byte groups (4 columns per lane, aligned float4 stores), so rounds 1-4
// sent such images down the stepwise path: fold + one launch per iteration, 44 B/pixel PER ITERATION (304 x 1218 x 64, 24 iterations: 4.85 ms
// against 0.28 ms for 304 x 1216).  Now the inputs are laid out once with a row pitch Wp = W rounded up to a multiple of 4 -- the pad columns
// hold zeros -- in the caller's workspace, the fused path runs on H x Wp, and the owned columns of its output are copied back.  The pad
// columns must behave exactly like "outside the image" (reference cspn.py:105-132, 149-168: zero padding):
//   * norms 8SUM / 8SUM_ABS: the normalisation is done HERE for the real width (normalize2d with an output pitch: G_k(p) = 0 for a
//     neighbour outside the real image, S(p) over the real neighbours, IEEE 0/0 -> NaN as in the reference) and the fused kernel runs the
//     pre-normalised contract (CSPN_NORM_PRENORM) on the padded planes; a pad pixel has all-zero weights and level 0 = 0, so it stays exactly
//     0 = finite through every iteration, and a real pixel's weight towards it is 0: contribution 0 x 0.
//   * norms NONE / PRENORM (weights used as given): the eight planes are copied; a pad pixel's weights are 0 (it stays 0), and a real pixel's
//     weight towards it multiplies that 0 -- what the reference's zero-padded depth gives.
__global__ __launch_bounds__(256) void normalize2d_pitch_kernel(const float* __restrict__ g, float* __restrict__ wb, int B, int H, int W, int Wp,
                                                                 int norm) {
    const size_t HWp = (size_t)H * Wp, HW = (size_t)H * W, total = (size_t)B * HWp;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HWp);
    const int r = (int)(idx - (size_t)b * HWp);
    const int y = r / Wp, x = r - y * Wp;
    float G[8], S = 0.f;
    if (x < W) {
        const float* gb = g + (size_t)b * 8 * HW;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int yy = y + dy2(k), xx = x + dx2(k);
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gb[k * HW + (size_t)yy * W + xx];
            if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
            G[k] = v;
            S += fabsf(v);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) wb[((size_t)b * 8 + k) * HWp + r] = x < W ? G[k] / S : 0.f;   // IEEE division: 0/0 = NaN (cspn.py:138)
}

// rows of W floats <-> rows of Wp >= W floats (pad columns zero)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int W, int Wp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * Wp) return;
    const size_t row = idx / Wp;
    const int x = (int)(idx - row * Wp);
    dst[idx] = x < W ? src[row * W + x] : 0.f;
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int W, int Wp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * W) return;
    const size_t row = idx / W;
    const int x = (int)(idx - row * W);
    dst[idx] = src[row * Wp + x];
}

static size_t padded_plane_bytes(int B, int H, int Wp) { return (((size_t)B * H * Wp * sizeof(float)) + 255) & ~(size_t)255; }

bool padded2d_supported(int B, int H, int W, int n_iter) {
    const int Wp = (W + 3) & ~3;
    return (W % 4) != 0 && fused2d_supported(B, H, Wp, n_iter) && (long long)B * H * Wp <= 0x7fffffffLL / 9;
}

size_t padded2d_workspace(int B, int H, int W, int n_iter) {
    const int Wp = (W + 3) & ~3;
    return 11 * padded_plane_bytes(B, H, Wp) + fused2d_workspace(B, H, Wp, n_iter);   // 8 weight planes, blur, sparse, out (+ the passes' ping buffer)
}

int padded2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W, int n_iter, int norm, void* ws,
                     hipStream_t st) {
    const int Wp = (W + 3) & ~3;
    const size_t pb = padded_plane_bytes(B, H, Wp), rows = (size_t)B * H, totp = rows * Wp;
    char* base = (char*)ws;
    float* wbp = (float*)base;
    float* blurp = (float*)(base + 8 * pb);
    float* spp = (float*)(base + 9 * pb);
    float* outp = (float*)(base + 10 * pb);
    void* wsi = base + 11 * pb;
    const bool normalise = norm == CSPN_NORM_8SUM || norm == CSPN_NORM_8SUM_ABS;
    if (normalise) {
        hipLaunchKernelGGL(normalize2d_pitch_kernel, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0, st, g, wbp, B, H, W, Wp, norm);
        if (int e = check_launch("normalize2d_pitch_kernel")) return e;
    } else {
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((8 * totp + 255) / 256)), dim3(256), 0, st, g, wbp, 8 * rows, W, Wp);
    }
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0, st, blur, blurp, rows, W, Wp);
    if (sparse) hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0, st, sparse, spp, rows, W, Wp);
    if (int e = check_launch("pad_rows_kernel")) return e;
    if (int e = fused2d_forward(wbp, blurp, sparse ? spp : nullptr, outp, B, H, Wp, n_iter, normalise ? CSPN_NORM_PRENORM : norm, wsi, st)) return e;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((rows * W + 255) / 256)), dim3(256), 0, st, outp, out, rows, W, Wp);
    return check_launch("unpad_rows_kernel");
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
