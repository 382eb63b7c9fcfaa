// cspn_aux.hip -- the small steps right next to the propagation path, on the device:
//   * depth metrics + masked L1 loss as ONE fused masked reduction (reference cspn_pytorch/utils.py:19-47 evaluate_error,
//     loss.py:16-23 Wighted_L1_Loss).  The reference copies every prediction to the host and reduces there
//     (train.py:204-206, eval.py:146-150);
//   * Unpool (reference models/torch_resnet_cspn_nyu.py:41-54: conv_transpose2d with a one-hot stride x stride kernel)
//     as the strided scatter it is, and its adjoint.
// All HBM-bound streaming kernels: 8 B/element in for the metrics, (1 + stride^2) * 4 B per input element for Unpool.
#include "cspn_common.h"

namespace cspn {
namespace {

constexpr int NSTAT = 10;  // count, sum d^2, sum d, sum d/gt, 6 threshold counts
constexpr int RB = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// per block: partial sums over a grid-stride range; valid = gt > 1e-4 (utils.py:21, loss.py:17)
__global__ __launch_bounds__(RB) void metrics_partial_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              size_t n, double* __restrict__ partial) {
    float acc[NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) acc[i] = 0.f;
    const float thr[6] = {1.02f, 1.05f, 1.10f, 1.25f, 1.25f * 1.25f, 1.25f * 1.25f * 1.25f};
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const float g = gt[i], p = pred[i];
        if (g > 0.0001f) {
            const float d = fabsf(g - p);
            acc[0] += 1.f;
            acc[1] += d * d;
            acc[2] += d;
            acc[3] += d / g;
            const float r = fmaxf(g / p, p / g);  // utils.py:38-40
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[4 + t] += r < thr[t] ? 1.f : 0.f;
        }
    }
    __shared__ float sm[RB / 64][NSTAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) sm[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NSTAT) {
        double s = 0.0;
        for (int w = 0; w < RB / 64; ++w) s += (double)sm[w][threadIdx.x];
        partial[(size_t)blockIdx.x * NSTAT + threadIdx.x] = s;
    }
}

// out[12]: n_valid, MSE, RMSE, ABS_REL, LG10 (the reference never fills it: 0), MAE, DELTA1.02 .. DELTA1.25^3
__global__ __launch_bounds__(RB) void metrics_final_kernel(const double* __restrict__ partial, int nblocks,
                                                            float* __restrict__ out) {
    __shared__ double sm[RB / 64][NSTAT];
    __shared__ double tot[NSTAT];
    double acc[NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) acc[i] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += RB) {
#pragma unroll
        for (int i = 0; i < NSTAT; ++i) acc[i] += partial[(size_t)b * NSTAT + i];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) sm[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSTAT) {
        double s2 = 0.0;
        for (int w = 0; w < RB / 64; ++w) s2 += sm[w][threadIdx.x];
        tot[threadIdx.x] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nv = tot[0];
        out[0] = (float)nv;
        if (nv > 0.0) {
            out[1] = (float)(tot[1] / nv);
            out[2] = (float)sqrt(tot[1] / nv);
            out[3] = (float)(tot[3] / nv);
            out[4] = 0.f;
            out[5] = (float)(tot[2] / nv);
            for (int t = 0; t < 6; ++t) out[6 + t] = (float)(tot[4 + t] / nv);
        } else {
            for (int i = 1; i < 12; ++i) out[i] = 0.f;  // utils.py:23-26: the dict keeps its zeros
        }
    }
}

// d(loss)/d(pred) of loss = sum_{label > 1e-4} |pred - label| / n_valid  (loss.py:16-23); stats[0] = n_valid
__global__ __launch_bounds__(256) void l1_backward_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                           const float* __restrict__ stats, const float* __restrict__ gscale,
                                                           float* __restrict__ gp, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float l = label[i], d = pred[i] - l;
    const float s = gscale[0] / stats[0];
    gp[i] = l > 0.0001f ? (d > 0.f ? s : (d < 0.f ? -s : 0.f)) : 0.f;
}

// out[nc][y*S + dy][x*S + dx] = (dy == 0 && dx == 0) ? x[nc][y][x] : 0
__global__ __launch_bounds__(256) void unpool_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n_out, int W,
                                                      int S) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int OW = W * S;
    const size_t row = o / OW;           // (nc, oy)
    const int ox = (int)(o - row * OW);
    const size_t ncH = row / S;          // (nc, y) when oy % S == 0
    const int oy_in = (int)(row - ncH * S);
    float v = 0.f;
    if (oy_in == 0 && ox % S == 0) v = x[ncH * W + ox / S];
    out[o] = v;
}

// stride 2, even W: one thread per input pair -> (x0,0,x1,0) on the even output row, zeros on the odd one (16-byte stores)
__global__ __launch_bounds__(256) void unpool2_kernel(const float2* __restrict__ x, float4* __restrict__ out, size_t n_pairs,
                                                       int W2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const size_t row = i / W2;  // (nc, y)
    const int xp = (int)(i - row * W2);
    const float2 v = x[i];
    float4* o = out + (row * 2) * (size_t)W2 + xp;
    o[0] = make_float4(v.x, 0.f, v.y, 0.f);
    o[W2] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void unpool_backward_kernel(const float* __restrict__ go, float* __restrict__ gx, size_t n_in,
                                                               int W, int S) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const size_t ncH = i / W;
    const int xx = (int)(i - ncH * W);
    gx[i] = go[(ncH * S) * ((size_t)W * S) + (size_t)xx * S];
}

int metric_blocks(size_t n) {
    size_t b = (n + (size_t)RB * 8 - 1) / ((size_t)RB * 8);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ---- sparse-depth sampling on the device (reference createSparseDepthImage: nyu_dataset_loader.py:135-144 keeps each pixel
// with probability n_sample / n_pixels, kitti_dataset_loader.py:138-148 with n_sample / n_valid, n_valid = #(depth > 1e-4)
// of that image; sparse = depth * bernoulli(p)).  Counter-based Philox4x32-10: one call yields the four uniforms of four
// consecutive pixels, keyed by (seed, image), so the mask does not depend on the launch geometry.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1;
    c[3] = (unsigned)p0;
    c[0] = n0;
    c[2] = n2;
}

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// per image: number of pixels with depth > 1e-4 (kitti_dataset_loader.py:141), one block per image
__global__ __launch_bounds__(RB) void count_valid_kernel(const float* __restrict__ depth, size_t hw, float* __restrict__ n_valid) {
    const float* d = depth + (size_t)blockIdx.x * hw;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < hw; i += RB) acc += d[i] > 0.0001f ? 1.f : 0.f;   // exact below 2^24 pixels per thread
    acc = wave_sum(acc);
    __shared__ float part[RB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < RB / 64; ++i) t += part[i];
        n_valid[blockIdx.x] = t;
    }
}

// n_valid == nullptr: p = n_sample / hw (NYU loader); otherwise p = n_sample / n_valid[image] (KITTI loader)
__global__ __launch_bounds__(256) void sparse_sample_kernel(const float* __restrict__ depth, float* __restrict__ out, size_t hw,
                                                             size_t quads_per_image, float n_sample,
                                                             const float* __restrict__ n_valid, unsigned seed_lo,
                                                             unsigned seed_hi) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned img = blockIdx.y;
    if (q >= quads_per_image) return;
    const float p = n_valid ? n_sample / n_valid[img] : n_sample / (float)hw;   // p >= 1 keeps every pixel
    unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), img, 0x43535031u};
    philox4x32_10(c, seed_lo, seed_hi);
    const size_t base = (size_t)img * hw + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (4 * q + i >= hw) break;
        const float u = (float)(c[i] >> 8) * (1.0f / 16777216.0f);   // 24 random bits -> [0, 1)
        out[base + i] = u < p ? depth[base + i] : 0.f;
    }
}

}  // namespace
}  // namespace cspn

using namespace cspn;

extern "C" {

size_t cspn_metrics_workspace_bytes(size_t n) { return (size_t)metric_blocks(n) * NSTAT * sizeof(double); }

int cspn_metrics_f32(const float* gt, const float* pred, size_t n, float* out12, void* ws, size_t ws_bytes,
                     cspn_stream_t stream) {
    if (!gt || !pred || !out12) { set_error("null pointer"); return CSPN_E_BADARG; }
    if (!ws || ws_bytes < cspn_metrics_workspace_bytes(n)) { set_error("workspace too small"); return CSPN_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = metric_blocks(n);
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(nb), dim3(RB), 0, st, gt, pred, n, (double*)ws);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(RB), 0, st, (const double*)ws, nb, out12);
    return check_launch("metrics kernels");
}

int cspn_l1_backward_f32(const float* pred, const float* label, const float* stats12, const float* grad_scale,
                         float* grad_pred, size_t n, cspn_stream_t stream) {
    if (!pred || !label || !stats12 || !grad_scale || !grad_pred) { set_error("null pointer"); return CSPN_E_BADARG; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(l1_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, label,
                       stats12, grad_scale, grad_pred, n);
    return check_launch("l1_backward_kernel");
}

int cspn_unpool_f32(const float* x, float* out, size_t NC, int H, int W, int stride, cspn_stream_t stream) {
    if (!x || !out || H <= 0 || W <= 0 || stride < 1) { set_error("bad argument"); return CSPN_E_BADARG; }
    const size_t n_out = NC * (size_t)H * stride * (size_t)W * stride;
    if (n_out == 0) return 0;
    if (stride == 2 && (W % 2) == 0 && ((uintptr_t)x & 7u) == 0 && ((uintptr_t)out & 15u) == 0) {
        const size_t n_pairs = NC * (size_t)H * (W / 2);
        hipLaunchKernelGGL(unpool2_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float2*)x, (float4*)out, n_pairs, W / 2);
        return check_launch("unpool2_kernel");
    }
    hipLaunchKernelGGL(unpool_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n_out, W,
                       stride);
    return check_launch("unpool_kernel");
}

size_t cspn_guidance_head_workspace_bytes(int C) { return C > 0 ? head_workspace(C) : 0; }

int cspn_guidance_head_f32(const float* x, const float* w_guidance, const float* w_blur, float* guidance_out, float* blur_out, int B, int C, int h, int w,
                           int H, int W, int norm_type, void* workspace, size_t workspace_bytes, cspn_stream_t stream) {
    if (!x || !w_guidance || !guidance_out || B < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) { set_error("bad argument"); return CSPN_E_BADARG; }
    if ((w_blur != nullptr) != (blur_out != nullptr)) { set_error("w_blur and blur_out come together"); return CSPN_E_BADARG; }
    if (H > 2 * h || W > 2 * w) { set_error("H x W = %d x %d exceeds the unpooled %d x %d", H, W, 2 * h, 2 * w); return CSPN_E_BADARG; }
    if (norm_type != CSPN_NORM_NONE && norm_type != CSPN_NORM_8SUM && norm_type != CSPN_NORM_8SUM_ABS) { set_error("norm_type must be NONE (raw guidance), 8SUM or 8SUM_ABS (gate_wb)"); return CSPN_E_BADARG; }
    if (!workspace || workspace_bytes < head_workspace(C) || ((uintptr_t)workspace & 7u) != 0) { set_error("workspace: need %zu bytes, 8-byte aligned", head_workspace(C)); return CSPN_E_BADARG; }
    if ((long long)B * 8 * H * W >= (1ll << 40) || (long long)C * h * w >= (1ll << 31)) { set_error("tensor too large"); return CSPN_E_UNSUPPORTED; }
    if (B == 0) return 0;
    const int mode = norm_type == CSPN_NORM_NONE ? 0 : (norm_type == CSPN_NORM_8SUM ? 1 : 2);
    return head_forward(x, w_guidance, w_blur, guidance_out, blur_out, B, C, h, w, H, W, mode, workspace, (hipStream_t)stream);
}

size_t cspn_guidance_head_backward_workspace_bytes(int B, int C, int h, int w) { return (B > 0 && C > 0 && h > 0 && w > 0) ? head_backward_workspace(B, C, h, w) : 0; }

int cspn_guidance_head_backward_f32(const float* x, const float* w_guidance, const float* w_blur, const float* grad_guidance, const float* grad_blur,
                                    float* grad_x, float* grad_w_guidance, float* grad_w_blur, int B, int C, int h, int w, int H, int W, void* workspace,
                                    size_t workspace_bytes, cspn_stream_t stream) {
    if (!x || !w_guidance || !grad_guidance || B < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) { set_error("bad argument"); return CSPN_E_BADARG; }
    if ((w_blur != nullptr) != (grad_blur != nullptr)) { set_error("w_blur and grad_blur come together"); return CSPN_E_BADARG; }
    if (grad_w_blur && !w_blur) { set_error("grad_w_blur without a blur head"); return CSPN_E_BADARG; }
    if (H > 2 * h || W > 2 * w) { set_error("H x W = %d x %d exceeds the unpooled %d x %d", H, W, 2 * h, 2 * w); return CSPN_E_BADARG; }
    if (B == 0 || (!grad_x && !grad_w_guidance && !grad_w_blur)) return 0;
    const size_t need = head_backward_workspace(B, C, h, w);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255u) != 0) { set_error("workspace: need %zu bytes, 256-byte aligned", need); return CSPN_E_WORKSPACE; }
    if ((long long)B * 8 * H * W >= (1ll << 40) || (long long)C * h * w >= (1ll << 31)) { set_error("tensor too large"); return CSPN_E_UNSUPPORTED; }
    return head_backward(x, w_guidance, w_blur, grad_guidance, grad_blur, grad_x, grad_w_guidance, grad_w_blur, B, C, h, w, H, W, workspace, (hipStream_t)stream);
}

int cspn_unpool_backward_f32(const float* grad_out, float* grad_x, size_t NC, int H, int W, int stride, cspn_stream_t stream) {
    if (!grad_out || !grad_x || H <= 0 || W <= 0 || stride < 1) { set_error("bad argument"); return CSPN_E_BADARG; }
    const size_t n_in = NC * (size_t)H * W;
    if (n_in == 0) return 0;
    hipLaunchKernelGGL(unpool_backward_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                       grad_x, n_in, W, stride);
    return check_launch("unpool_backward_kernel");
}

size_t cspn_sparse_sample_workspace_bytes(size_t n_images) { return n_images * sizeof(float); }

// depth, sparse_out: [n_images][hw]; mode 0: keep probability n_sample / hw (reference nyu_dataset_loader.py:135-144),
// mode 1: n_sample / (number of pixels of that image with depth > 1e-4) (kitti_dataset_loader.py:138-148)
int cspn_sparse_sample_f32(const float* depth, float* sparse_out, size_t n_images, size_t hw, int n_sample, int mode,
                           unsigned long long seed, void* ws, size_t ws_bytes, cspn_stream_t stream) {
    if (!depth || !sparse_out || hw == 0 || n_sample < 0 || (mode != 0 && mode != 1)) { set_error("bad argument"); return CSPN_E_BADARG; }
    if (n_images == 0) return 0;
    if (n_images > 65535) { set_error("at most 65535 images per call"); return CSPN_E_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    float* n_valid = nullptr;
    if (mode == 1) {
        if (!ws || ws_bytes < cspn_sparse_sample_workspace_bytes(n_images)) { set_error("workspace too small"); return CSPN_E_WORKSPACE; }
        n_valid = (float*)ws;
        hipLaunchKernelGGL(count_valid_kernel, dim3((unsigned)n_images), dim3(RB), 0, st, depth, hw, n_valid);
    }
    const size_t quads = (hw + 3) / 4;
    hipLaunchKernelGGL(sparse_sample_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)n_images), dim3(256), 0, st, depth,
                       sparse_out, hw, quads, (float)n_sample, n_valid, (unsigned)seed, (unsigned)(seed >> 32));
    return check_launch("sparse_sample_kernel");
}

}  // extern "C"
