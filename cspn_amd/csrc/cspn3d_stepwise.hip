// cspn3d_stepwise.hip -- 3x3x3 propagation (26 neighbours), one launch per iteration.
// API surface: reference cspn_paddle/demo.py:41-43,50-52 (fluid.layers.affinity_propagate,
// kernel source NOT in the reference tree -> parity unpinned; semantics = the 3D
// generalisation of cspn_pytorch/models/cspn.py:42-172, see oracle/cspn_oracle.c).
// Channel order: raster over (f,t,l) in {0,1,2}^3 without (1,1,1); offset (1-f,1-t,1-l).
#include "cspn_common.h"

namespace cspn {

__host__ __device__ constexpr int ch3(int k) { return k < 13 ? k : k + 1; }  // skip the centre (index 13)
__host__ __device__ constexpr int dz3(int k) { return 1 - ch3(k) / 9; }
__host__ __device__ constexpr int dy3(int k) { return 1 - (ch3(k) / 3) % 3; }
__host__ __device__ constexpr int dx3(int k) { return 1 - ch3(k) % 3; }

// wf: [27][B*D*H*W] (26 folded weights + c')
__global__ __launch_bounds__(256) void fold3d_kernel(const float* __restrict__ g, const float* __restrict__ feat,
                                                      const float* __restrict__ sparse, float* __restrict__ wf,
                                                      int B, int D, int H, int W, int norm) {
    const size_t HW = (size_t)H * W, V = (size_t)D * HW, total = (size_t)B * V;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / V);
    const size_t r = idx - (size_t)b * V;
    const int z = (int)(r / HW);
    const int r2 = (int)(r - (size_t)z * HW);
    const int y = r2 / W, x = r2 - y * W;
    const float* gb = g + (size_t)b * 26 * V;
    float G[26], S = 0.f;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        float v;
        if (norm == CSPN_NORM_NONE) {
            v = gb[k * V + r];
        } else {
            const int zz = z + dz3(k), yy = y + dy3(k), xx = x + dx3(k);
            v = 0.f;
            if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
                v = gb[k * V + ((size_t)zz * H + yy) * W + xx];
            if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
        }
        G[k] = v;
        S += fabsf(v);
    }
    float sigma = 0.f;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        if (norm != CSPN_NORM_NONE) G[k] = G[k] / S;
        sigma += G[k];
    }
    const float h0 = feat[idx];
    const float m = sparse ? signf(sparse[idx]) : 0.f;
    const float om = 1.f - m;
    float c = (norm == CSPN_NORM_NONE) ? 0.f : (1.f - sigma) * h0;
    if (sparse) {
        c = om * c + m * h0;
#pragma unroll
        for (int k = 0; k < 26; ++k) G[k] *= om;
    }
#pragma unroll
    for (int k = 0; k < 26; ++k) wf[k * total + idx] = G[k];
    wf[26 * total + idx] = c;
}

__global__ __launch_bounds__(256) void step3d_kernel(const float* __restrict__ wf, const float* __restrict__ hin,
                                                      float* __restrict__ hout, int B, int D, int H, int W) {
    const size_t HW = (size_t)H * W, V = (size_t)D * HW, total = (size_t)B * V;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / V);
    const size_t r = idx - (size_t)b * V;
    const int z = (int)(r / HW);
    const int r2 = (int)(r - (size_t)z * HW);
    const int y = r2 / W, x = r2 - y * W;
    const float* hb = hin + (size_t)b * V;
    float acc = wf[26 * total + idx];
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        const int zz = z + dz3(k), yy = y + dy3(k), xx = x + dx3(k);
        float hv = 0.f;
        if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W) hv = hb[((size_t)zz * H + yy) * W + xx];
        acc = fmaf(wf[k * total + idx], hv, acc);
    }
    hout[idx] = acc;
}

size_t stepwise3d_workspace(int B, int D, int H, int W, int n_iter) {
    (void)n_iter;
    return (27 + 2) * (size_t)B * D * H * W * sizeof(float);
}

int stepwise3d_forward(const float* g, const float* feat, const float* sparse, float* out, int B, int D, int H,
                       int W, int n_iter, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * D * H * W;
    float* wf = (float*)ws;
    float* ping[2] = {wf + 27 * total, wf + 28 * total};
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fold3d_kernel, dim3(blocks), dim3(256), 0, st, g, feat, sparse, wf, B, D, H, W, norm);
    if (int e = check_launch("fold3d_kernel")) return e;
    const float* src = feat;
    for (int it = 0; it < n_iter; ++it) {
        float* dst = (it == n_iter - 1) ? out : ping[it & 1];
        hipLaunchKernelGGL(step3d_kernel, dim3(blocks), dim3(256), 0, st, wf, src, dst, B, D, H, W);
        src = dst;
    }
    return check_launch("step3d_kernel");
}

}  // namespace cspn
