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

// The Paddle contract (norm_type NONE, no sparse): gates are used as given, centre-sited, no centre term
// (reference cspn_paddle/README.md:54-56, demo.py:41-52).  One iteration then needs exactly the algorithmic traffic of a
// single propagation step -- 26 gates + value in, value out = 112 B/voxel -- so it reads the gate tensor directly: no
// fold pass, no coefficient planes.  4 voxels per thread along x (16-byte loads of the gate planes).
__global__ __launch_bounds__(256) void step3d_direct_kernel(const float* __restrict__ g, const float* __restrict__ hin,
                                                             float* __restrict__ hout, int B, int D, int H, int W4) {
    const int W = 4 * W4;
    const size_t HW = (size_t)H * W, V = (size_t)D * HW, total4 = (size_t)B * D * H * W4;
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const size_t idx = 4 * i4;
    const int b = (int)(idx / V);
    const size_t r = idx - (size_t)b * V;
    const int z = (int)(r / HW);
    const int r2 = (int)(r - (size_t)z * HW);
    const int y = r2 / W, x = r2 - y * W;
    const float* hb = hin + (size_t)b * V;
    const float* gb = g + (size_t)b * 26 * V + r;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < 9; ++n) {  // the 9 neighbour rows (dz, dy); three x-taps each
        const int dz = 1 - n / 3, dy = 1 - n % 3;
        const int zz = z + dz, yy = y + dy;
        float hm = 0.f, hp = 0.f;
        float4 hc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (zz >= 0 && zz < D && yy >= 0 && yy < H) {
            const float* row = hb + ((size_t)zz * H + yy) * W;
            hc = *reinterpret_cast<const float4*>(row + x);
            if (x > 0) hm = row[x - 1];
            if (x + 4 < W) hp = row[x + 4];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {  // dx = 1 - t
            const int c27 = n * 3 + t;   // raster index (f,t,l) with f = n/3, t = n%3, l = t
            if (c27 == 13) continue;     // the centre has no gate
            const int k = c27 < 13 ? c27 : c27 - 1;
            const float4 w = *reinterpret_cast<const float4*>(gb + (size_t)k * V);
            const int dx = 1 - t;
            const float h0 = dx > 0 ? hc.y : (dx < 0 ? hm : hc.x);
            const float h1 = dx > 0 ? hc.z : (dx < 0 ? hc.x : hc.y);
            const float h2 = dx > 0 ? hc.w : (dx < 0 ? hc.y : hc.z);
            const float h3 = dx > 0 ? hp : (dx < 0 ? hc.z : hc.w);
            acc.x = fmaf(w.x, h0, acc.x);
            acc.y = fmaf(w.y, h1, acc.y);
            acc.z = fmaf(w.z, h2, acc.z);
            acc.w = fmaf(w.w, h3, acc.w);
        }
    }
    *reinterpret_cast<float4*>(hout + idx) = acc;
}

// one step of the Paddle contract for other files (the backward keeps the value levels): W % 4 == 0, 16-byte aligned tensors
int step3d_direct(const float* g, const float* hin, float* hout, int B, int D, int H, int W, hipStream_t st) {
    const size_t total = (size_t)B * D * H * W;
    hipLaunchKernelGGL(step3d_direct_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, g, hin, hout, B, D, H, W / 4);
    return 0;
}

static bool direct3d_ok(const float* g, const float* feat, const float* sparse, const float* out, int W, int norm,
                        const void* ws) {
    return norm == CSPN_NORM_NONE && sparse == nullptr && (W % 4) == 0 &&
           (((uintptr_t)g | (uintptr_t)feat | (uintptr_t)out | (uintptr_t)ws) & 15u) == 0;
}

size_t stepwise3d_workspace(int B, int D, int H, int W, int n_iter) {
    (void)n_iter;
    // large enough for every mode: the 27 fold planes + what the persistent kernel wants (two value volumes = the ping-pong of
    // the per-step path, its exchange buffers)
    return 27 * (size_t)B * D * H * W * sizeof(float) + persistent3d_workspace(B, D, H, W);
}

// what the chosen path really needs: the Paddle contract (gates used as given, no mask) keeps two value volumes (+ the sync
// words of the persistent kernel), the folding modes 27 coefficient planes more
size_t forward3d_workspace(int B, int D, int H, int W, int n_iter, int norm, bool has_sparse) {
    (void)n_iter;
    if (norm == CSPN_NORM_NONE && !has_sparse && (W % 4) == 0) return persistent3d_workspace(B, D, H, W);
    return stepwise3d_workspace(B, D, H, W, n_iter);
}

// algo: 0 auto, 1 one launch per step, 2 persistent (gates resident across steps)
int stepwise3d_forward(const float* g, const float* feat, const float* sparse, float* out, int B, int D, int H,
                       int W, int n_iter, int norm, void* ws, hipStream_t st, int algo) {
    const size_t total = (size_t)B * D * H * W;
    float* wf = (float*)ws;
    const bool direct = direct3d_ok(g, feat, sparse, out, W, norm, ws);
    if (algo == 2 && direct && !persistent3d_supported(B, D, H, W, n_iter)) {
        set_error("persistent 3D kernel does not take this call (needs W %% 4 == 0, 16-byte aligned tensors, 2 <= n_iter <= 60, a chunk per device)");
        return CSPN_E_UNSUPPORTED;
    }
    if (direct && algo != 1 && persistent3d_supported(B, D, H, W, n_iter))
        return persistent3d_forward(g, feat, out, B, D, H, W, n_iter, ws, st);
    if (direct) {
        float* pp[2] = {wf, wf + total};
        const unsigned blocks4 = (unsigned)((total / 4 + 255) / 256);
        const float* src = feat;
        for (int it = 0; it < n_iter; ++it) {
            float* dst = (it == n_iter - 1) ? out : pp[it & 1];
            hipLaunchKernelGGL(step3d_direct_kernel, dim3(blocks4), dim3(256), 0, st, g, src, dst, B, D, H, W / 4);
            src = dst;
        }
        return check_launch("step3d_direct_kernel");
    }
    float* ping[2] = {wf + 27 * total, wf + 28 * total};
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fold3d_kernel, dim3(blocks), dim3(256), 0, st, g, feat, sparse, wf, B, D, H, W, norm);
    if (int e = check_launch("fold3d_kernel")) return e;
    // the normalising / masked modes fused: H_{t+1} = c' + sum_k w'_k H_t(p + off_k) with the folded planes resident in the
    // persistent kernel's registers (c' in LDS): fold once, then one pass over the 27 planes for all steps
    const bool aligned = ((((uintptr_t)feat | (uintptr_t)out | (uintptr_t)ws) & 15u) == 0) && (W % 4) == 0 && (total % 4) == 0;
    if (algo == 2 && !(aligned && persistent3d_supported(B, D, H, W, n_iter))) {
        set_error("persistent 3D kernel does not take this call (needs W %% 4 == 0, 16-byte aligned tensors, 2 <= n_iter <= 60, a chunk per device)");
        return CSPN_E_UNSUPPORTED;
    }
    if (algo != 1 && aligned && persistent3d_supported(B, D, H, W, n_iter))
        return persistent3d_forward_folded(wf, feat, out, B, D, H, W, n_iter, wf + 27 * total, st);
    const float* src = feat;
    for (int it = 0; it < n_iter; ++it) {
        float* dst = (it == n_iter - 1) ? out : ping[it & 1];
        hipLaunchKernelGGL(step3d_kernel, dim3(blocks), dim3(256), 0, st, wf, src, dst, B, D, H, W);
        src = dst;
    }
    return check_launch("step3d_kernel");
}

}  // namespace cspn
