// cspn2d_backward.hip -- gradient of Affinity_Propagate.forward (reference cspn_pytorch/models/cspn.py:42-83) with respect
// to guidance and blur_depth: what torch autograd computes when reference train.py:196-198 back-propagates through the
// module.
//
// Forward, folded (cspn2d_stepwise.hip):  H_{t+1} = c' + sum_k w'_k * shift_k(H_t),  w'_k = (1-m) w_k,
//   c' = (1-m)(1-sigma) H_0 + m H_0,  w_k = G_k / S,  S = sum_j |G_j|,  G_k(p) = g~_k(p + off_k),  sigma = sum_k w_k.
// Adjoint:  A_N = dL/dout,   A_t(p) = sum_k (w'_k A_{t+1})(p - off_k)          (bwd_step_kernel, N launches)
//           dW'_k(p) = sum_t A_{t+1}(p) H_t(p + off_k),   dC(p) = sum_t A_{t+1}(p)   (bwd_final_kernel, from the two histories)
//           dL/dw_k = (1-m)(dW'_k - dC H_0);   dL/dH_0 = A_0 + dC ((1-m)(1-sigma) + m)
//           dL/dG_k = dL/dw_k / S - sign(G_k) (sum_j dL/dw_j G_j) / S^2          (torch: d|x|/dx = sign(x), 0 at 0)
//           dL/dg_k(p + off_k) = dL/dG_k(p)  [* sign(g) for '8sum_abs'];  elements no pixel reads get 0.
// CSPN_NORM_PRENORM (round 6): `guidance` is the reference's gate_wb (cspn.py:85-144), w_k(p) = wb_k(p) at the pixel itself: the chain ends at
//           dL/dwb_k = (1-m)(dW'_k - dC H_0),   dL/dH_0 = A_0 + dC ((1-m)(1 - sum_k wb_k) + m).
// 24-iteration passes on images the ring kernel takes: two sweeps of that kernel (forward keeping H_4, H_8 .. H_20 and the folded
// coefficients, adjoint keeping A_20 .. A_4) + bwd_final_mx_kernel, which recomputes the levels in between tile by tile.
// Everything else: fold + one launch per step for both recursions (every level kept) + bwd_final_kernel.
#include <cstdlib>
#include <type_traits>

#include "cspn_common.h"

namespace cspn {

// from cspn2d_stepwise.hip
__global__ void fold2d_kernel(const float* __restrict__ g, const float* __restrict__ blur, const float* __restrict__ sparse,
                              float* __restrict__ wf, int B, int H, int W, int norm);
__global__ void step2d_kernel(const float* __restrict__ wf, const float* __restrict__ hin, float* __restrict__ hout, int B,
                              int H, int W);

namespace {

// wt_k(p) = w'_k(p - off_k) (0 outside): the adjoint stencil then reads its eight coefficient planes at p itself, like
// the forward step does
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ wf, float* __restrict__ wt, int B, int H,
                                                           int W) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y - dy2(k), xx = x - dx2(k);
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = wf[k * total + (size_t)b * HW + (size_t)yy * W + xx];
        wt[k * total + idx] = v;
    }
}

// A_t(p) = sum_k wt_k(p) A_{t+1}(p - off_k)
__global__ __launch_bounds__(256) void bwd_step_kernel(const float* __restrict__ wt, const float* __restrict__ ain,
                                                        float* __restrict__ aout, int B, int H, int W) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    const size_t base = (size_t)b * HW;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y - dy2(k), xx = x - dx2(k);
        float a = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) a = ain[base + (size_t)yy * W + xx];
        acc = fmaf(wt[k * total + idx], a, acc);
    }
    aout[idx] = acc;
}

// final pass of the stepwise sweeps (every level kept): hh = H_1 .. H_{N-1} (H_0 = blur), ah = A_0 .. A_{N-1}
__global__ __launch_bounds__(256) void bwd_final_kernel(const float* __restrict__ g, const float* __restrict__ blur,
                                                         const float* __restrict__ sparse, const float* __restrict__ hh,
                                                         const float* __restrict__ ah,
                                                         const float* __restrict__ gout,
                                                         float* __restrict__ gg, float* __restrict__ gb, int B, int H, int W,
                                                         int n_iter, int norm) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    const size_t base = (size_t)b * HW;
    int noff[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y + dy2(k), xx = x + dx2(k);
        ok[k] = yy >= 0 && yy < H && xx >= 0 && xx < W;
        noff[k] = ok[k] ? yy * W + xx : r;
    }
    float dW[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dC = 0.f;
    for (int t = 0; t < n_iter; ++t) {
        float a;
        if (t + 1 == n_iter) a = gout[idx];
        else a = ah[(size_t)(t + 1) * total + idx];
        const float* ht = (t == 0) ? blur : hh + (size_t)(t - 1) * total;
        dC += a;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float hv = 0.f;
            if (ok[k]) hv = ht[base + noff[k]];
            dW[k] = fmaf(a, hv, dW[k]);
        }
    }
    const float h0 = blur[idx];
    const float m = sparse ? signf(sparse[idx]) : 0.f;
    const float om = 1.f - m;
    const float a0 = ah[idx];
    const float* gbp = g + (size_t)b * 8 * HW;
    if (norm == CSPN_NORM_NONE) {  // gates used as given, centre-sited, no centre term: c' = m H_0
        if (gb) gb[idx] = a0 + dC * m;
        if (gg) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gg[(size_t)b * 8 * HW + k * HW + r] = om * dW[k];
        }
        return;
    }
    if (norm == CSPN_NORM_PRENORM) {  // the planes are the reference's gate_wb, read at the pixel itself: w'_k = (1-m) wb_k, c' = ((1-m)(1-sigma) + m) H_0
        float sigma = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sigma += gbp[k * HW + r];
        if (gb) gb[idx] = a0 + dC * (om * (1.f - sigma) + m);
        if (gg) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gg[(size_t)b * 8 * HW + k * HW + r] = om * (dW[k] - dC * h0);
        }
        return;
    }
    float G[8], S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = ok[k] ? gbp[k * HW + noff[k]] : 0.f;
        if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
        G[k] = v;
        S += fabsf(v);
    }
    float sigma = 0.f, T1 = 0.f, dw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sigma += G[k] / S;
        dw[k] = om * (dW[k] - dC * h0);
        T1 = fmaf(dw[k], G[k], T1);
    }
    if (gb) gb[idx] = a0 + dC * (om * (1.f - sigma) + m);
    if (gg) {
        // g_k(q) with q - off_k outside the image is read by no pixel (the gather sees the zero padding instead): gradient 0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ys = y - dy2(k), xs = x - dx2(k);
            if (ys < 0 || ys >= H || xs < 0 || xs >= W) gg[(size_t)b * 8 * HW + k * HW + r] = 0.f;
        }
        const float t2 = T1 / (S * S);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!ok[k]) continue;  // the zero padding is a constant
            const float sg = G[k] > 0.f ? 1.f : (G[k] < 0.f ? -1.f : 0.f);
            float d = dw[k] / S - sg * t2;
            if (norm == CSPN_NORM_8SUM_ABS) {
                const float raw = gbp[k * HW + noff[k]];
                d *= raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
            }
            gg[(size_t)b * 8 * HW + k * HW + noff[k]] = d;  // g_k(p + off_k) is read by pixel p only
        }
    }
}


// ---- helpers of the final pass of the assembly sweeps: one group of 4 columns per thread ------------------------------
__device__ __forceinline__ float dpp_shr1(float v) {   // within each row of 16 lanes: lane i <- lane i-1 (first lane: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
}
__device__ __forceinline__ float dpp_shl1(float v) {   // within each row of 16 lanes: lane i <- lane i+1 (last lane: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));   // row_shl:1
}
__device__ __forceinline__ float4 ld4u(const float* p) {   // 16 bytes, 4-byte aligned
    float4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void st4u(float* p, float4 v) { __builtin_memcpy(p, &v, 16); }

// ---- the end of the final pass for one group of 4 columns (b, y, x .. x + 3): from dW'_k and dC through the fold,
// the normalisation and the neighbour-sited gather to dL/dguidance and dL/dblur_depth (see the file header)
// INTERIOR: every run (row y - 1 .. y + 1, columns x - 1 .. x + 4) lies inside the image -- no bounds checks, no border zero fill
template <bool INTERIOR>
__device__ __forceinline__ void bwd_epilogue4(const float* __restrict__ g, const float* __restrict__ blur, const float* __restrict__ sparse,
                                              const float* __restrict__ a0p, float* __restrict__ gg, float* __restrict__ gb, int b, int y,
                                              int x, size_t idx, size_t HW, int H, int W, int norm, const float (&dW)[8][4],
                                              const float (&dC)[4]) {
    // ---- epilogue: the chain through the fold, the normalisation and the neighbour-sited gather (see the file header)
    const float4 h0q = *reinterpret_cast<const float4*>(blur + idx);
    const float h0[4] = {h0q.x, h0q.y, h0q.z, h0q.w};
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    if (sparse) {
        const float4 sq = *reinterpret_cast<const float4*>(sparse + idx);
        m[0] = signf(sq.x); m[1] = signf(sq.y); m[2] = signf(sq.z); m[3] = signf(sq.w);
    }
    const float4 a0q = *reinterpret_cast<const float4*>(a0p + idx);
    const float a0[4] = {a0q.x, a0q.y, a0q.z, a0q.w};
    const float* gbp = g + (size_t)b * 8 * HW;
    float* ggp = gg ? gg + (size_t)b * 8 * HW : nullptr;
    if (norm == CSPN_NORM_NONE) {  // gates used as given, centre-sited, no centre term: c' = m H_0
        if (gb) *reinterpret_cast<float4*>(gb + idx) = make_float4(a0[0] + dC[0] * m[0], a0[1] + dC[1] * m[1], a0[2] + dC[2] * m[2],
                                                                    a0[3] + dC[3] * m[3]);
        if (ggp) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4*>(ggp + k * HW + (size_t)y * W + x) =
                    make_float4((1.f - m[0]) * dW[k][0], (1.f - m[1]) * dW[k][1], (1.f - m[2]) * dW[k][2], (1.f - m[3]) * dW[k][3]);
        }
        return;
    }
    if (norm == CSPN_NORM_PRENORM) {  // gate_wb as given, at the pixel itself (see bwd_final_kernel): no normalisation chain, no scatter
        float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(gbp + k * HW + (size_t)y * W + x);
            sg[0] += q.x; sg[1] += q.y; sg[2] += q.z; sg[3] += q.w;
        }
        float om[4], ch[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { om[i] = 1.f - m[i]; ch[i] = dC[i] * h0[i]; }
        if (gb) *reinterpret_cast<float4*>(gb + idx) = make_float4(a0[0] + dC[0] * (om[0] * (1.f - sg[0]) + m[0]), a0[1] + dC[1] * (om[1] * (1.f - sg[1]) + m[1]),
                                                                    a0[2] + dC[2] * (om[2] * (1.f - sg[2]) + m[2]), a0[3] + dC[3] * (om[3] * (1.f - sg[3]) + m[3]));
        if (ggp) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4*>(ggp + k * HW + (size_t)y * W + x) =
                    make_float4(om[0] * (dW[k][0] - ch[0]), om[1] * (dW[k][1] - ch[1]), om[2] * (dW[k][2] - ch[2]), om[3] * (dW[k][3] - ch[3]));
        }
        return;
    }
    // G_k(p) = g~_k(p + off_k): a run of four columns of plane k in row y + dy_k starting at x + dx_k, zero outside the image; the
    // eight runs stay in registers for the second half (the coefficient registers of the level loop are free by now)
    auto run = [&](int k, float (&v)[4]) -> bool {
        const int yy = y + dy2(k), xs = x + dx2(k);
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (!INTERIOR && (yy < 0 || yy >= H)) return false;
        const float* src = gbp + k * HW + (size_t)yy * W;
        if (INTERIOR || (xs >= 0 && xs + 3 < W)) {
            const float4 q = ld4u(src + xs);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xs + i >= 0 && xs + i < W) v[i] = src[xs + i];
        }
        return true;
    };
    float om[4], ch[4], S[4] = {0.f, 0.f, 0.f, 0.f}, T1[4] = {0.f, 0.f, 0.f, 0.f}, gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { om[i] = 1.f - m[i]; ch[i] = dC[i] * h0[i]; }
    float vk[8][4];
    bool rowin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        rowin[k] = run(k, vk[k]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float G = norm == CSPN_NORM_8SUM_ABS ? fabsf(vk[k][i]) : vk[k][i];
            S[i] += fabsf(vk[k][i]);
            gs[i] += G;
            T1[i] = fmaf(om[i] * (dW[k][i] - ch[i]), G, T1[i]);
        }
    }
    float rS[4], t2[4];
#pragma unroll
#ifdef BWD_IEEE_DIV   // (accuracy A/B, tools/build_bwdvar.sh: no difference against a float64 oracle, profiles/r03_fuzz_parity.txt)
    for (int i = 0; i < 4; ++i) { rS[i] = 1.f / S[i]; t2[i] = T1[i] / (S[i] * S[i]); }
#else
    for (int i = 0; i < 4; ++i) { rS[i] = __builtin_amdgcn_rcpf(S[i]); t2[i] = T1[i] * rS[i] * rS[i]; }   // (v_rcp_f32: 1 ulp; three IEEE divisions per pixel were ~150 instructions per thread)
#endif
    if (gb) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = a0[i] + dC[i] * (om[i] * (1.f - gs[i] * rS[i]) + m[i]);
        *reinterpret_cast<float4*>(gb + idx) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (ggp) {
        // g_k(q) with q - off_k outside the image is read by no pixel (the gather sees the zero padding instead): gradient 0
        // (only groups on the image's border have such elements)
        if (!INTERIOR && (y == 0 || y == H - 1 || x == 0 || x + 4 >= W)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ys = y - dy2(k);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xq = x + i - dx2(k);
                    if (ys < 0 || ys >= H || xq < 0 || xq >= W) ggp[k * HW + (size_t)y * W + x + i] = 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float (&v)[4] = vk[k];
            if (!rowin[k]) continue;  // the zero padding is a constant
            const int yy = y + dy2(k), xs = x + dx2(k);
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float G = norm == CSPN_NORM_8SUM_ABS ? fabsf(v[i]) : v[i];
                const float sg = G > 0.f ? 1.f : (G < 0.f ? -1.f : 0.f);
                float r = om[i] * (dW[k][i] - ch[i]) * rS[i] - sg * t2[i];
                if (norm == CSPN_NORM_8SUM_ABS) r *= v[i] > 0.f ? 1.f : (v[i] < 0.f ? -1.f : 0.f);
                d[i] = r;
            }
            float* dst = ggp + k * HW + (size_t)yy * W;   // g_k(p + off_k) is read by pixel p only
            if (INTERIOR || (xs >= 0 && xs + 3 < W)) st4u(dst + xs, make_float4(d[0], d[1], d[2], d[3]));
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xs + i >= 0 && xs + i < W) dst[xs + i] = d[i];
            }
        }
    }
}

// The same epilogue in TWO passes over the eight guidance runs, for the 64-row final pass (round 5): the first pass only sums (S = sum |G|,
// sum G, T1 = sum dL/dw_k G_k), the second loads each run again (it is in L2) and writes its gradient -- the runs are never all live, so the
// kernel fits 128 registers and four waves per SIMD.  Same arithmetic, same order of the sums as bwd_epilogue4.
template <bool INTERIOR, class GetDW>
__device__ __attribute__((noinline)) void bwd_epilogue4_2pass(const float* __restrict__ g, const float* __restrict__ blur, const float* __restrict__ sparse,
                                                    const float* __restrict__ a0p, float* __restrict__ gg, float* __restrict__ gb, int b, int y,
                                                    int x, size_t idx, size_t HW, int H, int W, int norm, GetDW get_dw /* k -> dW'_k of the 4 pixels */,
                                                    const float (&dC)[4]) {
    const float4 h0q = *reinterpret_cast<const float4*>(blur + idx);
    const float h0[4] = {h0q.x, h0q.y, h0q.z, h0q.w};
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    if (sparse) {
        const float4 sq = *reinterpret_cast<const float4*>(sparse + idx);
        m[0] = signf(sq.x); m[1] = signf(sq.y); m[2] = signf(sq.z); m[3] = signf(sq.w);
    }
    const float* gbp = g + (size_t)b * 8 * HW;
    float* ggp = gg ? gg + (size_t)b * 8 * HW : nullptr;
    if (norm == CSPN_NORM_NONE) {  // gates used as given, centre-sited, no centre term: c' = m H_0
        if (gb) {
            const float4 a0q = *reinterpret_cast<const float4*>(a0p + idx);
            *reinterpret_cast<float4*>(gb + idx) = make_float4(a0q.x + dC[0] * m[0], a0q.y + dC[1] * m[1], a0q.z + dC[2] * m[2], a0q.w + dC[3] * m[3]);
        }
        if (ggp) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 w = get_dw(k);
                *reinterpret_cast<float4*>(ggp + k * HW + (size_t)y * W + x) =
                    make_float4((1.f - m[0]) * w.x, (1.f - m[1]) * w.y, (1.f - m[2]) * w.z, (1.f - m[3]) * w.w);
            }
        }
        return;
    }
    if (norm == CSPN_NORM_PRENORM) {  // gate_wb as given, at the pixel itself (see bwd_final_kernel)
        float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(gbp + k * HW + (size_t)y * W + x);
            sg[0] += q.x; sg[1] += q.y; sg[2] += q.z; sg[3] += q.w;
        }
        float om[4], ch[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { om[i] = 1.f - m[i]; ch[i] = dC[i] * h0[i]; }
        if (gb) {
            const float4 a0q = *reinterpret_cast<const float4*>(a0p + idx);
            *reinterpret_cast<float4*>(gb + idx) = make_float4(a0q.x + dC[0] * (om[0] * (1.f - sg[0]) + m[0]), a0q.y + dC[1] * (om[1] * (1.f - sg[1]) + m[1]),
                                                                a0q.z + dC[2] * (om[2] * (1.f - sg[2]) + m[2]), a0q.w + dC[3] * (om[3] * (1.f - sg[3]) + m[3]));
        }
        if (ggp) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 w = get_dw(k);
                *reinterpret_cast<float4*>(ggp + k * HW + (size_t)y * W + x) =
                    make_float4(om[0] * (w.x - ch[0]), om[1] * (w.y - ch[1]), om[2] * (w.z - ch[2]), om[3] * (w.w - ch[3]));
            }
        }
        return;
    }
    auto run = [&](int k, float (&v)[4]) -> bool {
        const int yy = y + dy2(k), xs = x + dx2(k);
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (!INTERIOR && (yy < 0 || yy >= H)) return false;
        const float* src = gbp + k * HW + (size_t)yy * W;
        if (INTERIOR || (xs >= 0 && xs + 3 < W)) {
            const float4 q = ld4u(src + xs);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xs + i >= 0 && xs + i < W) v[i] = src[xs + i];
        }
        return true;
    };
    float om[4], ch[4], S[4] = {0.f, 0.f, 0.f, 0.f}, T1[4] = {0.f, 0.f, 0.f, 0.f}, gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { om[i] = 1.f - m[i]; ch[i] = dC[i] * h0[i]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // pass 1: a run is consumed at once
        float v[4];
        run(k, v);
        const float4 wq = get_dw(k);
        const float dWk[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float G = norm == CSPN_NORM_8SUM_ABS ? fabsf(v[i]) : v[i];
            S[i] += fabsf(v[i]);
            gs[i] += G;
            T1[i] = fmaf(om[i] * (dWk[i] - ch[i]), G, T1[i]);
        }
    }
    float rS[4], t2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rS[i] = __builtin_amdgcn_rcpf(S[i]); t2[i] = T1[i] * rS[i] * rS[i]; }
    if (gb) {
        const float4 a0q = *reinterpret_cast<const float4*>(a0p + idx);
        const float a0[4] = {a0q.x, a0q.y, a0q.z, a0q.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = a0[i] + dC[i] * (om[i] * (1.f - gs[i] * rS[i]) + m[i]);
        *reinterpret_cast<float4*>(gb + idx) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (ggp) {
        if (!INTERIOR && (y == 0 || y == H - 1 || x == 0 || x + 4 >= W)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ys = y - dy2(k);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xq = x + i - dx2(k);
                    if (ys < 0 || ys >= H || xq < 0 || xq >= W) ggp[k * HW + (size_t)y * W + x + i] = 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // pass 2
            float v[4];
            if (!run(k, v)) continue;  // the zero padding is a constant
            const int yy = y + dy2(k), xs = x + dx2(k);
            const float4 wq = get_dw(k);
            const float dWk[4] = {wq.x, wq.y, wq.z, wq.w};
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float G = norm == CSPN_NORM_8SUM_ABS ? fabsf(v[i]) : v[i];
                const float sg = G > 0.f ? 1.f : (G < 0.f ? -1.f : 0.f);
                float r = om[i] * (dWk[i] - ch[i]) * rS[i] - sg * t2[i];
                if (norm == CSPN_NORM_8SUM_ABS) r *= v[i] > 0.f ? 1.f : (v[i] < 0.f ? -1.f : 0.f);
                d[i] = r;
            }
            float* dst = ggp + k * HW + (size_t)yy * W;   // g_k(p + off_k) is read by pixel p only
            if (INTERIOR || (xs >= 0 && xs + 3 < W)) st4u(dst + xs, make_float4(d[0], d[1], d[2], d[3]));
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xs + i >= 0 && xs + i < W) dst[xs + i] = d[i];
            }
        }
    }
}

// ---- final pass from CHECKPOINTS (round 3): the sweeps keep every fourth level only ----------------------------------------
// The forward sweep stores H_4, H_8 .. H_20, the adjoint sweep A_20, A_16 .. A_4 (generator option hist_every; H_0 = blur and
// A_24 = dL/dout are inputs): 10 level planes through HBM instead of 46.  This pass recomputes the three levels in between,
// tile by tile: a block of 512 threads owns a REGION of 32 rows x 16 groups of 4 columns and produces the TILE that is left
// 4 pixels inside it (24 x 56); thread = one group (a wave = 4 region rows of 16 lanes, so the columns beside a group come
// from the neighbouring lane of the 16-lane DPP row, the rows above / below through LDS).  Per segment s = 0, 4 .. 20:
//   H_{s+1..s+3} = c' + sum_k w'_k shift_k(H)            (pull form; valid one pixel further inside the region per step)
//   for t = s+3 .. s:  dW'_k += A_{t+1} H_t(. + off_k), dC += A_{t+1};  A_t(q) = sum_k (w'_k A_{t+1})(q - off_k)
// the adjoint step in PUSH form, so that both recurrences use the thread's own, centre-sited w'_k(p) -- 32 registers loaded
// once per tile, no coefficient traffic per step: p sends P_k = w'_k(p) A_{t+1}(p) to q = p + off_k; what goes to the row
// below / above is summed over its three columns first (T_dn / T_up, one quad each through LDS), the in-row part stays in
// the lanes.  c' is not stored: c' = H_0 (1 - sum_k w'_k) for the normalising modes (any sparse sign), m H_0 for norm NONE.
// Outside the image everything is exactly 0, as in the forward's zero padding.
#if (defined(BWD_EXP_NOEPI) || defined(BWD_EXP_NOLOOP) || defined(BWD_EXP_NOBAR)) && !defined(BWD_EXPERIMENT_BUILD)
#error "BWD_EXP_* switch timing variants that give WRONG RESULTS: tools/build_bwdvar.sh defines BWD_EXPERIMENT_BUILD for them"
#endif
constexpr int CK = 4, CK_GR = 16, CK_TG = CK_GR - 2, NCKP = 24 / CK - 1;   // NCKP: level planes a sweep keeps

[[maybe_unused]] __device__ __forceinline__ float4 reg_to_img(float4 q) { return make_float4(q.x, q.z, q.w, q.y); }   // (c0,c3,c1,c2) -> (c0..c3)
typedef float v2f __attribute__((ext_vector_type(2)));

// workgroup barrier for LDS traffic only: __syncthreads() also waits for every global load in flight (vmcnt 0), i.e. for the next
// segment's checkpoints that are meant to arrive under this segment's arithmetic
#ifdef BWD_EXP_NOBAR   // (timing-only build, WRONG RESULTS: what do the 42 barriers per tile cost?)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// CK_ROWS = region rows: 48 (one block of 768 threads per CU, three waves per SIMD).  Measured alternatives
// (profiles/r03_backward_checkpoints.md): 24 and 32 rows with two blocks per CU, 64 rows with 1024 threads -- all slower.
template <int CK_ROWS>
__global__ __launch_bounds__(CK_ROWS * CK_GR) __attribute__((amdgpu_waves_per_eu(3, 3))) void bwd_final_ck_kernel(
    const float* __restrict__ g, const float* __restrict__ blur, const float* __restrict__ sparse, const float* __restrict__ hh,
    const float* __restrict__ ah, const float* __restrict__ wf, const float* __restrict__ a0p, const float* __restrict__ gout,
    float* __restrict__ gg, float* __restrict__ gb, int B, int H, int W, int norm, int nseg) {
    constexpr int CK_NT = CK_ROWS * CK_GR, CK_TR = CK_ROWS - 2 * CK;
    const int NSEG = nseg;   // n_iter / 4 segments of four levels (round 5: n_iter = 4, 8 .. 24; 6 for the reference's 24)
    __shared__ __attribute__((aligned(16))) float4 sH[2][CK_NT];       // H_{s+l} of the region, two planes alternating (image order inside a quad)
    __shared__ __attribute__((aligned(16))) float4 sA[CK][CK_NT];      // A_{s+1} .. A_{s+4}: every thread's own group only
    __shared__ __attribute__((aligned(16))) float4 sT[2][2][CK_NT];    // [parity][to the row below | above]
    const int tid = threadIdx.x, gx = tid & (CK_GR - 1), ry = tid >> 4;
    const int W4 = W >> 2;
    // 1-D grid; the hardware deals block i to XCD i % 8: every XCD gets a contiguous run of tiles (numbered along x, then y, then
    // the batch), so that the tiles whose regions overlap run on the same XCD at about the same time and share its L2
    const int nbx = (W4 + CK_TG - 1) / CK_TG, nby = (H + CK_TR - 1) / CK_TR, ntile = nbx * nby * B, per = (ntile + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= ntile || (int)(blockIdx.x >> 3) >= per) return;   // (whole block: no barrier is left waiting)
    const int bx = tile % nbx, by = (tile / nbx) % nby, b = tile / (nbx * nby);
    const int y = by * CK_TR - CK + ry, xg = bx * CK_TG - 1 + gx;
    const bool inimg = y >= 0 && y < H && xg >= 0 && xg < W4;
    const bool intile = inimg && ry >= CK && ry < CK_ROWS - CK && gx >= 1 && gx < CK_GR - 1;
    const int x = 4 * (inimg ? xg : 0);
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)b * HW + (size_t)(inimg ? y : 0) * W + x;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (idx is a valid group of its row even for threads outside the image -- clamped above --: load unconditionally, select the value.
    // A conditional load through this lambda had been compiled to four FLAT dword loads, which also tie up lgkmcnt, the counter
    // every LDS barrier below waits on)
    auto ld = [&](const float* p) {
        typedef float v4g __attribute__((ext_vector_type(4)));
        const v4g v = *(const __attribute__((address_space(1))) v4g*)(p + idx);   // one global_load_dwordx4
        return inimg ? make_float4(v.x, v.y, v.z, v.w) : z4;
    };
    // the thread's own coefficients and constant term; the arithmetic below runs on pixel pairs (v_pk_fma_f32 / v_pk_mul_f32 /
    // v_pk_add_f32: the pass is bound by VALU issue, not by memory)
    v2f w[8][2], cp[2];
    {
        const float4 h0 = ld(blur);
        v2f sw[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = ld(wf + (size_t)k * total);
            w[k][0] = v2f{q.x, q.y}; w[k][1] = v2f{q.z, q.w};
            sw[0] += w[k][0]; sw[1] += w[k][1];
        }
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        if (sparse) { const float4 sq = ld(sparse); m[0] = signf(sq.x); m[1] = signf(sq.y); m[2] = signf(sq.z); m[3] = signf(sq.w); }
        const float h0a[4] = {h0.x, h0.y, h0.z, h0.w};
        float c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = inimg ? (norm == CSPN_NORM_NONE ? m[i] * h0a[i] : h0a[i] * (1.f - (i < 2 ? sw[0][i] : sw[1][i - 2]))) : 0.f;
        cp[0] = v2f{c[0], c[1]}; cp[1] = v2f{c[2], c[3]};
    }
    v2f dW[8][2], dC[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 8; ++k) dW[k][0] = dW[k][1] = v2f{0.f, 0.f};
    const v2f zero2 = {0.f, 0.f};
    // the three rows around the thread's row of a level plane in LDS as pixel pairs: hp[d][o][half] = columns (x-1+o+2 half, +1) of
    // row y+1 (d 0), y (1), y-1 (2), o = 0, 1, 2: what the taps dx = -1, 0, +1 multiply with
    // (indices, not pointers: a pointer to a level plane that passes through a lambda or a select becomes a generic pointer and
    // its reads flat loads that take the vector-memory path -- ten times an LDS read's latency; at the region's first / last
    // row the thread reads its own row again: halo)
    const int tdn_i = ry < CK_ROWS - 1 ? tid + CK_GR : tid, tup_i = ry > 0 ? tid - CK_GR : tid;
    // pixel pairs of the three rows around the thread's row: hp[d][o][half] = columns (x-1+o+2 half, +1) of row y+1 (d 0), y (1),
    // y-1 (2), o = 0, 1, 2: what the taps dx = -1, 0, +1 multiply with.  The own row comes from the registers.
    auto rows_of = [&](int pl, const float4 own, v2f (&hp)[3][3][2]) {
        const float4 dn = sH[pl][tdn_i], up = sH[pl][tup_i];
        const float4 q[3] = {dn, own, up};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float l = dpp_shr1(q[d].w), r = dpp_shl1(q[d].x);   // (the first / last lane of a region row gets 0: halo)
            hp[d][0][0] = v2f{l, q[d].x};      hp[d][0][1] = v2f{q[d].y, q[d].z};
            hp[d][1][0] = v2f{q[d].x, q[d].y}; hp[d][1][1] = v2f{q[d].z, q[d].w};
            hp[d][2][0] = v2f{q[d].y, q[d].z}; hp[d][2][1] = v2f{q[d].w, r};
        }
    };
    // tap k multiplies with row d = (0,0,0,1,1,2,2,2)[k], column offset o = (2,1,0,2,0,2,1,0)[k]
    // raw quads (the sweeps' planes are in register order, blur / dL/dout in image order): reordered when they are USED, so that a
    // prefetched quad is not waited for at the point of the request
    auto seg_h = [&](int j) { return ld(j == 0 ? blur : hh + (size_t)(j - 1) * total); };                       // H_{4j}
    auto seg_a = [&](int j) { return ld(j == NSEG - 1 ? gout : ah + (size_t)(NSEG - 2 - j) * total); };         // A_{4j+4}
    const bool wave_in_tile_rows = (ry & ~3) >= CK && (ry & ~3) < CK_ROWS - CK;   // a wave = 4 region rows: the first / last wave only feeds
    float4 nh = seg_h(0), na = seg_a(0);   // the checkpoints are requested one segment ahead
    int par = 0;
#ifdef BWD_EXP_NOLOOP
    const int NSEG_RUN = 0;
#else
    const int NSEG_RUN = NSEG;
#endif
    // the whole segment loop, twice: blocks whose region lies inside the image (two thirds of them at KITTI size) need no
    // "outside the image -> 0" selects
    auto segments = [&](auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
    #pragma unroll 1
        for (int j = 0; j < NSEG_RUN; ++j) {
            float4 hq, aq4;   // H_s, A_{s+4} of the thread's group
            hq = j == 0 ? nh : reg_to_img(nh);
            aq4 = j == NSEG - 1 ? na : reg_to_img(na);
            if (j + 1 < NSEG) { nh = seg_h(j + 1); na = seg_a(j + 1); }   // the next segment's checkpoints arrive under this one's arithmetic
            // ---- the adjoint levels first (they do not depend on H): A_{s+4} -> A_{s+3}, A_{s+2}, A_{s+1}, each parked in the thread's
            // own LDS slot (sA[i] = A_{s+i+1}; nobody else reads it: no barrier for these)
            v2f a[2] = {v2f{aq4.x, aq4.y}, v2f{aq4.z, aq4.w}};
            sA[CK - 1][tid] = aq4;
    #pragma unroll 1
            for (int l = CK - 1; l >= 1; --l) {
                // A_t from A_{t+1}: P_k(p) = w'_k(p) A_{t+1}(p) goes to q = p + off_k; off_k = (+1,+1) (+1,0) (+1,-1) (0,+1) (0,-1) (-1,+1) (-1,0) (-1,-1).
                // Column x' of the destination row takes P_k from column x' - dx_k of this row: dx = +1 -> the value one to the left.
                v2f tdn[2], tup[2], mid[2];
                auto send = [&](int k, v2f (&dst)[2], int dx, bool first) {
                    const v2f p0 = w[k][0] * a[0], p1 = w[k][1] * a[1];
                    v2f v0, v1;
                    if (dx > 0) { v0 = v2f{dpp_shr1(p1[1]), p0[0]}; v1 = v2f{p0[1], p1[0]}; }
                    else if (dx < 0) { v0 = v2f{p0[1], p1[0]}; v1 = v2f{p1[1], dpp_shl1(p0[0])}; }
                    else { v0 = p0; v1 = p1; }
                    dst[0] = first ? v0 : dst[0] + v0;
                    dst[1] = first ? v1 : dst[1] + v1;
                };
                send(0, tdn, 1, true); send(1, tdn, 0, false); send(2, tdn, -1, false);
                send(3, mid, 1, true); send(4, mid, -1, false);
                send(5, tup, 1, true); send(6, tup, 0, false); send(7, tup, -1, false);
                sT[par][0][tid] = make_float4(tdn[0][0], tdn[0][1], tdn[1][0], tdn[1][1]);
                sT[par][1][tid] = make_float4(tup[0][0], tup[0][1], tup[1][0], tup[1][1]);
                lds_barrier();
                const float4 fa = sT[par][0][tup_i];     // from the row above, sent down
                const float4 fb = sT[par][1][tdn_i];     // from the row below, sent up
                par ^= 1;
                a[0] = mid[0] + v2f{fa.x, fa.y} + v2f{fb.x, fb.y};
                if (MASKED && !inimg) a[0] = zero2;
                a[1] = mid[1] + v2f{fa.z, fa.w} + v2f{fb.z, fb.w};
                if (MASKED && !inimg) a[1] = zero2;
                sA[l - 1][tid] = make_float4(a[0][0], a[0][1], a[1][0], a[1][1]);
            }
            // ---- then H_s -> H_{s+3}; the pixel pairs built for a step are also what A_{t+1} multiplies with for dW'
            sH[0][tid] = hq;
            lds_barrier();
    #pragma unroll 1
            for (int l = 0; l < CK; ++l) {
                v2f hp[3][3][2];
                rows_of(l & 1, hq, hp);
                if (wave_in_tile_rows) {
                    const float4 aq = sA[l][tid];   // A_{s+l+1}
                    const v2f av[2] = {v2f{aq.x, aq.y}, v2f{aq.z, aq.w}};
    #pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        dC[hlf] += av[hlf];
                        dW[0][hlf] = __builtin_elementwise_fma(av[hlf], hp[0][2][hlf], dW[0][hlf]); dW[1][hlf] = __builtin_elementwise_fma(av[hlf], hp[0][1][hlf], dW[1][hlf]);
                        dW[2][hlf] = __builtin_elementwise_fma(av[hlf], hp[0][0][hlf], dW[2][hlf]); dW[3][hlf] = __builtin_elementwise_fma(av[hlf], hp[1][2][hlf], dW[3][hlf]);
                        dW[4][hlf] = __builtin_elementwise_fma(av[hlf], hp[1][0][hlf], dW[4][hlf]); dW[5][hlf] = __builtin_elementwise_fma(av[hlf], hp[2][2][hlf], dW[5][hlf]);
                        dW[6][hlf] = __builtin_elementwise_fma(av[hlf], hp[2][1][hlf], dW[6][hlf]); dW[7][hlf] = __builtin_elementwise_fma(av[hlf], hp[2][0][hlf], dW[7][hlf]);
                    }
                }
                if (l == CK - 1) break;
                v2f n[2];
    #pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    v2f t = cp[hlf];
                    t = __builtin_elementwise_fma(w[0][hlf], hp[0][2][hlf], t); t = __builtin_elementwise_fma(w[1][hlf], hp[0][1][hlf], t);
                    t = __builtin_elementwise_fma(w[2][hlf], hp[0][0][hlf], t); t = __builtin_elementwise_fma(w[3][hlf], hp[1][2][hlf], t);
                    t = __builtin_elementwise_fma(w[4][hlf], hp[1][0][hlf], t); t = __builtin_elementwise_fma(w[5][hlf], hp[2][2][hlf], t);
                    t = __builtin_elementwise_fma(w[6][hlf], hp[2][1][hlf], t); t = __builtin_elementwise_fma(w[7][hlf], hp[2][0][hlf], t);
                    n[hlf] = (MASKED && !inimg) ? zero2 : t;
                }
                hq = make_float4(n[0][0], n[0][1], n[1][0], n[1][1]);
                sH[(l + 1) & 1][tid] = hq;   // (the plane read two steps ago: everybody is past the barrier in between)
                lds_barrier();
            }
        }
    };
    const int ry0 = by * CK_TR - CK, xg0 = bx * CK_TG - 1;
    const bool blk_in = ry0 >= 0 && ry0 + CK_ROWS <= H && xg0 >= 0 && xg0 + CK_GR <= W4;
    if (blk_in) segments(std::false_type{});
    else segments(std::true_type{});
#ifdef BWD_EXP_NOEPI
    if (dC[0][0] != 12345.f) return;
#endif
    if (!intile) return;
    float dWs[8][4], dCs[4] = {dC[0][0], dC[0][1], dC[1][0], dC[1][1]};
#pragma unroll
    for (int k = 0; k < 8; ++k) { dWs[k][0] = dW[k][0][0]; dWs[k][1] = dW[k][0][1]; dWs[k][2] = dW[k][1][0]; dWs[k][3] = dW[k][1][1]; }
    if (blk_in) bwd_epilogue4<true>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, dWs, dCs);
    else bwd_epilogue4<false>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, dWs, dCs);
}


// ---- round 5: the same pass with MIXED pixel pairs (the trick of the forward ring) -----------------------------------
// bwd_final_ck_kernel spends a third of its ~3 550 instructions per wave and tile on v_mov: a group's four pixels sit in image
// order (c0, c1, c2, c3), so the pairs the x +- 1 taps multiply with -- (c-1, c0), (c1, c2), (c3, c+1) -- straddle the aligned
// register pairs and are copied together for every row of every step.  Here a group lives in REGISTER order (c0, c3, c1, c2) --
// the order the sweeps store their checkpoints in anyway --, X = (c0, c3), Y = (c1, c2), D = (c3 of the lane before, c0 of the lane
// after: two DPP moves), and every packed FMA combines the dx = +1 tap of its low pixel with the dx = -1 tap of its high pixel:
//   X += (w_+(c0), w_-(c3)) * Y      X += (w_-(c0), w_+(c3)) * D      X += (w_0(c0), w_0(c3)) * X
//   Y += (w_-(c1), w_+(c2)) * X      Y += (w_0(c1), w_0(c2)) * Y      Ysw += (w_-(c2), w_+(c1)) * Y   (Y += swap(Ysw) once per step)
// per row of taps (w_+, w_0, w_- = the row's dx = +1, 0, -1 coefficients), with the coefficient pairs mixed ONCE per tile.  The same
// pairs serve the adjoint step in push form (products with A instead of H; what leaves for the next lane goes through the same two
// DPP moves) and the gradient accumulators dW', which stay mixed until the epilogue.  No operand is ever copied: an H step with its 16
// products is 32 v_pk_fma_f32 + 6 DPP + 3 swaps, an adjoint step 16 packed multiplies / FMAs + 6 DPP + 1 swap.  Same arithmetic per
// pixel as bwd_final_ck_kernel up to the order of the nine terms of a sum.
__device__ __forceinline__ float4 img_to_reg(float4 q) { return make_float4(q.x, q.w, q.y, q.z); }   // (c0..c3) -> (c0,c3,c1,c2)
__device__ __forceinline__ v2f swp2(v2f v) { return __builtin_shufflevector(v, v, 1, 0); }

template <int CK_ROWS>
__global__ __launch_bounds__(CK_ROWS * CK_GR) __attribute__((amdgpu_waves_per_eu(CK_ROWS / 16, CK_ROWS / 16))) void bwd_final_mx_kernel(
    const float* __restrict__ g, const float* __restrict__ blur, const float* __restrict__ sparse, const float* __restrict__ hh,
    const float* __restrict__ ah, const float* __restrict__ wf, const float* __restrict__ a0p, const float* __restrict__ gout,
    float* __restrict__ gg, float* __restrict__ gb, int B, int H, int W, int norm, int nseg) {
    constexpr int CK_NT = CK_ROWS * CK_GR, CK_TR = CK_ROWS - 2 * CK;
    const int NSEG = nseg;   // n_iter / 4 segments of four levels (round 5: n_iter = 4, 8 .. 24; 6 for the reference's 24)
    __shared__ __attribute__((aligned(16))) float4 sH[2][CK_NT];       // H_{s+l} of the region, two planes alternating (REGISTER order inside a quad)
    __shared__ __attribute__((aligned(16))) float4 sA[CK][CK_NT];      // A_{s+1} .. A_{s+4}: every thread's own group only
    __shared__ __attribute__((aligned(16))) float4 sT[2][2][CK_NT];    // [parity][to the row below | above]
    const int tid = threadIdx.x, gx = tid & (CK_GR - 1), ry = tid >> 4;
    const int W4 = W >> 2;
    const int nbx = (W4 + CK_TG - 1) / CK_TG, nby = (H + CK_TR - 1) / CK_TR, ntile = nbx * nby * B, per = (ntile + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);   // XCD-aware tile order, as bwd_final_ck_kernel
    if (tile >= ntile || (int)(blockIdx.x >> 3) >= per) return;
    const int bx = tile % nbx, by = (tile / nbx) % nby, b = tile / (nbx * nby);
    const int y = by * CK_TR - CK + ry, xg = bx * CK_TG - 1 + gx;
    const bool inimg = y >= 0 && y < H && xg >= 0 && xg < W4;
    const bool intile = inimg && ry >= CK && ry < CK_ROWS - CK && gx >= 1 && gx < CK_GR - 1;
    const int x = 4 * (inimg ? xg : 0);
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)b * HW + (size_t)(inimg ? y : 0) * W + x;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](const float* p) {
        typedef float v4g __attribute__((ext_vector_type(4)));
        const v4g v = *(const __attribute__((address_space(1))) v4g*)(p + idx);   // one global_load_dwordx4
        return inimg ? make_float4(v.x, v.y, v.z, v.w) : z4;
    };
    // coefficient pairs, mixed once per tile.  Row of taps d = 0 (dy = +1: k = 0, 1, 2), 1 (dy = 0: k = 3, 4), 2 (dy = -1: k = 5, 6, 7);
    // in a row: w_+ = the dx = +1 tap, w_0 = dx = 0, w_- = dx = -1
    v2f mAX[3], mBX[3], mAYs[3], mBY[3], n0X[3], n0Y[3], cpX, cpY;
    {
        const float4 h0 = ld(blur);
        float4 wq[8];
        float sw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wq[k] = ld(wf + (size_t)k * total);
            sw[0] += wq[k].x; sw[1] += wq[k].y; sw[2] += wq[k].z; sw[3] += wq[k].w;
        }
        constexpr int KP[3] = {0, 3, 5}, K0[3] = {1, 1, 6}, KM[3] = {2, 4, 7};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mAX[d] = v2f{wq[KP[d]].x, wq[KM[d]].w};    // (w_+(c0), w_-(c3))
            mBX[d] = v2f{wq[KM[d]].x, wq[KP[d]].w};    // (w_-(c0), w_+(c3))
            mAYs[d] = v2f{wq[KM[d]].z, wq[KP[d]].y};   // (w_-(c2), w_+(c1))
            mBY[d] = v2f{wq[KM[d]].y, wq[KP[d]].z};    // (w_-(c1), w_+(c2))
            n0X[d] = d == 1 ? v2f{0.f, 0.f} : v2f{wq[K0[d]].x, wq[K0[d]].w};
            n0Y[d] = d == 1 ? v2f{0.f, 0.f} : v2f{wq[K0[d]].y, wq[K0[d]].z};
        }
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        if (sparse) { const float4 sq = ld(sparse); m[0] = signf(sq.x); m[1] = signf(sq.y); m[2] = signf(sq.z); m[3] = signf(sq.w); }
        const float h0a[4] = {h0.x, h0.y, h0.z, h0.w};
        float c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = inimg ? (norm == CSPN_NORM_NONE ? m[i] * h0a[i] : h0a[i] * (1.f - sw[i])) : 0.f;
        cpX = v2f{c[0], c[3]}; cpY = v2f{c[1], c[2]};
    }
    const v2f zero2 = {0.f, 0.f};
    v2f dAX[3], dBX[3], dAYs[3], dBY[3], dN0X[3], dN0Y[3], dCX = zero2, dCY = zero2;   // the gradient accumulators, mixed like the coefficients
#pragma unroll
    for (int d = 0; d < 3; ++d) dAX[d] = dBX[d] = dAYs[d] = dBY[d] = dN0X[d] = dN0Y[d] = zero2;
    const int tdn_i = ry < CK_ROWS - 1 ? tid + CK_GR : tid, tup_i = ry > 0 ? tid - CK_GR : tid;
    auto seg_h = [&](int j) { return ld(j == 0 ? blur : hh + (size_t)(j - 1) * total); };                       // H_{4j}
    auto seg_a = [&](int j) { return ld(j == NSEG - 1 ? gout : ah + (size_t)(NSEG - 2 - j) * total); };         // A_{4j+4}
    const bool wave_in_tile_rows = (ry & ~3) >= CK && (ry & ~3) < CK_ROWS - CK;
#ifndef BWD_FINAL_MERGED   // the product: the adjoint levels of a segment first, then its H steps (7 barriers per segment, one dependent chain at a time)
    float4 nh = seg_h(0), na = seg_a(0);   // the checkpoints are requested one segment ahead
    int par = 0;
    auto segments = [&](auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
#pragma unroll 1
        for (int j = 0; j < NSEG; ++j) {
            // (the sweeps' planes are in register order already; blur / dL/dout are in image order)
            float4 hq = j == 0 ? img_to_reg(nh) : nh;
            const float4 aq4 = j == NSEG - 1 ? img_to_reg(na) : na;
            if (j + 1 < NSEG) { nh = seg_h(j + 1); na = seg_a(j + 1); }
            // ---- the adjoint levels first: A_{s+4} -> A_{s+3}, A_{s+2}, A_{s+1} (push form), each parked in the thread's own LDS slot
            v2f aX = v2f{aq4.x, aq4.y}, aY = v2f{aq4.z, aq4.w};
            sA[CK - 1][tid] = aq4;
#pragma unroll 1
            for (int l = CK - 1; l >= 1; --l) {
                const v2f aYs = swp2(aY);
                v2f tX[3], tY[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const v2f M = mBX[d] * aX;                                   // (P_-(c0), P_+(c3)): what leaves for the neighbouring lanes
                    tX[d] = v2f{dpp_shr1(M[1]), dpp_shl1(M[0])};                 // (P_+(c3 of the lane before), P_-(c0 of the lane after))
                    tX[d] = __builtin_elementwise_fma(mBY[d], aY, tX[d]);        // + (P_-(c1), P_+(c2))
                    tY[d] = mAX[d] * aX;                                         // (P_+(c0), P_-(c3))
                    tY[d] = __builtin_elementwise_fma(mAYs[d], aYs, tY[d]);      // + (P_-(c2), P_+(c1))
                    if (d != 1) {
                        tX[d] = __builtin_elementwise_fma(n0X[d], aX, tX[d]);
                        tY[d] = __builtin_elementwise_fma(n0Y[d], aY, tY[d]);
                    }
                }
                sT[par][0][tid] = make_float4(tX[0][0], tX[0][1], tY[0][0], tY[0][1]);   // dy = +1: to the row below
                sT[par][1][tid] = make_float4(tX[2][0], tX[2][1], tY[2][0], tY[2][1]);   // dy = -1: to the row above
                lds_barrier();
                const float4 fa = sT[par][0][tup_i];     // from the row above, sent down
                const float4 fb = sT[par][1][tdn_i];     // from the row below, sent up
                par ^= 1;
                aX = tX[1] + v2f{fa.x, fa.y} + v2f{fb.x, fb.y};
                aY = tY[1] + v2f{fa.z, fa.w} + v2f{fb.z, fb.w};
                if (MASKED && !inimg) { aX = zero2; aY = zero2; }
                sA[l - 1][tid] = make_float4(aX[0], aX[1], aY[0], aY[1]);
            }
            // ---- then H_s -> H_{s+3}; the row pairs of a step are also what A_{t+1} multiplies with for dW'
            sH[0][tid] = hq;
            lds_barrier();
#pragma unroll 1
            for (int l = 0; l < CK; ++l) {
                const float4 q[3] = {sH[l & 1][tdn_i], hq, sH[l & 1][tup_i]};   // rows y + 1, y, y - 1 (at the region's first / last row: the own row again, halo)
                v2f Xr[3], Yr[3], Dr[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    Xr[d] = v2f{q[d].x, q[d].y};
                    Yr[d] = v2f{q[d].z, q[d].w};
                    Dr[d] = v2f{dpp_shr1(q[d].y), dpp_shl1(q[d].x)};   // (c3 of the lane before, c0 of the lane after; 0 at the region's edge: halo)
                }
                if (wave_in_tile_rows) {
                    const float4 aq = sA[l][tid];   // A_{s+l+1}
                    const v2f bX = v2f{aq.x, aq.y}, bY = v2f{aq.z, aq.w}, bYs = swp2(bY);
                    dCX += bX;
                    dCY += bY;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        dAX[d] = __builtin_elementwise_fma(bX, Yr[d], dAX[d]);
                        dBX[d] = __builtin_elementwise_fma(bX, Dr[d], dBX[d]);
                        dBY[d] = __builtin_elementwise_fma(bY, Xr[d], dBY[d]);
                        dAYs[d] = __builtin_elementwise_fma(bYs, Yr[d], dAYs[d]);
                        if (d != 1) {
                            dN0X[d] = __builtin_elementwise_fma(bX, Xr[d], dN0X[d]);
                            dN0Y[d] = __builtin_elementwise_fma(bY, Yr[d], dN0Y[d]);
                        }
                    }
                }
                if (l == CK - 1) break;
                v2f nX = cpX, nY = cpY, nYs = mAYs[0] * Yr[0];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    nX = __builtin_elementwise_fma(mAX[d], Yr[d], nX);
                    nX = __builtin_elementwise_fma(mBX[d], Dr[d], nX);
                    nY = __builtin_elementwise_fma(mBY[d], Xr[d], nY);
                    if (d != 0) nYs = __builtin_elementwise_fma(mAYs[d], Yr[d], nYs);
                    if (d != 1) {
                        nX = __builtin_elementwise_fma(n0X[d], Xr[d], nX);
                        nY = __builtin_elementwise_fma(n0Y[d], Yr[d], nY);
                    }
                }
                nY += swp2(nYs);
                if (MASKED && !inimg) { nX = zero2; nY = zero2; }
                hq = make_float4(nX[0], nX[1], nY[0], nY[1]);
                sH[(l + 1) & 1][tid] = hq;   // (the plane read two steps ago: everybody is past the barrier in between)
                lds_barrier();
            }
        }
    };
#else
    // Round 6 A/B build (-DBWD_FINAL_MERGED; parity-green, 3 % SLOWER: profiles/r06_backward_merge_ab.md): the adjoint steps of segment j + 1 run INSIDE the H
    // steps of segment j -- two independent chains per iteration for the three waves of a SIMD to interleave, one LDS round trip and ONE barrier for both (4 per
    // segment instead of 7).  A_{s'+4} (the checkpoint) and A_{s'+3} of the next segment wait in
    // registers until the segment ends (their LDS slots are still read by this segment's products), A_{s'+2} and A_{s'+1} go straight to their slots.
    float4 nh = seg_h(0), na = seg_a(0);   // H checkpoints are requested one segment ahead, A checkpoints two
    int par = 0;
    auto adj_products = [&](v2f aX, v2f aY, v2f (&tX)[3], v2f (&tY)[3]) {   // what a pixel sends to its eight neighbours (push form), summed per target row
        const v2f aYs = swp2(aY);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const v2f M = mBX[d] * aX;                                   // (P_-(c0), P_+(c3)): what leaves for the neighbouring lanes
            tX[d] = v2f{dpp_shr1(M[1]), dpp_shl1(M[0])};                 // (P_+(c3 of the lane before), P_-(c0 of the lane after))
            tX[d] = __builtin_elementwise_fma(mBY[d], aY, tX[d]);        // + (P_-(c1), P_+(c2))
            tY[d] = mAX[d] * aX;                                         // (P_+(c0), P_-(c3))
            tY[d] = __builtin_elementwise_fma(mAYs[d], aYs, tY[d]);      // + (P_-(c2), P_+(c1))
            if (d != 1) {
                tX[d] = __builtin_elementwise_fma(n0X[d], aX, tX[d]);
                tY[d] = __builtin_elementwise_fma(n0Y[d], aY, tY[d]);
            }
        }
    };
    auto segments = [&](auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        {   // the adjoint levels of segment 0: A_4 -> A_3, A_2, A_1, each parked in the thread's own LDS slot
            const float4 aq4 = NSEG == 1 ? img_to_reg(na) : na;
            v2f aX = v2f{aq4.x, aq4.y}, aY = v2f{aq4.z, aq4.w};
            sA[CK - 1][tid] = aq4;
#pragma unroll 1
            for (int l = CK - 1; l >= 1; --l) {
                v2f tX[3], tY[3];
                adj_products(aX, aY, tX, tY);
                sT[par][0][tid] = make_float4(tX[0][0], tX[0][1], tY[0][0], tY[0][1]);   // dy = +1: to the row below
                sT[par][1][tid] = make_float4(tX[2][0], tX[2][1], tY[2][0], tY[2][1]);   // dy = -1: to the row above
                lds_barrier();
                const float4 fa = sT[par][0][tup_i], fb = sT[par][1][tdn_i];
                par ^= 1;
                aX = tX[1] + v2f{fa.x, fa.y} + v2f{fb.x, fb.y};
                aY = tY[1] + v2f{fa.z, fa.w} + v2f{fb.z, fb.w};
                if (MASKED && !inimg) { aX = zero2; aY = zero2; }
                sA[l - 1][tid] = make_float4(aX[0], aX[1], aY[0], aY[1]);
            }
            if (NSEG > 1) na = seg_a(1);
        }
#pragma unroll 1
        for (int j = 0; j < NSEG; ++j) {
            // (the sweeps' planes are in register order already; blur / dL/dout are in image order)
            float4 hq = j == 0 ? img_to_reg(nh) : nh;
            const bool chain = j + 1 < NSEG;                                   // the next segment's adjoint levels are computed during this one
            const float4 keep3 = chain ? (j + 1 == NSEG - 1 ? img_to_reg(na) : na) : z4;   // A_{s'+4}: the next segment's checkpoint
            float4 keep2 = z4;                                                 // A_{s'+3}
            if (j + 1 < NSEG) nh = seg_h(j + 1);
            if (j + 2 < NSEG) na = seg_a(j + 2);
            v2f aX = v2f{keep3.x, keep3.y}, aY = v2f{keep3.z, keep3.w};
            sH[0][tid] = hq;
            lds_barrier();
            auto rows = [&](int l, v2f (&Xr)[3], v2f (&Yr)[3], v2f (&Dr)[3]) {   // rows y + 1, y, y - 1 of H_{s+l} (at the region's first / last row: the own row again, halo)
                const float4 q[3] = {sH[l & 1][tdn_i], hq, sH[l & 1][tup_i]};
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    Xr[d] = v2f{q[d].x, q[d].y};
                    Yr[d] = v2f{q[d].z, q[d].w};
                    Dr[d] = v2f{dpp_shr1(q[d].y), dpp_shl1(q[d].x)};   // (c3 of the lane before, c0 of the lane after; 0 at the region's edge: halo)
                }
            };
            auto products = [&](int l, const v2f (&Xr)[3], const v2f (&Yr)[3], const v2f (&Dr)[3]) {   // dW' += A_{s+l+1} x the neighbours of H_{s+l}; dC += A_{s+l+1}
                const float4 aq = sA[l][tid];
                const v2f bX = v2f{aq.x, aq.y}, bY = v2f{aq.z, aq.w}, bYs = swp2(bY);
                dCX += bX;
                dCY += bY;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    dAX[d] = __builtin_elementwise_fma(bX, Yr[d], dAX[d]);
                    dBX[d] = __builtin_elementwise_fma(bX, Dr[d], dBX[d]);
                    dBY[d] = __builtin_elementwise_fma(bY, Xr[d], dBY[d]);
                    dAYs[d] = __builtin_elementwise_fma(bYs, Yr[d], dAYs[d]);
                    if (d != 1) {
                        dN0X[d] = __builtin_elementwise_fma(bX, Xr[d], dN0X[d]);
                        dN0Y[d] = __builtin_elementwise_fma(bY, Yr[d], dN0Y[d]);
                    }
                }
            };
#pragma unroll 1
            for (int l = 0; l < CK - 1; ++l) {
                v2f Xr[3], Yr[3], Dr[3];
                rows(l, Xr, Yr, Dr);
                if (wave_in_tile_rows) products(l, Xr, Yr, Dr);
                // ONE basic block: the H step and an adjoint step of the next segment (of zeros in the last segment), for the scheduler to interleave
                v2f tX[3], tY[3];
                adj_products(aX, aY, tX, tY);
                v2f nX = cpX, nY = cpY, nYs = mAYs[0] * Yr[0];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    nX = __builtin_elementwise_fma(mAX[d], Yr[d], nX);
                    nX = __builtin_elementwise_fma(mBX[d], Dr[d], nX);
                    nY = __builtin_elementwise_fma(mBY[d], Xr[d], nY);
                    if (d != 0) nYs = __builtin_elementwise_fma(mAYs[d], Yr[d], nYs);
                    if (d != 1) {
                        nX = __builtin_elementwise_fma(n0X[d], Xr[d], nX);
                        nY = __builtin_elementwise_fma(n0Y[d], Yr[d], nY);
                    }
                }
                nY += swp2(nYs);
                if (MASKED && !inimg) { nX = zero2; nY = zero2; }
                hq = make_float4(nX[0], nX[1], nY[0], nY[1]);
                sT[par][0][tid] = make_float4(tX[0][0], tX[0][1], tY[0][0], tY[0][1]);   // dy = +1: to the row below
                sT[par][1][tid] = make_float4(tX[2][0], tX[2][1], tY[2][0], tY[2][1]);   // dy = -1: to the row above
                sH[(l + 1) & 1][tid] = hq;   // (the plane read two steps ago: everybody is past the barrier in between)
                lds_barrier();               // (one barrier for both exchanges)
                const float4 fa = sT[par][0][tup_i], fb = sT[par][1][tdn_i];     // from the row above, sent down / from the row below, sent up
                par ^= 1;
                aX = tX[1] + v2f{fa.x, fa.y} + v2f{fb.x, fb.y};
                aY = tY[1] + v2f{fa.z, fa.w} + v2f{fb.z, fb.w};
                if (MASKED && !inimg) { aX = zero2; aY = zero2; }
                const float4 an = make_float4(aX[0], aX[1], aY[0], aY[1]);     // A_{s'+3-l} of the next segment
                if (l == 0) keep2 = an;                       // its slot (and the checkpoint's) is still read by this segment's products
                else if (chain) sA[CK - 2 - l][tid] = an;     // l = 1 -> slot 1 (read above, in this iteration), l = 2 -> slot 0 (read at l = 0)
            }
            {
                v2f Xr[3], Yr[3], Dr[3];
                rows(CK - 1, Xr, Yr, Dr);
                if (wave_in_tile_rows) products(CK - 1, Xr, Yr, Dr);
            }
            if (chain) { sA[CK - 1][tid] = keep3; sA[CK - 2][tid] = keep2; }   // (own slots, read by this thread only: no barrier)
        }
    };
#endif
    const int ry0 = by * CK_TR - CK, xg0 = bx * CK_TG - 1;
    const bool blk_in = ry0 >= 0 && ry0 + CK_ROWS <= H && xg0 >= 0 && xg0 + CK_GR <= W4;
    if (blk_in) segments(std::false_type{});
    else segments(std::true_type{});
    if (!intile) return;
    // un-mix: dW'_k per pixel in image order, as the epilogue wants it
    float dWs[8][4], dCs[4] = {dCX[0], dCY[0], dCY[1], dCX[1]};
    constexpr int KP[3] = {0, 3, 5}, K0[3] = {1, 1, 6}, KM[3] = {2, 4, 7};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        dWs[KP[d]][0] = dAX[d][0];  dWs[KM[d]][3] = dAX[d][1];
        dWs[KM[d]][0] = dBX[d][0];  dWs[KP[d]][3] = dBX[d][1];
        dWs[KM[d]][2] = dAYs[d][0]; dWs[KP[d]][1] = dAYs[d][1];
        dWs[KM[d]][1] = dBY[d][0];  dWs[KP[d]][2] = dBY[d][1];
        if (d != 1) {
            dWs[K0[d]][0] = dN0X[d][0]; dWs[K0[d]][3] = dN0X[d][1];
            dWs[K0[d]][1] = dN0Y[d][0]; dWs[K0[d]][2] = dN0Y[d][1];
        }
    }
#ifdef BWD_EXP_MX_NOEPI   // (compile-time probe, WRONG RESULTS: how many registers does the level loop need on its own?)
    float acc = dCs[0] + dCs[1] + dCs[2] + dCs[3];
    for (int k = 0; k < 8; ++k) acc += dWs[k][0] + dWs[k][1] + dWs[k][2] + dWs[k][3];
    gb[idx] = acc;
#else
    if (CK_ROWS > 48) {
        // four waves per SIMD = 128 registers: the 32 gradient accumulators wait in the thread's OWN LDS slots while the epilogue runs
        // (sA is never read by anybody else; sT has not been read since the last segment's adjoint steps, a barrier ago): no barrier needed
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sA[k][tid] = make_float4(dWs[k][0], dWs[k][1], dWs[k][2], dWs[k][3]);
            sT[k >> 1][k & 1][tid] = make_float4(dWs[4 + k][0], dWs[4 + k][1], dWs[4 + k][2], dWs[4 + k][3]);
        }
        struct LdsDW {   // (by value into the non-inlined epilogue: the thread's eight slots)
            const float4* a; const float4* t; int nt;
            __device__ float4 operator()(int k) const { return k < 4 ? a[k * nt] : t[(k - 4) * nt]; }
        } get_dw{&sA[0][tid], &sT[0][0][tid], CK_NT};
        if (blk_in) bwd_epilogue4_2pass<true>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, get_dw, dCs);
        else bwd_epilogue4_2pass<false>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, get_dw, dCs);
    } else {
        if (blk_in) bwd_epilogue4<true>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, dWs, dCs);
        else bwd_epilogue4<false>(g, blur, sparse, a0p, gg, gb, b, y, x, idx, HW, H, W, norm, dWs, dCs);
    }
#endif
}

}  // namespace

constexpr size_t FRONT_PAD = 65536;  // bytes kept addressable in front of the folded planes (the adjoint sweep reads plane 0
                                     // one row up and one pixel left of its first row)
// n_iter = 4, 8 .. 24 (round 5; rounds 3-4: 24 only).  The sweeps always run the ring's 24 levels; for n_iter < 24 the levels beyond n_iter are
// computed and ignored: the forward's result H_n is its checkpoint plane n/4 - 1, the adjoint sweep started from A_n = dL/dout leaves A_{n-4-4i} in its
// plane i and A_0 in plane n/4 - 1 (both in register order: reg_to_img_plane below), and the final pass runs n/4 segments.
static bool asm_path(int B, int H, int W, int n_iter) {
    return n_iter >= 4 && n_iter <= 24 && (n_iter % 4) == 0 && tsw2d_supported(B, H, W) && 4ull * W + 16 <= FRONT_PAD &&
           (unsigned long long)B * H * W * 32ull < (1ull << 32);  // 8 coefficient planes of per-lane byte offsets
}

size_t backward2d_workspace(int B, int H, int W, int n_iter) {
    const size_t total = (size_t)B * H * W;
    if (asm_path(B, H, W, n_iter))  // forward checkpoints 5 + folded coefficients 8 + adjoint checkpoints 5 + A_0 + a scratch output
        return FRONT_PAD + (size_t)(NCKP + 8 + NCKP + 2) * total * sizeof(float) + 256;
    return (size_t)(9 + 8 + (n_iter > 0 ? n_iter - 1 : 0) + n_iter) * total * sizeof(float);
}

// a level plane in the sweeps' register order (c0,c3,c1,c2 per aligned 4-column group) -> image order
__global__ __launch_bounds__(256) void reg_to_img_plane_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { const float4 q = src[i]; dst[i] = make_float4(q.x, q.z, q.w, q.y); }
}
static void reg_to_img_plane(const float* src, float* dst, size_t total, hipStream_t st) {
    hipLaunchKernelGGL(reg_to_img_plane_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, (const float4*)src, (float4*)dst, total / 4);
}

// final pass of the assembly-sweep backward: from the checkpoints of both sweeps
static void launch_final_ck(const float* g, const float* blur, const float* sparse, const float* hh, const float* ah, const float* wf,
                            const float* a0, const float* gout, float* gg, float* gb, int B, int H, int W, int norm, int nseg, hipStream_t st) {
#ifdef BWD_FINAL_ROWS   // (A/B build: 64 = 1 024 threads, four waves per SIMD at 128 registers, all 160 KB of LDS)
    constexpr int ROWS = BWD_FINAL_ROWS, TROWS = ROWS - 2 * CK;
#else
    constexpr int ROWS = 48, TROWS = ROWS - 2 * CK;
#endif
    const int ntile = ((W / 4 + CK_TG - 1) / CK_TG) * ((H + TROWS - 1) / TROWS) * B, per = (ntile + 7) / 8;
#ifdef BWD_FINAL_CK   // (A/B build: the round-3 kernel, image-order pixel pairs)
    hipLaunchKernelGGL(bwd_final_ck_kernel<ROWS>, dim3((unsigned)(per * 8)), dim3(ROWS * CK_GR), 0, st, g, blur, sparse, hh, ah, wf, a0, gout,
                       gg, gb, B, H, W, norm, nseg);
#else
    hipLaunchKernelGGL(bwd_final_mx_kernel<ROWS>, dim3((unsigned)(per * 8)), dim3(ROWS * CK_GR), 0, st, g, blur, sparse, hh, ah, wf, a0, gout,
                       gg, gb, B, H, W, norm, nseg);
#endif
}

int backward2d(const float* g, const float* blur, const float* sparse, const float* gout, float* gg, float* gb, int B, int H,
               int W, int n_iter, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    float* wf = (float*)ws;
    if (asm_path(B, H, W, n_iter)) {
        // both sweeps run in the fused ring kernel (cspn2d_tsw.hip), each keeping every fourth of its levels: the forward
        // as it is (it also leaves the 8 folded coefficient planes right behind its level planes), the adjoint as a
        // propagation whose coefficients are those planes read neighbour-sited with the channel order reversed (generator
        // option adj); the final pass recomputes the levels in between
        float* hh = (float*)((char*)ws + FRONT_PAD);
        wf = hh + NCKP * total;
        float* ah = wf + 8 * total;
        float* a0 = ah + NCKP * total;
        float* scratch = a0 + total;
        const int nseg = n_iter / CK;
        if (int e = tsw2d_pass(g, blur, blur, sparse, scratch, B, H, W, norm, st, hh)) return e;
        if (int e = tsw2d_adjoint_pass(wf, gout, a0, B, H, W, st, ah)) return e;
        if (nseg <= NCKP) reg_to_img_plane(ah + (size_t)(nseg - 1) * total, a0, total, st);   // A_0 of a sweep shorter than the ring
        launch_final_ck(g, blur, sparse, hh, ah, wf, a0, gout, gg, gb, B, H, W, norm, nseg, st);
        return check_launch("bwd_final_ck_kernel");
    }
    float* wt = wf + 9 * total;                       // transposed coefficients of the adjoint stencil
    float* hh = wt + 8 * total;                       // H_1 .. H_{N-1}
    float* ah = hh + (size_t)(n_iter - 1) * total;    // A_0 .. A_{N-1}
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fold2d_kernel, dim3(blocks), dim3(256), 0, st, g, blur, sparse, wf, B, H, W, norm);
    if (int e = check_launch("fold2d_kernel")) return e;
    for (int t = 1; t < n_iter; ++t)
        hipLaunchKernelGGL(step2d_kernel, dim3(blocks), dim3(256), 0, st, wf, t == 1 ? blur : hh + (size_t)(t - 2) * total,
                           hh + (size_t)(t - 1) * total, B, H, W);
    hipLaunchKernelGGL(transpose_w_kernel, dim3(blocks), dim3(256), 0, st, wf, wt, B, H, W);
    for (int t = n_iter - 1; t >= 0; --t)
        hipLaunchKernelGGL(bwd_step_kernel, dim3(blocks), dim3(256), 0, st, wt,
                           t == n_iter - 1 ? gout : ah + (size_t)(t + 1) * total, ah + (size_t)t * total, B, H, W);
    if (int e = check_launch("bwd_step_kernel")) return e;
    hipLaunchKernelGGL(bwd_final_kernel, dim3(blocks), dim3(256), 0, st, g, blur, sparse, hh, ah, gout, gg, gb, B,
                       H, W, n_iter, norm);
    return check_launch("bwd_final_kernel");
}

// ---- training mode: the forward keeps its checkpoints, the backward starts from them ------------------------------------------
// history = [FRONT_PAD bytes][H_4, H_8 .. H_20][w'_0 .. w'_7] (what the forward sweep of backward2d leaves behind)
size_t history2d_bytes(int B, int H, int W, int n_iter) {
    return asm_path(B, H, W, n_iter) ? FRONT_PAD + (size_t)(NCKP + 8) * B * H * W * sizeof(float) : 0;
}

int forward2d_history(const float* g, const float* blur, const float* sparse, float* out, void* history, int B, int H, int W,
                      int n_iter, int norm, void* ws, hipStream_t st) {
    float* hh = (float*)((char*)history + FRONT_PAD);
    (void)ws;
    if (int e = tsw2d_pass(g, blur, blur, sparse, out, B, H, W, norm, st, hh)) return e;   // (n_iter < 24: `out` = level 24 for a moment)
    const int nseg = n_iter / CK;
    if (nseg <= NCKP) {   // the result is the checkpoint H_n
        reg_to_img_plane(hh + (size_t)(nseg - 1) * B * H * W, out, (size_t)B * H * W, st);
        return check_launch("reg_to_img_plane_kernel");
    }
    return 0;
}

size_t backward2d_history_workspace(int B, int H, int W) {
    return (size_t)(NCKP + 1) * B * H * W * sizeof(float) + 256;
}

int backward2d_history(const float* g, const float* blur, const float* sparse, const float* gout, const void* history, float* gg,
                       float* gb, int B, int H, int W, int n_iter, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    const float* hh = (const float*)((const char*)history + FRONT_PAD);
    const float* wf = hh + NCKP * total;
    float* ah = (float*)ws;
    float* a0 = ah + NCKP * total;
    const int nseg = n_iter / CK;
    if (int e = tsw2d_adjoint_pass(wf, gout, a0, B, H, W, st, ah)) return e;
    if (nseg <= NCKP) reg_to_img_plane(ah + (size_t)(nseg - 1) * total, a0, total, st);
    launch_final_ck(g, blur, sparse, hh, ah, wf, a0, gout, gg, gb, B, H, W, norm, nseg, st);
    return check_launch("bwd_final_ck_kernel");
}

}  // namespace cspn
