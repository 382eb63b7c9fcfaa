// cspn2d_backward.hip -- gradient of Affinity_Propagate.forward (reference cspn_pytorch/models/cspn.py:42-83) with respect
// to guidance and blur_depth: what torch autograd computes when reference train.py:196-198 back-propagates through the
// module.  First version: correct and coalesced, one launch per iteration; not yet fused.
//
// Forward, folded (cspn2d_stepwise.hip):  H_{t+1} = c' + sum_k w'_k * shift_k(H_t),  w'_k = (1-m) w_k,
//   c' = (1-m)(1-sigma) H_0 + m H_0,  w_k = G_k / S,  S = sum_j |G_j|,  G_k(p) = g~_k(p + off_k),  sigma = sum_k w_k.
// Adjoint:  A_N = dL/dout,   A_t(p) = sum_k (w'_k A_{t+1})(p - off_k)          (bwd_step_kernel, N launches)
//           dW'_k(p) = sum_t A_{t+1}(p) H_t(p + off_k),   dC(p) = sum_t A_{t+1}(p)   (bwd_final_kernel, from the two histories)
//           dL/dw_k = (1-m)(dW'_k - dC H_0);   dL/dH_0 = A_0 + dC ((1-m)(1-sigma) + m)
//           dL/dG_k = dL/dw_k / S - sign(G_k) (sum_j dL/dw_j G_j) / S^2          (torch: d|x|/dx = sign(x), 0 at 0)
//           dL/dg_k(p + off_k) = dL/dG_k(p)  [* sign(g) for '8sum_abs'];  elements no pixel reads get 0.
// The H_t history is recomputed here with the stepwise kernels (the fused forward keeps nothing).
#include <cstdlib>

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

__device__ __forceinline__ size_t swz(size_t i) {  // register order (c0,c3,c1,c2) inside each aligned group of 4 columns
    const size_t e = i & 3;                           // (tools/tswgen/kernel.py: the pairs X = (c0,c3), Y = (c1,c2))
    return (i & ~(size_t)3) | (e == 1 ? 2 : (e == 2 ? 3 : (e == 3 ? 1 : 0)));
}

// hh: H_1 .. H_{N-1} (H_0 = blur).  SWZ = false: ah = A_0 .. A_{N-1}, plain layout (stepwise sweeps).
// SWZ = true (assembly passes, N = 24): hh and ah are level histories in register order; ah level n = A_{24-n}, a0p = A_0.
template <bool SWZ>
__global__ __launch_bounds__(256) void bwd_final_kernel(const float* __restrict__ g, const float* __restrict__ blur,
                                                         const float* __restrict__ sparse, const float* __restrict__ hh,
                                                         const float* __restrict__ ah, const float* __restrict__ a0p,
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
        else a = SWZ ? ah[(size_t)(n_iter - 2 - t) * total + swz(idx)] : ah[(size_t)(t + 1) * total + idx];
        const float* ht = (t == 0) ? blur : hh + (size_t)(t - 1) * total;
        dC += a;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float hv = 0.f;
            if (ok[k]) hv = (SWZ && t > 0) ? ht[swz(base + noff[k])] : ht[base + noff[k]];
            dW[k] = fmaf(a, hv, dW[k]);
        }
    }
    const float h0 = blur[idx];
    const float m = sparse ? signf(sparse[idx]) : 0.f;
    const float om = 1.f - m;
    const float a0 = SWZ ? a0p[idx] : ah[idx];
    const float* gbp = g + (size_t)b * 8 * HW;
    if (norm == CSPN_NORM_NONE) {  // gates used as given, centre-sited, no centre term: c' = m H_0
        if (gb) gb[idx] = a0 + dC * m;
        if (gg) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gg[(size_t)b * 8 * HW + k * HW + r] = om * dW[k];
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


// ---- final pass for the assembly sweeps, 4 columns (one register-order group) per thread -------------------------------
// Same arithmetic as bwd_final_kernel<true>.  Per level a thread reads its group of A (16 bytes) and, for each of the three
// neighbour rows, its group of H (16 bytes) plus the two single columns beside it: 10 loads for 4 pixels instead of 36, and
// the 32 dW' products of a level come out of registers.  The epilogue reads / writes the eight guidance planes as 4-column
// runs (16-byte accesses at 4-byte alignment where the run lies inside the row).
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

__device__ __forceinline__ float dpp_wshr1(float v) {   // across the wave: lane i <- lane i-1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   // wave_shr:1
}
__device__ __forceinline__ float dpp_wshl1(float v) {   // across the wave: lane i <- lane i+1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));   // wave_shl:1
}

// RL = lanes per image row inside a wave: 16 (a wave = 4 rows x 16 groups, a block = 16 rows x 64 columns) or 64 (a wave = one
// row of 64 groups = 1 KB per load, a block = 4 rows x 256 columns, blocks numbered so that vertical neighbours share an XCD)
// NB = level buffers: NB - 1 levels are in flight while one is being multiplied.  2 at three waves per SIMD is what ships: 3 buffers
// need 224 registers (two waves per SIMD) and ran 1.53 ms against 1.41 (profiles/r03_backward_final_variants.txt)
template <int RL, int NB>
__device__ __forceinline__ void bwd_final4_body(const float* __restrict__ g, const float* __restrict__ blur,
                                                          const float* __restrict__ sparse, const float* __restrict__ hh,
                                                          const float* __restrict__ ah, const float* __restrict__ a0p,
                                                          const float* __restrict__ gout, float* __restrict__ gg,
                                                          float* __restrict__ gb, int B, int H, int W, int norm) {
    constexpr int N = 24;
    const int W4 = W >> 2;
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    // a block = 16 rows x 16 groups (64 columns): a wave holds 4 rows of 16 groups, so the rows above / below a thread's row are
    // read by the same CU (L1) instead of by another XCD, and the columns beside a group come from the neighbouring lane of
    // the 16-lane DPP row
    const int lane = threadIdx.x & 63, gx = lane & (RL - 1);
    int b, y, xg;
    if (RL == 16) {
        b = blockIdx.z;
        y = blockIdx.y * 16 + ((threadIdx.x >> 6) << 2) + (lane >> 4);
        xg = blockIdx.x * 16 + gx;
    } else {
        // 1-D grid; hardware deals block i to XCD i % 8: give every XCD a contiguous run of tiles, numbered rows-first inside a
        // (image, column-of-blocks) strip, so that the blocks above / below a block run on the same XCD at about the same time
        const int nbx = (W4 + 63) / 64, nby = (H + 3) / 4, ntile = nbx * nby * B, per = (ntile + 7) / 8;
        const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        const bool live = t < ntile && (int)(blockIdx.x >> 3) < per;
        const int tt = live ? t : 0;
        const int by = tt % nby, r = tt / nby, bx = r % nbx;
        b = r / nbx;
        y = live ? by * 4 + (int)(threadIdx.x >> 6) : H;
        xg = bx * 64 + gx;
    }
    const bool valid = y < H && xg < W4;
    const int x = 4 * (valid ? xg : 0);
    const size_t base = (size_t)b * HW, idx = base + (size_t)(valid ? y : 0) * W + x;
    float dW[8][4], dC[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) dW[k][i] = 0.f;
    // One level = A_{t+1} times the three rows of H_t around the thread's row.  All ten loads of a level (A, three H groups, the
    // columns beside the group for the end lanes of a 16-lane row) are issued together and one level AHEAD of the arithmetic, so
    // that a wave always has a level in flight (issued one by one behind their uses, every load paid a full memory round trip).
    // RL == 16: a wave holds four consecutive image rows (16 lanes each), so the H row above / below a lane's row IS the own row of
    // the lane 16 further down / up: only the wave's first / last row of lanes load theirs, the others take it from that lane with
    // ds_bpermute_b32 once the level has arrived -- 1 + 1 + 2 x 1/4 quad loads per lane and level instead of 4 (the pass was
    // running the L1 request path at ~3/4 of its rate: 4 KB of requests per wave and level at 64 B/clk, twelve waves per CU).
    constexpr bool VSHARE = RL == 16;
    const int lrow = lane >> 4;   // (RL == 16) the lane's row inside the wave
    struct Lvl { float4 a, r[3]; float e0[3], e5[3]; };
    // t = 0: H_0 = blur (image order), A_1 = adjoint level N-2;  t = 1..N-2: histories;  t = N-1: A_N = dL/dout (image order)
    auto fetch = [&](int t, Lvl& L, bool first, bool last) {
        const float* ap = last ? gout : ah + (size_t)(N - 2 - t) * total;
        const float* ht = first ? blur : hh + (size_t)(t - 1) * total;
        const bool h_reg_order = !first;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        L.a = z4;
        if (valid) L.a = *reinterpret_cast<const float4*>(ap + idx);
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // dy = 1, 0, -1
            const int yy = y + 1 - d;
            bool rowin = valid && yy >= 0 && yy < H;
            if (VSHARE && ((d == 0 && lrow < 3) || (d == 2 && lrow > 0))) rowin = false;   // comes from the lane 16 down / up
            const float* row = ht + base + (size_t)(rowin ? yy : 0) * W;
            L.r[d] = z4;
            L.e0[d] = L.e5[d] = 0.f;
            if (rowin) L.r[d] = *reinterpret_cast<const float4*>(row + x);
            // c3 of the group to the left sits at position 1 of a register-order group, c0 of the group to the right at position 0
            if (gx == 0 && rowin && x > 0) L.e0[d] = row[h_reg_order ? x - 4 + 1 : x - 1];
            if (gx == RL - 1 && rowin && x + 4 < W) L.e5[d] = row[x + 4];
        }
    };
    auto from_lane = [&](float v, int src_lane) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
    };
    auto compute = [&](Lvl& L, bool first, bool last) {
        if (VSHARE) {   // rows above / below from the neighbouring rows of lanes (an invalid lane holds zeros: outside the image)
            const int dn = (lane + 16) & 63, up = (lane - 16) & 63;
            const float4 o = L.r[1];
            const float4 fd = make_float4(from_lane(o.x, dn), from_lane(o.y, dn), from_lane(o.z, dn), from_lane(o.w, dn));
            const float4 fu = make_float4(from_lane(o.x, up), from_lane(o.y, up), from_lane(o.z, up), from_lane(o.w, up));
            const float e0d = from_lane(L.e0[1], dn), e5d = from_lane(L.e5[1], dn);
            const float e0u = from_lane(L.e0[1], up), e5u = from_lane(L.e5[1], up);
            if (lrow < 3) { L.r[0] = fd; L.e0[0] = e0d; L.e5[0] = e5d; }
            if (lrow > 0) { L.r[2] = fu; L.e0[2] = e0u; L.e5[2] = e5u; }
        }
        const bool a_reg_order = !last, h_reg_order = !first;
        float a[4];
        if (a_reg_order) { a[0] = L.a.x; a[1] = L.a.z; a[2] = L.a.w; a[3] = L.a.y; }   // (c0,c3,c1,c2)
        else { a[0] = L.a.x; a[1] = L.a.y; a[2] = L.a.z; a[3] = L.a.w; }
#pragma unroll
        for (int i = 0; i < 4; ++i) dC[i] += a[i];
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // dy = 1, 0, -1: planes 0..2, 3..4, 5..7
            float h[6];   // columns x-1 .. x+4
            const float4 q = L.r[d];
            if (!h_reg_order) { h[1] = q.x; h[2] = q.y; h[3] = q.z; h[4] = q.w; }
            else { h[1] = q.x; h[2] = q.z; h[3] = q.w; h[4] = q.y; }
            // the columns beside the group belong to the neighbouring lanes (the next / previous group of the same row); the end
            // lanes of a 16-lane row have fetched theirs (0 outside the image)
            h[0] = RL == 16 ? dpp_shr1(h[4]) : dpp_wshr1(h[4]);
            h[5] = RL == 16 ? dpp_shl1(h[1]) : dpp_wshl1(h[1]);
            if (gx == 0) h[0] = L.e0[d];
            if (gx == RL - 1) h[5] = L.e5[d];
            if (x + 4 >= W) h[5] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (d == 0) {
                    dW[0][i] = fmaf(a[i], h[i + 2], dW[0][i]);
                    dW[1][i] = fmaf(a[i], h[i + 1], dW[1][i]);
                    dW[2][i] = fmaf(a[i], h[i], dW[2][i]);
                } else if (d == 1) {
                    dW[3][i] = fmaf(a[i], h[i + 2], dW[3][i]);
                    dW[4][i] = fmaf(a[i], h[i], dW[4][i]);
                } else {
                    dW[5][i] = fmaf(a[i], h[i + 2], dW[5][i]);
                    dW[6][i] = fmaf(a[i], h[i + 1], dW[6][i]);
                    dW[7][i] = fmaf(a[i], h[i], dW[7][i]);
                }
            }
        }
    };
    static_assert(N % NB == 0 && N / NB >= 2, "the level loop rotates NB buffers");
    {
        Lvl L[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) fetch(i, L[i], i == 0, false);
#pragma unroll
        for (int i = 0; i < NB; ++i) {   // levels 0 .. NB - 1
            compute(L[i], i == 0, false);
            fetch(NB + i, L[i], false, NB + i == N - 1);
        }
#pragma unroll 1
        for (int t = NB; t < N - NB; t += NB) {   // L[i] holds level t + i
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                compute(L[i], false, false);
                fetch(t + NB + i, L[i], false, t + NB + i == N - 1);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) compute(L[i], false, i == NB - 1);   // levels N - NB .. N - 1
    }
    if (!valid) return;
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
    // Two passes over the eight planes (the second re-reads its 4-column runs from cache) instead of keeping G, the raw values
    // and dL/dw of all planes in registers: the kernel has to stay at 4+ waves per SIMD to hide its loads.
    // G_k(p) = g~_k(p + off_k): a run of four columns of plane k in row y + dy_k starting at x + dx_k, zero outside the image
    auto run = [&](int k, float (&v)[4]) -> bool {
        const int yy = y + dy2(k), xs = x + dx2(k);
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (yy < 0 || yy >= H) return false;
        const float* src = gbp + k * HW + (size_t)yy * W;
        if (xs >= 0 && xs + 3 < W) {
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
    for (int k = 0; k < 8; ++k) {
        float v[4];
        run(k, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float G = norm == CSPN_NORM_8SUM_ABS ? fabsf(v[i]) : v[i];
            S[i] += fabsf(v[i]);
            gs[i] += G;
            T1[i] = fmaf(om[i] * (dW[k][i] - ch[i]), G, T1[i]);
        }
    }
    float rS[4], t2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rS[i] = 1.f / S[i]; t2[i] = T1[i] / (S[i] * S[i]); }
    if (gb) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = a0[i] + dC[i] * (om[i] * (1.f - gs[i] / S[i]) + m[i]);
        *reinterpret_cast<float4*>(gb + idx) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (ggp) {
        // g_k(q) with q - off_k outside the image is read by no pixel (the gather sees the zero padding instead): gradient 0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ys = y - dy2(k);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xq = x + i - dx2(k);
                if (ys < 0 || ys >= H || xq < 0 || xq >= W) ggp[k * HW + (size_t)y * W + x + i] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v[4];
            if (!run(k, v)) continue;  // the zero padding is a constant
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
            if (xs >= 0 && xs + 3 < W) st4u(dst + xs, make_float4(d[0], d[1], d[2], d[3]));
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xs + i >= 0 && xs + i < W) dst[xs + i] = d[i];
            }
        }
    }
}

template <int RL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void bwd_final4_kernel(const float* __restrict__ g, const float* __restrict__ blur,
                                                          const float* __restrict__ sparse, const float* __restrict__ hh,
                                                          const float* __restrict__ ah, const float* __restrict__ a0p,
                                                          const float* __restrict__ gout, float* __restrict__ gg,
                                                          float* __restrict__ gb, int B, int H, int W, int norm) {
    bwd_final4_body<RL, 2>(g, blur, sparse, hh, ah, a0p, gout, gg, gb, B, H, W, norm);
}

}  // namespace

// final pass of the assembly-sweep backward; CSPN_BWD_FINAL_RL=64 selects the one-row-per-wave mapping (A/B)
static void launch_final4(const float* g, const float* blur, const float* sparse, const float* hh, const float* ah, const float* a0,
                          const float* gout, float* gg, float* gb, int B, int H, int W, int norm, hipStream_t st) {
    static const bool rl64 = [] { const char* e = getenv("CSPN_BWD_FINAL_RL"); return e && atoi(e) == 64; }();
    if (rl64) {
        const int nbx = (W / 4 + 63) / 64, nby = (H + 3) / 4, ntile = nbx * nby * B, per = (ntile + 7) / 8;
        hipLaunchKernelGGL(bwd_final4_kernel<64>, dim3(per * 8), dim3(256), 0, st, g, blur, sparse, hh, ah, a0, gout, gg, gb, B, H, W, norm);
    } else {
        const dim3 grid((W / 4 + 15) / 16, (H + 15) / 16, B);
        hipLaunchKernelGGL(bwd_final4_kernel<16>, grid, dim3(256), 0, st, g, blur, sparse, hh, ah, a0, gout, gg, gb, B, H, W, norm);
    }
}

constexpr size_t FRONT_PAD = 65536;  // bytes kept addressable in front of the folded planes (the adjoint sweep reads plane 0
                                     // one row up and one pixel left of its first row)
static bool asm_path(int B, int H, int W, int n_iter) {
    return n_iter == 24 && tsw2d_supported(B, H, W) && 4ull * W + 16 <= FRONT_PAD &&
           (unsigned long long)B * H * W * 32ull < (1ull << 32);  // 8 coefficient planes of per-lane byte offsets
}

size_t backward2d_workspace(int B, int H, int W, int n_iter) {
    const size_t total = (size_t)B * H * W;
    if (asm_path(B, H, W, n_iter))  // forward levels 23 + folded coefficients 8 + adjoint levels 23 + A_0 + a scratch output
        return FRONT_PAD + (size_t)(23 + 8 + 23 + 2) * total * sizeof(float) + 256;
    return (size_t)(9 + 8 + (n_iter > 0 ? n_iter - 1 : 0) + n_iter) * total * sizeof(float);
}

int backward2d(const float* g, const float* blur, const float* sparse, const float* gout, float* gg, float* gb, int B, int H,
               int W, int n_iter, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    float* wf = (float*)ws;
    if (asm_path(B, H, W, n_iter)) {
        // both sweeps run in the fused ring kernel (cspn2d_tsw.hip), each writing its 23 intermediate levels: the forward
        // as it is, the adjoint as a propagation whose coefficients are the folded planes read neighbour-sited with the
        // channel order reversed (generator option adj)
        // the forward sweep also leaves the 8 folded coefficient planes right behind its 23 level planes
        float* hh = (float*)((char*)ws + FRONT_PAD);
        wf = hh + 23 * total;
        float* ah = wf + 8 * total;
        float* a0 = ah + 23 * total;
        float* scratch = a0 + total;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (int e = tsw2d_pass(g, blur, blur, sparse, scratch, B, H, W, norm, st, hh)) return e;
        if (int e = tsw2d_adjoint_pass(wf, gout, a0, B, H, W, st, ah)) return e;
        static const bool final1 = getenv("CSPN_BWD_FINAL1") != nullptr;   // A/B switch: the one-pixel-per-thread final pass
        if (final1)
            hipLaunchKernelGGL(bwd_final_kernel<true>, dim3(blocks), dim3(256), 0, st, g, blur, sparse, hh, ah, a0, gout, gg, gb, B,
                               H, W, n_iter, norm);
        else
            launch_final4(g, blur, sparse, hh, ah, a0, gout, gg, gb, B, H, W, norm, st);
        return check_launch("bwd_final_kernel");
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
    hipLaunchKernelGGL(bwd_final_kernel<false>, dim3(blocks), dim3(256), 0, st, g, blur, sparse, hh, ah, nullptr, gout, gg, gb, B,
                       H, W, n_iter, norm);
    return check_launch("bwd_final_kernel");
}

// ---- training mode: the forward keeps its level history, the backward starts from it ------------------------------------
// history = [FRONT_PAD bytes][H_1 .. H_23][w'_0 .. w'_7] (what the forward sweep of backward2d leaves behind)
size_t history2d_bytes(int B, int H, int W, int n_iter) {
    return asm_path(B, H, W, n_iter) ? FRONT_PAD + (size_t)(23 + 8) * B * H * W * sizeof(float) : 0;
}

int forward2d_history(const float* g, const float* blur, const float* sparse, float* out, void* history, int B, int H, int W,
                      int norm, void* ws, hipStream_t st) {
    float* hh = (float*)((char*)history + FRONT_PAD);
    (void)ws;
    return tsw2d_pass(g, blur, blur, sparse, out, B, H, W, norm, st, hh);
}

size_t backward2d_history_workspace(int B, int H, int W) {
    return (size_t)(23 + 1) * B * H * W * sizeof(float) + 256;
}

int backward2d_history(const float* g, const float* blur, const float* sparse, const float* gout, const void* history, float* gg,
                       float* gb, int B, int H, int W, int norm, void* ws, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    const float* hh = (const float*)((const char*)history + FRONT_PAD);
    const float* wf = hh + 23 * total;
    float* ah = (float*)ws;
    float* a0 = ah + 23 * total;
    if (int e = tsw2d_adjoint_pass(wf, gout, a0, B, H, W, st, ah)) return e;
    launch_final4(g, blur, sparse, hh, ah, a0, gout, gg, gb, B, H, W, norm, st);
    return check_launch("bwd_final4_kernel");
}

}  // namespace cspn
