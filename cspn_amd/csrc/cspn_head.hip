// cspn_head.hip -- the producer of the propagation's inputs (the producer half): the two heads
// Simple_Gudi_UpConv_Block_Last_Layer of the reference backbone (cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206, instantiated
// :318-319 as gud_up_proj_layer5 (64 -> 1: blur depth) and gud_up_proj_layer6 (64 -> 8: guidance), called :372-373) as ONE kernel:
//     Unpool (:41-54: x[i][j] -> U[2i][2j], zeros elsewhere; narrowed to oheight x owidth :196-201)  ->  3x3 conv, padding 1, no bias (:190)
// and, optionally, the propagation's own first step fused behind it: affinity_normalization (cspn.py:85-144) of the 8 guidance channels, i.e.
// the kernel emits gate_wb -- the CSPN_NORM_PRENORM input contract of the forward -- so that no stand-alone normalisation pass exists.
//
// Three quarters of the unpooled taps are structurally zero: an input pixel (i, j) owns the 2 x 2 output block (2i + a, 2j + b) and
//     out[2i  ][2j  ] = W11 x00                      out[2i  ][2j+1] = W10 x00 + W12 x01
//     out[2i+1][2j  ] = W01 x00 + W21 x10            out[2i+1][2j+1] = W00 x00 + W02 x01 + W20 x10 + W22 x11
// (x00 = x[i][j], x01 = x[i][j+1], x10 = x[i+1][j], x11 = x[i+1][j+1]; Wyx = the 3x3 kernel): 9 products per input pixel, channel and output channel
// instead of 36.  One thread = one input pixel, all 9 output channels: per input channel 4 loads and, with the weights packed as the pairs
// (W11,W00) (W10,W01) (W12,W21) (W02,W20) W22, four v_pk_fma_f32 with an SGPR-pair weight operand + one v_fma_f32 per output channel -- 45 vector
// instructions for 81 FMAs.  fp32 throughout (the reference's conv is fp32; tolerance of the parity tests 1e-4 relative).
//
// Fused normalisation (MODE 1 / 2): a workgroup of 4 waves marches down a stripe of 62 owned input columns (lanes 1..62; lanes 0 and 63 compute the
// halo columns, 3 % redundant work), four input rows = eight output rows per iteration; the raw guidance of the last 10 output rows lives in an
// LDS ring, and an iteration normalises the eight rows ending one row above its newest (whose lower neighbour exists by then): no vertical
// recomputation inside a chunk of rows, one extra row at a chunk's top.
#include <cstdint>

#include "cspn_common.h"

namespace cspn {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int HT = 256;        // 4 waves: 4 input rows per iteration
constexpr int OWN = 62;        // owned input columns of a stripe
constexpr int LC = 128;        // output columns a stripe computes (2 per lane)
constexpr int RING = 10;       // output rows of raw guidance kept in LDS: an iteration reads rows [Y0 - 2, Y0 + 7] while none of them is overwritten

// packed weights: [c][o = 0..8][5] pairs; o < 8: guidance channel o (w6 [8][C][3][3]), o = 8: blur (w5 [1][C][3][3], zeros if absent)
__global__ __launch_bounds__(256) void head_pack_kernel(const float* __restrict__ w6, const float* __restrict__ w5, f2* __restrict__ wp, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * 9) return;
    const int c = idx / 9, o = idx - c * 9;
    const float* w = o < 8 ? w6 + ((size_t)o * C + c) * 9 : (w5 ? w5 + (size_t)c * 9 : nullptr);
    auto W = [&](int ky, int kx) { return w ? w[ky * 3 + kx] : 0.f; };
    f2* d = wp + (size_t)idx * 5;
    d[0] = f2{W(1, 1), W(0, 0)};
    d[1] = f2{W(1, 0), W(0, 1)};
    d[2] = f2{W(1, 2), W(2, 1)};
    d[3] = f2{W(0, 2), W(2, 0)};
    d[4] = f2{W(2, 2), 0.f};
}

// MODE 0: raw guidance [B][8][H][W] + blur; 1: gate_wb of '8sum'; 2: gate_wb of '8sum_abs'
template <int MODE>
__global__ __launch_bounds__(HT) void head_kernel(const float* __restrict__ x, const f2* __restrict__ wp, float* __restrict__ gout,
                                                   float* __restrict__ bout, int C, int h, int w, int H, int W, int nstripe, int nchunk,
                                                   int rows_per_chunk) {
    __shared__ float ring[MODE ? 8 * RING * LC : 1];   // [k][output row mod 10][column of the stripe]
    int bid = blockIdx.x;
    const int ch = bid % nchunk;
    bid /= nchunk;
    const int s = bid % nstripe, b = bid / nstripe;
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = s * OWN - 1 + lane;                      // this lane's input column (lanes 0 / 63: the halo columns)
    const int i_begin = ch * rows_per_chunk, i_end = (i_begin + rows_per_chunk < h) ? i_begin + rows_per_chunk : h;
    if (i_begin >= h) return;
    const bool col_ok = j >= 0 && j < w && 2 * j < W;       // (narrow: unpooled columns >= W are cut before the conv)
    const bool own = lane >= 1 && lane <= OWN && j < w;
    const size_t hw = (size_t)h * w, HWo = (size_t)H * W;
    const float* xb = x + (size_t)b * C * hw;
    const int X0 = 2 * j;                                   // output columns X0, X0 + 1
    for (int i0 = (MODE && i_begin > 0) ? i_begin - 4 : i_begin; i0 < i_end; i0 += 4) {
        const int i = i0 + ty;
        const bool row_in = i < i_end && (i >= i_begin || (MODE && i == i_begin - 1));
        f2 U[9], V[9], T[9];
#pragma unroll
        for (int o = 0; o < 9; ++o) { U[o] = f2{0.f, 0.f}; V[o] = f2{0.f, 0.f}; T[o] = f2{0.f, 0.f}; }
        if (row_in && col_ok && 2 * i < H) {
            const bool r1 = i + 1 < h && 2 * (i + 1) < H, c1 = j + 1 < w && 2 * (j + 1) < W;
            const float* p = xb + (size_t)i * w + j;
            const f2* wc = wp;
            for (int c = 0; c < C; ++c, p += hw, wc += 45) {
                const float x00 = p[0];
                const float x01 = c1 ? p[1] : 0.f;
                const float x10 = r1 ? p[w] : 0.f;
                const float x11 = (r1 && c1) ? p[w + 1] : 0.f;
                const f2 xa = f2{x00, x00}, xc = f2{x01, x10};
#pragma unroll
                for (int o = 0; o < 9; ++o) {
                    U[o] = __builtin_elementwise_fma(wc[o * 5 + 0], xa, U[o]);   // (p00, p11a) += (W11, W00) x00
                    V[o] = __builtin_elementwise_fma(wc[o * 5 + 1], xa, V[o]);   // (p01, p10) += (W10, W01) x00
                    V[o] = __builtin_elementwise_fma(wc[o * 5 + 2], xc, V[o]);   //             += (W12 x01, W21 x10)
                    T[o] = __builtin_elementwise_fma(wc[o * 5 + 3], xc, T[o]);   // (p11b, p11c) += (W02 x01, W20 x10)
                    U[o].y = __builtin_fmaf(wc[o * 5 + 4].x, x11, U[o].y);      //  p11a += W22 x11
                }
            }
        }
        const int Y0 = 2 * i;
        const bool st = own && row_in && i >= i_begin;      // this thread's block is this workgroup's to store
        // blur (o = 8) and, MODE 0, the raw guidance: straight to memory
        if (st) {
#pragma unroll
            for (int o = (MODE ? 8 : 0); o < 9; ++o) {
                float* dst = o < 8 ? gout + ((size_t)b * 8 + o) * HWo : (bout ? bout + (size_t)b * HWo : nullptr);
                if (!dst) continue;
                const float p00 = U[o].x, p01 = V[o].x, p10 = V[o].y, p11 = U[o].y + T[o].x + T[o].y;
                if (Y0 < H) {
                    if (X0 < W) dst[(size_t)Y0 * W + X0] = p00;
                    if (X0 + 1 < W) dst[(size_t)Y0 * W + X0 + 1] = p01;
                }
                if (Y0 + 1 < H) {
                    if (X0 < W) dst[(size_t)(Y0 + 1) * W + X0] = p10;
                    if (X0 + 1 < W) dst[(size_t)(Y0 + 1) * W + X0 + 1] = p11;
                }
            }
        }
        if (MODE) {
            // raw guidance of this thread's 2 x 2 block -> ring (zeros outside the image: the reference pads the affinity planes with zeros)
            if (row_in) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const float p00 = U[o].x, p01 = V[o].x, p10 = V[o].y, p11 = U[o].y + T[o].x + T[o].y;
                    const bool xin0 = j >= 0 && X0 < W, xin1 = j >= 0 && X0 + 1 < W;
                    float* r0 = ring + ((size_t)o * RING + ((Y0 + RING) % RING)) * LC + 2 * lane;
                    float* r1 = ring + ((size_t)o * RING + ((Y0 + 1 + RING) % RING)) * LC + 2 * lane;
                    *(f2*)r0 = f2{(xin0 && Y0 < H) ? p00 : 0.f, (xin1 && Y0 < H) ? p01 : 0.f};
                    *(f2*)r1 = f2{(xin0 && Y0 + 1 < H) ? p10 : 0.f, (xin1 && Y0 + 1 < H) ? p11 : 0.f};
                }
            }
            __syncthreads();
            // normalise the rows whose lower neighbour exists now: [2 i0 - 1, 2 i0 + 6], clipped to what this chunk owns
            const int YB = 2 * i0;
            int lo = YB - 1, hi = YB + 6;
            const int own_lo = i_begin > 0 ? 2 * i_begin - 1 : 0;
            if (lo < own_lo) lo = own_lo;
            if (i0 + 4 >= i_end) hi = (i_end == h) ? H - 1 : 2 * i_end - 2;
            if (hi > H - 1) hi = H - 1;
            const int lx = threadIdx.x & (LC - 1), rg = threadIdx.x >> 7;     // 128 columns x 2 row groups of 4
            const int X = 2 * (s * OWN - 1) + lx;
            const bool xs = lx >= 2 && lx < 2 + 2 * OWN && X < W;
            if (xs) {
                for (int q = 0; q < 4 + rg; ++q) {   // row group 0: rows YB - 1 .. YB + 2; group 1: YB + 3 .. YB + 7 (the last one only at the image's bottom)
                    const int Y = YB - 1 + rg * 4 + q;
                    if (Y < lo || Y > hi) continue;
                    float G[8], S = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int yy = Y + dy2(k), xx = lx + dx2(k);
                        float v = 0.f;
                        if (yy >= 0 && yy < H) v = ring[((size_t)k * RING + ((yy + RING) % RING)) * LC + xx];
                        if (MODE == 2) v = fabsf(v);
                        G[k] = v;
                        S += fabsf(v);
                    }
                    const float r = 1.0f / S;                                                                // 0 * (1 / 0) = NaN where the reference has 0 / 0 (cspn.py:138)
#pragma unroll
                    for (int k = 0; k < 8; ++k) gout[(((size_t)b * 8 + k) * H + Y) * W + X] = G[k] * r;
                }
            }
            __syncthreads();
        }
    }
}

// MODE 0 (raw guidance + blur) only.  One thread = one column of TWO input rows (A = i0 = 2 * pair, B = i0 + 1: feature rows i0 .. i0 + 2).  A packed FMA
// carries the SAME tap for both rows -- accumulator pairs (p_A, p_B), the weight a scalar operand broadcast to both halves -- so the 9 non-zero products
// per (input pixel, channel, output plane) are exactly 9 packed lanes: 81 v_pk_fma_f32 per channel and wave for 128 input pixels, 72 accumulator registers.
//   p00 = W11 x00            p01 = W10 x00 + W12 x01            p10 = W01 x00 + W21 x10            p11 = W00 x00 + W02 x01 + W20 x10 + W22 x11
// with x00 = the pixel, x01 its right, x10 its lower, x11 its lower right neighbour (zero beyond the image / the narrowed output).
//   * The right neighbour comes from the neighbouring lane (DPP wave_shl:1; lane 63 only serves lane 62: segments of 63 owned columns).
//   * The feature map is read ONCE, from HBM: ~2 us of latency per access against ~0.2 us of arithmetic per channel.  A wave keeps the three feature rows of
//     the NEXT 8 channels on their way into its own LDS slots by LDS-DMA (global_load_lds_dword: no VGPR in flight, no other wave involved, hence no barrier):
//     slot c % 8 is requested again right after channel c was consumed, and waited for -- a counted s_waitcnt vmcnt -- 8 channels later.
//   * Every mask lives in a READ ADDRESS: each slot has a fourth row that nothing ever writes (zeros); a lane whose column, or a wave whose row, is outside
//     reads that row.  Nothing but 3 address adds, 4 LDS reads, 4 DPP moves and the 81 FMAs runs per channel.
//   * A channel's 81 weights are scalar loads ([c][o][ky][kx], 84-float records): with the loop state they fit the 102 SGPRs without spills.  (The first
//     versions of this kernel -- one row per wave, masks as selects, 90 weight dwords -- spent a third of their issue slots on v_readlane / v_cndmask around
//     spilled scalars: 1.09 ms where the arithmetic alone took 0.69, profiles/r06_head.md.)
#ifndef HEAD_ABL
#define HEAD_ABL 0   // timing builds only (tools/r06/build_abl_head.sh): 1 = no feature reads, 2 = one channel's weights for all, 4 = 4 of the 9 output planes
#endif
constexpr int RAW_DEPTH = 8;
constexpr int WREC = 84;     // floats per channel in the raw kernel's weight records (81 + pad: 16-byte multiples)
__global__ __launch_bounds__(256) void head_rawpack_kernel(const float* __restrict__ w6, const float* __restrict__ w5, float* __restrict__ wr, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * WREC) return;
    const int c = idx / WREC, r = idx - c * WREC, o = r / 9, t = r - o * 9;
    wr[idx] = r >= 81 ? 0.f : o < 8 ? w6[((size_t)o * C + c) * 9 + t] : (w5 ? w5[(size_t)c * 9 + t] : 0.f);
}

__global__ __launch_bounds__(256) void head_raw_kernel(const float* __restrict__ x, const float* __restrict__ wr, float* __restrict__ gout,
                                                        float* __restrict__ bout, int C, int h, int w, int H, int W, int B) {
    __shared__ float xs[4][RAW_DEPTH][4][64];            // [wave][slot][feature row i0, i0 + 1, i0 + 2, zeros][lane]
    const int wq = (w + 62) / 63, hp = (h + 1) / 2;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup ids go round the 8 XCDs: each XCD takes a contiguous eighth of the units, so that the row pairs which share a feature row (i0 + 2 of one =
    // i0 of the next) meet in one L2
    const int per_xcd = gridDim.x >> 3;
    const int unit = (((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3)) * 4 + wv;   // (wave-uniform, in scalar registers: so is everything derived from it)
    const int seg = unit % wq;
    const int i0 = 2 * ((unit / wq) % hp), b = unit / (wq * hp);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int j = seg * 63 + lane;
    const size_t hw = (size_t)h * w, HWo = (size_t)H * W;
    f2 P00[9], P01[9], P10[9], P11[9];                   // (.x: row A, .y: row B)
#pragma unroll
    for (int o = 0; o < 9; ++o) { P00[o] = P01[o] = P10[o] = P11[o] = f2{0.f, 0.f}; }
    if (2 * i0 < H) {
        const bool c0k = j < w && 2 * j < W;
        const bool r1 = i0 + 1 < h && 2 * (i0 + 1) < H, r2 = i0 + 2 < h && 2 * (i0 + 2) < H;
#pragma unroll
        for (int sl = 0; sl < RAW_DEPTH; ++sl) xs[wv][sl][3][lane] = 0.f;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&xs[wv][0][0][0]);
        const unsigned la = lds0 + 4u * lane;
        const unsigned va0 = la + (c0k ? 0u : 768u), va1 = la + (c0k && r1 ? 256u : 768u), va2 = la + (c0k && r2 ? 512u : 768u);
        const unsigned voff = 4u * (unsigned)(j < w ? j : w - 1);             // the lane's column, the same for every row and channel
        const float* row0 = x + (size_t)b * C * hw + (size_t)i0 * w;          // scalar: the row's address
        const unsigned d1 = i0 + 1 < h ? 4u * w : 0u, d2 = i0 + 2 < h ? 4u * w : 0u;   // (rows beyond the image: the row before again; never read back)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto request = [&](int c) {
            const unsigned dst = lds0 + (unsigned)(c % RAW_DEPTH) * 1024u;
            const float* q0 = row0 + (size_t)c * hw;
            const float* q1 = (const float*)((const char*)q0 + d1);
            const float* q2 = (const float*)((const char*)q1 + d2);
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2\n\t"
                         "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %0, %3"
                         :: "v"(voff), "s"(q0), "s"(q1), "s"(q2), "s"(dst) : "memory");
        };
        auto lds = [&](unsigned a) { return *(const volatile __attribute__((address_space(3))) float*)a; };
        auto right = [&](float t) {   // the value of the lane to the right (column j + 1); lane 63: 0, it owns no output
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
        };
        const int pre = C < RAW_DEPTH ? C : RAW_DEPTH;
        if (!(HEAD_ABL & 1)) for (int c = 0; c < pre; ++c) request(c);
        for (int c = 0; c < C; ++c) {
            if (HEAD_ABL & 1) ;
            else if (c + RAW_DEPTH <= C) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RAW_DEPTH - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the last RAW_DEPTH - 1 channels: the queue drains)
            float u0, u1, u2;
            if (HEAD_ABL & 1) { u0 = (float)(lane + c); u1 = (float)(lane ^ c); u2 = (float)(lane - c); }
            else {
                const unsigned so = (unsigned)(c % RAW_DEPTH) * 1024u;
                u0 = lds(va0 + so); u1 = lds(va1 + so); u2 = lds(va2 + so);
            }
            const float e0 = right(u0), e1 = right(u1), e2 = right(u2);
            const f2 xa = f2{u0, u1}, xb = f2{u1, u2}, xr = f2{e0, e1}, xd = f2{e1, e2};    // x00, x10, x01, x11 of rows (A, B)
            const float* wc = wr + (size_t)((HEAD_ABL & 2) ? 0 : c) * WREC;
#pragma unroll
            for (int o = 0; o < ((HEAD_ABL & 4) ? 4 : 9); ++o) {
                const float* k = wc + o * 9;
                P00[o] = __builtin_elementwise_fma(f2{k[4], k[4]}, xa, P00[o]);
                P01[o] = __builtin_elementwise_fma(f2{k[3], k[3]}, xa, P01[o]);
                P01[o] = __builtin_elementwise_fma(f2{k[5], k[5]}, xr, P01[o]);
                P10[o] = __builtin_elementwise_fma(f2{k[1], k[1]}, xa, P10[o]);
                P10[o] = __builtin_elementwise_fma(f2{k[7], k[7]}, xb, P10[o]);
                P11[o] = __builtin_elementwise_fma(f2{k[0], k[0]}, xa, P11[o]);
                P11[o] = __builtin_elementwise_fma(f2{k[2], k[2]}, xr, P11[o]);
                P11[o] = __builtin_elementwise_fma(f2{k[6], k[6]}, xb, P11[o]);
                P11[o] = __builtin_elementwise_fma(f2{k[8], k[8]}, xd, P11[o]);
            }
            if (!(HEAD_ABL & 1) && c + RAW_DEPTH < C) request(c + RAW_DEPTH);   // (its slot's values are in registers: the FMAs above consumed the reads)
        }
    }
    if (j >= w || lane == 63 || 2 * j >= W) return;
    const int X0 = 2 * j;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        float* dst = o < 8 ? gout + ((size_t)b * 8 + o) * HWo : (bout ? bout + (size_t)b * HWo : nullptr);
        if (!dst) continue;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int Y0 = 2 * (i0 + half);
            if (i0 + half >= h) continue;
            const float p00 = half ? P00[o].y : P00[o].x, p01 = half ? P01[o].y : P01[o].x, p10 = half ? P10[o].y : P10[o].x, p11 = half ? P11[o].y : P11[o].x;
            if (Y0 < H) {
                dst[(size_t)Y0 * W + X0] = p00;
                if (X0 + 1 < W) dst[(size_t)Y0 * W + X0 + 1] = p01;
            }
            if (Y0 + 1 < H) {
                dst[(size_t)(Y0 + 1) * W + X0] = p10;
                if (X0 + 1 < W) dst[(size_t)(Y0 + 1) * W + X0 + 1] = p11;
            }
        }
    }
}

}  // namespace

// packed weights of the fused kernel ([c][o][5] pairs), then the raw kernel's records ([c][84] floats)
static size_t head_pack_bytes(int C) { return ((size_t)C * 9 * 5 * sizeof(f2) + 255) & ~(size_t)255; }
size_t head_workspace(int C) { return head_pack_bytes(C) + (((size_t)C * WREC * sizeof(float) + 255) & ~(size_t)255); }

// mode 0: raw guidance; 1 / 2: gate_wb of '8sum' / '8sum_abs'
int head_forward(const float* x, const float* w6, const float* w5, float* gout, float* bout, int B, int C, int h, int w, int H, int W, int mode,
                 void* ws, hipStream_t st) {
    f2* wp = (f2*)ws;
    if (mode == 0) {
        float* wr = (float*)((char*)ws + head_pack_bytes(C));
        hipLaunchKernelGGL(head_rawpack_kernel, dim3((C * WREC + 255) / 256), dim3(256), 0, st, w6, w5, wr, C);
        const long long units = (long long)B * ((h + 1) / 2) * ((w + 62) / 63);
        const long long groups = ((units + 3) / 4 + 7) / 8 * 8;   // (a multiple of 8: see the kernel's XCD mapping; the spare waves return at once)
        hipLaunchKernelGGL(head_raw_kernel, dim3((unsigned)groups), dim3(256), 0, st, x, wr, gout, bout, C, h, w, H, W, B);
        return check_launch("head_raw_kernel");
    }
    hipLaunchKernelGGL(head_pack_kernel, dim3((C * 9 + 255) / 256), dim3(256), 0, st, w6, w5, wp, C);
    const int nstripe = (w + OWN - 1) / OWN;
    // chunks of rows (multiples of 4) so that a few thousand workgroups exist: 4 waves each, several per CU, balanced to ~1 %
    int rows = 32;
    while ((long long)B * nstripe * ((h + rows - 1) / rows) < 2048 && rows > 8) rows /= 2;
    const int nchunk = (h + rows - 1) / rows;
    const dim3 grid((unsigned)(B * nstripe * nchunk));
    switch (mode) {
        case 1: hipLaunchKernelGGL(head_kernel<1>, grid, dim3(HT), 0, st, x, wp, gout, bout, C, h, w, H, W, nstripe, nchunk, rows); break;
        default: hipLaunchKernelGGL(head_kernel<2>, grid, dim3(HT), 0, st, x, wp, gout, bout, C, h, w, H, W, nstripe, nchunk, rows); break;
    }
    return check_launch("head_kernel");
}

}  // namespace cspn
