// cspn_head.hip -- the producer of the propagation's inputs (the producer half): the two heads
// Simple_Gudi_UpConv_Block_Last_Layer of the reference backbone (cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206, instantiated
// :318-319 as gud_up_proj_layer5 (64 -> 1: blur depth) and gud_up_proj_layer6 (64 -> 8: guidance), called :372-373) as ONE kernel:
//     Unpool (:41-54: x[i][j] -> U[2i][2j], zeros elsewhere; narrowed to oheight x owidth :196-201)  ->  3x3 conv, padding 1, no bias (:190)
// and, optionally, the propagation's own first step behind it: affinity_normalization (cspn.py:85-144) of the 8 guidance channels, i.e. the call
// emits gate_wb -- the CSPN_NORM_PRENORM input contract of the forward.
//
// Three quarters of the unpooled taps are structurally zero: an input pixel (i, j) owns the 2 x 2 output block (2i + a, 2j + b) and
//     out[2i  ][2j  ] = W11 x00                      out[2i  ][2j+1] = W10 x00 + W12 x01
//     out[2i+1][2j  ] = W01 x00 + W21 x10            out[2i+1][2j+1] = W00 x00 + W02 x01 + W20 x10 + W22 x11
// (x00 = x[i][j], x01 = x[i][j+1], x10 = x[i+1][j], x11 = x[i+1][j+1]; Wyx = the 3x3 kernel): 9 products per input pixel, channel and output channel
// instead of 36.  fp32 throughout (the reference's conv is fp32; tolerance of the parity tests 1e-5 of the plane maximum).
//
// gate_wb (norm_type 8SUM / 8SUM_ABS) in two steps on the output tensor itself: head_raw_kernel<SITED> stores guidance plane k CONSUMER-SITED -- the value it
// computed for pixel q goes to p = q - off_k, where the propagation reads it (G_k(p) = g_k(p + off_k), cspn.py:105-128) --, after which the normalisation
// is per pixel and runs in place (head_norm_sited_kernel: 8 planes read, 8 written; positions whose neighbour lies outside the image were never
// stored: they count as 0).  Round 6's first version fused the normalisation through an LDS ring of raw output rows inside the conv kernel: 1.86 ms at
// KITTI x 64 against 0.81 + 0.3 ms for these two (profiles/r06_head.md).
#include <cstdint>

#include "cspn_common.h"

namespace cspn {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// One thread = one column of TWO input rows (A = i0 = 2 * pair, B = i0 + 1: feature rows i0 .. i0 + 2).  A packed FMA
// carries the SAME tap for both rows -- accumulator pairs (p_A, p_B), the weight a scalar operand broadcast to both halves -- so the 9 non-zero products
// per (input pixel, channel, output plane) are exactly 9 packed lanes: 81 v_pk_fma_f32 per channel and wave for 128 input pixels, 72 accumulator registers.
//   p00 = W11 x00            p01 = W10 x00 + W12 x01            p10 = W01 x00 + W21 x10            p11 = W00 x00 + W02 x01 + W20 x10 + W22 x11
// with x00 = the pixel, x01 its right, x10 its lower, x11 its lower right neighbour (zero beyond the image / the narrowed output).
//   * The right neighbour comes from the neighbouring lane (DPP wave_shl:1; lane 63 only serves lane 62: segments of 63 owned columns).
//   * The feature map is read ONCE, from HBM: ~2 us of latency per access against ~0.2 us of arithmetic per channel.  A wave keeps the three feature rows of
//     the NEXT 4 channels on their way into its own LDS slots by LDS-DMA (global_load_lds_dword: no VGPR in flight, no other wave involved, hence no barrier):
//     slot c % 4 is requested again right after channel c was consumed, and waited for -- a counted s_waitcnt vmcnt -- 4 channels later.
//   * Every mask lives in a READ ADDRESS: each slot has a fourth row that nothing ever writes (zeros); a lane whose column, or a wave whose row, is outside
//     reads that row.  Nothing but 3 address adds, 4 LDS reads, 4 DPP moves and the 81 FMAs runs per channel.
//   * A channel's 81 weights are scalar loads ([c][o][ky][kx], 84-float records): with the loop state they fit the 102 SGPRs without spills.  (The first
//     versions of this kernel -- one row per wave, masks as selects, 90 weight dwords -- spent a third of their issue slots on v_readlane / v_cndmask around
//     spilled scalars: 1.09 ms where the arithmetic alone took 0.69, profiles/r06_head.md.)
#ifndef HEAD_ABL
#define HEAD_ABL 0   // timing builds only (tools/r06/build_abl_head.sh): 1 = no feature reads, 2 = one channel's weights for all, 4 = 4 of the 9 output planes, 8 = every feature request reads channel 0 or 1 (from L2), 16 = nothing stored, 32 = LDS reads of stale slots without requests, 64 = requests without LDS reads
#endif
#ifndef HEAD_DEPTH
#define HEAD_DEPTH 4     // channels in flight per wave: 2, 3, 4, 6, 8 measure the same within 1.5 % (4 and 3 best): the feed is a throughput cost, not a latency
                         // (A/B builds: tools/r06/build_abl_head.sh 0 "-DHEAD_DEPTH=8" d8)
#endif
#ifndef HEAD_OCC
#define HEAD_OCC 5       // workgroups (= waves per SIMD) the register budget is set for
#endif
constexpr int RAW_DEPTH = HEAD_DEPTH;
constexpr int WREC = 84;     // floats per channel in the raw kernel's weight records (81 + pad: 16-byte multiples)
__global__ __launch_bounds__(256) void head_rawpack_kernel(const float* __restrict__ w6, const float* __restrict__ w5, float* __restrict__ wr, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * WREC) return;
    const int c = idx / WREC, r = idx - c * WREC, o = r / 9, t = r - o * 9;
    wr[idx] = r >= 81 ? 0.f : o < 8 ? w6[((size_t)o * C + c) * 9 + t] : (w5 ? w5[(size_t)c * 9 + t] : 0.f);
}

template <bool SITED>
__global__ __launch_bounds__(256, HEAD_OCC) void head_raw_kernel(const float* __restrict__ x, const float* __restrict__ wr, float* __restrict__ gout,
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
        auto request = [&](int c, unsigned slot_off) {
            const unsigned dst = lds0 + slot_off;
            const float* q0 = row0 + (size_t)((HEAD_ABL & 8) ? (c & 1) : c) * hw;   // (8: every request reads channel 0 / 1: L2-resident)
            const float* q1 = (const float*)((const char*)q0 + d1);
            const float* q2 = (const float*)((const char*)q1 + d2);
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2\n\t"
                         "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %0, %3"
                         :: "v"(voff), "s"(q0), "s"(q1), "s"(q2), "s"(dst) : "memory", "m0", "scc");
        };
        auto lds = [&](unsigned a) { return *(const volatile __attribute__((address_space(3))) float*)a; };
        auto right = [&](float t) {   // the value of the lane to the right (column j + 1); lane 63: 0, it owns no output
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
        };
        const int pre = C < RAW_DEPTH ? C : RAW_DEPTH;
        if (!(HEAD_ABL & (1 | 32))) for (int c = 0; c < pre; ++c) request(c, (unsigned)c * 1024u);
        unsigned so = 0;                                                          // slot of channel c: (c mod RAW_DEPTH) KiB
        for (int c = 0; c < C; ++c) {
            if (HEAD_ABL & (1 | 32)) ;
            else if (c + RAW_DEPTH <= C) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (RAW_DEPTH - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the last RAW_DEPTH - 1 channels: the queue drains)
            float u0, u1, u2;
            if (HEAD_ABL & (1 | 64)) { u0 = (float)(lane + c); u1 = (float)(lane ^ c); u2 = (float)(lane - c); }
            else {
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
            if (!(HEAD_ABL & (1 | 32)) && c + RAW_DEPTH < C) request(c + RAW_DEPTH, so);   // (its slot's values are in registers: the FMAs above consumed the reads)
            so = so + 1024u == RAW_DEPTH * 1024u ? 0u : so + 1024u;
        }
    }
    if (j >= w || lane == 63 || 2 * j >= W) return;
    if ((HEAD_ABL & 16) && C > 0) return;
    const int X0 = 2 * j;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        float* dst = o < 8 ? gout + ((size_t)b * 8 + o) * HWo : (bout ? bout + (size_t)b * HWo : nullptr);
        if (!dst) continue;
        // SITED: guidance plane o is stored where the propagation reads it: the value of pixel q at p = q - off_o (only inside the image; the
        // positions no value reaches are the normalisation's zeros)
        const int sy = (SITED && o < 8) ? dy2(o) : 0, sx = (SITED && o < 8) ? dx2(o) : 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int Y0 = 2 * (i0 + half);
            if (i0 + half >= h) continue;
            const float p00 = half ? P00[o].y : P00[o].x, p01 = half ? P01[o].y : P01[o].x, p10 = half ? P10[o].y : P10[o].x, p11 = half ? P11[o].y : P11[o].x;
            // a pixel's two outputs of a row are neighbours in memory (also when shifted): one 8-byte store where both exist
            auto put2 = [&](int Y, float va, float vb) {
                if (Y >= H) return;                                  // (the narrowed output)
                const int yp = Y - sy, xa = X0 - sx;
                if (SITED && (yp < 0 || yp >= H)) return;
                float* d = dst + (size_t)yp * W + xa;
                const bool oka = xa >= 0 && xa < W, okb = X0 + 1 < W && xa + 1 >= 0 && xa + 1 < W;
                if (oka && okb) { const f2 v = f2{va, vb}; __builtin_memcpy(d, &v, 8); }
                else if (oka) d[0] = va;
                else if (okb) d[1] = vb;
            };
            put2(Y0, p00, p01);
            put2(Y0 + 1, p10, p11);
        }
    }
}

// affinity_normalization (cspn.py:85-144) in place on consumer-sited planes: wb_k(p) = G_k(p) / sum_j |G_j(p)| (G = |G| first for '8sum_abs'), IEEE division
// (0 / 0 = NaN as torch.div, cspn.py:138); G_k(p) = 0 where p + off_k lies outside the image (nothing was stored there)
__global__ __launch_bounds__(256) void head_norm_sited_kernel(float* __restrict__ g, int B, int H, int W, int norm) {
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / HW);
    const int r = (int)(idx - (size_t)b * HW);
    const int y = r / W, x = r - y * W;
    float* gb = g + (size_t)b * 8 * HW + r;
    float G[8], S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = y + dy2(k), xx = x + dx2(k);
        float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? gb[k * HW] : 0.f;
        if (norm == CSPN_NORM_8SUM_ABS) v = fabsf(v);
        G[k] = v;
        S += fabsf(v);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) gb[k * HW] = G[k] / S;
}

}  // namespace

// the raw kernel's weight records ([c][84] floats)
size_t head_workspace(int C) { return ((size_t)C * WREC * sizeof(float) + 255) & ~(size_t)255; }

// mode 0: raw guidance; 1 / 2: gate_wb of '8sum' / '8sum_abs'
int head_forward(const float* x, const float* w6, const float* w5, float* gout, float* bout, int B, int C, int h, int w, int H, int W, int mode,
                 void* ws, hipStream_t st) {
    float* wr = (float*)ws;
    hipLaunchKernelGGL(head_rawpack_kernel, dim3((C * WREC + 255) / 256), dim3(256), 0, st, w6, w5, wr, C);
    const long long units = (long long)B * ((h + 1) / 2) * ((w + 62) / 63);
    const long long groups = ((units + 3) / 4 + 7) / 8 * 8;   // (a multiple of 8: see the kernel's XCD mapping; the spare waves return at once)
    if (mode == 0) {
        hipLaunchKernelGGL(head_raw_kernel<false>, dim3((unsigned)groups), dim3(256), 0, st, x, wr, gout, bout, C, h, w, H, W, B);
        return check_launch("head_raw_kernel");
    }
    hipLaunchKernelGGL(head_raw_kernel<true>, dim3((unsigned)groups), dim3(256), 0, st, x, wr, gout, bout, C, h, w, H, W, B);
    if (int e = check_launch("head_raw_kernel (sited)")) return e;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(head_norm_sited_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gout, B, H, W,
                       mode == 2 ? CSPN_NORM_8SUM_ABS : CSPN_NORM_8SUM);
    return check_launch("head_norm_sited_kernel");
}

}  // namespace cspn
