// cspn_head_backward.hip -- the gradient of the guidance heads of cspn_head.hip (reference cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206, Unpool :41-54;
// what autograd computes when the training loop back-propagates through :372-373): raw mode only (the heads' own outputs; the chain through a normalising
// epilogue is the PRENORM backward's, cspn2d_backward.hip).
//     out[o][Y][X] = sum_{c,ky,kx} W[o][c][ky][kx] U[c][Y + ky - 1][X + kx - 1],   U[c][2i][2j] = x[c][i][j], zeros elsewhere / beyond the narrowed H x W
//     dL/dx[c][i][j]      = sum_{o,ky,kx} W[o][c][ky][kx] g[o][2i + 1 - ky][2j + 1 - kx]                        (g = dL/dout, zero outside the output)
//     dL/dW[o][c][ky][kx] = sum_{b,i,j}   x[b][c][i][j]   g[b][o][2i + 1 - ky][2j + 1 - kx]
// Both are the forward's 81 products per (input pixel, channel) again: 61 GFLOP each at [64,64,152,608].
//   * head_bwd_x_kernel: one lane = one input pixel; its 9 x 3 x 3 window of g (81 values) is loaded ONCE into registers (27 8-byte loads + the left column from the
//     neighbouring lane by DPP), then every channel takes 41 packed FMAs over PAIRS of taps with the weights as scalar operands (the pair-of-channels form
//     broadcasts g, and the compiler hoists the 81 broadcast pairs out of the channel loop: 168 registers).
//   * head_bwd_w2_kernel: a GEMM -- [81 taps] x [pixels] x [C channels], K = every input pixel of the batch -- on the matrix cores: v_mfma_f32_32x32x2_f32, taps
//     padded to 96 (3 row blocks), channels in blocks of 32; a wave keeps all 3 x 2 accumulator blocks (96 registers) over its share of the pixels and writes them
//     once; head_bwd_w_reduce_kernel sums the waves' partial sums in a fixed order (deterministic: no atomics).  Tiles of 32 pixels are read coalesced and go
//     through the wave's own LDS region into the matrix layout (the first version, head_bwd_w_kernel, gathers them: -DDW_V1).
#include <cstdint>

#include "cspn_common.h"

namespace cspn {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_shr1(float t) {   // the value of the lane before (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}

// ---- dL/dx ----------------------------------------------------------------------------------------------------------------------------
// weights of a channel in the order the kernel walks the window: wq[c][t = o * 9 + r * 3 + k] = W[o][c][2 - r][2 - k]  (r = window row: Y = 2i - 1 + r,
// k = window column: X = 2j - 1 + k); o = 8: the blur head (zeros if absent); t = 81: 0 (the window is held as 41 register PAIRS of neighbouring taps)
constexpr int XREC = 82;
__global__ __launch_bounds__(256) void head_bwdx_pack_kernel(const float* __restrict__ w6, const float* __restrict__ w5, float* __restrict__ wq, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * XREC) return;
    const int c = idx / XREC, t = idx - c * XREC;
    const int o = t / 9, r = (t - o * 9) / 3, k = t - o * 9 - r * 3;
    float v = 0.f;
    if (t < 81) {
        const int tap = (2 - r) * 3 + (2 - k);
        v = o < 8 ? w6[((size_t)o * C + c) * 9 + tap] : (w5 ? w5[(size_t)c * 9 + tap] : 0.f);
    }
    wq[idx] = v;
}

__global__ __launch_bounds__(256) void head_bwd_x_kernel(const float* __restrict__ gg, const float* __restrict__ gb, const float* __restrict__ wq,
                                                          float* __restrict__ dx, int C, int h, int w, int H, int W, int B) {
    const int wq_ = (w + 62) / 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per_xcd = gridDim.x >> 3;   // (a contiguous eighth of the units per XCD: neighbouring rows share a row of g)
    const int unit = (((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3)) * 4 + wv;
    const int seg = unit % wq_;
    const int i = (unit / wq_) % h, b = unit / (wq_ * h);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int j = seg * 63 + lane - 1;                    // lane 0: the column before the segment (it only serves lane 1)
    const size_t HWo = (size_t)H * W, hw = (size_t)h * w;
    // the window G[t = o * 9 + r * 3 + k] = g[o][2i - 1 + r][2j - 1 + k] (zero outside the output) as pairs (G[2p], G[2p + 1]): a packed FMA takes two taps
    f2 GP[41];
    GP[40] = f2{0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const float* src = o < 8 ? gg + ((size_t)b * 8 + o) * HWo : (gb ? gb + (size_t)b * HWo : nullptr);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int Y = 2 * i - 1 + r, X = 2 * j;
            float a = 0.f, c = 0.f;
            if (src && Y >= 0 && Y < H && j >= 0) {
                const float* p = src + (size_t)Y * W + X;
                if (X + 1 < W) { f2 v; __builtin_memcpy(&v, p, 8); a = v.x; c = v.y; }
                else if (X < W) a = p[0];
            }
            const float l = wave_shr1(c);                 // column 2j - 1 = the lane before's 2(j - 1) + 1
            const int t = o * 9 + r * 3;
            GP[t >> 1][t & 1] = l;
            GP[(t + 1) >> 1][(t + 1) & 1] = a;
            GP[(t + 2) >> 1][(t + 2) & 1] = c;
        }
    }
    const bool own = lane >= 1 && j < w;
    const bool fed = own && 2 * i < H && 2 * j < W;       // (an input whose unpooled position lies beyond the narrowed output fed nothing: gradient 0)
    float* dst = dx + (size_t)b * C * hw + (size_t)i * w + (own ? j : 0);
    for (int c = 0; c < C; ++c) {
        const f2* wc = reinterpret_cast<const f2*>(wq + (size_t)c * XREC);   // scalar loads: 82 dwords per channel
        f2 acc = wc[0] * GP[0], acd = wc[1] * GP[1];      // (two chains)
#pragma unroll
        for (int p = 2; p < 41; p += 2) {
            acc = __builtin_elementwise_fma(wc[p], GP[p], acc);
            if (p + 1 < 41) acd = __builtin_elementwise_fma(wc[p + 1], GP[p + 1], acd);
        }
        acc += acd;
        if (own) dst[(size_t)c * hw] = fed ? acc.x + acc.y : 0.f;
    }
}

// ---- dL/dW ----------------------------------------------------------------------------------------------------------------------------
// D[tap][channel] += sum over pixels A[tap][pixel] B[pixel][channel], A = the window value g[o][2i - 1 + r][2j - 1 + k] (tap t = o * 9 + r * 3 + k, padded to 96),
// B = x[channel][i][j].  v_mfma_f32_32x32x2_f32: A 32 x 2 (lane l: row l % 32, k = l / 32), B 2 x 32 (lane l: k = l / 32, column l % 32), D 32 x 32 in 16
// registers (lane l: column l % 32; register q: row (q / 4) * 8 + (l / 32) * 4 + q % 4).  A tile = 8 consecutive pixels of an input row: lanes 0-31 take
// pixels j0 .. j0 + 3, lanes 32-63 pixels j0 + 4 .. j0 + 7, four MFMA steps per (tap block, channel block) -- a lane reads its tap's four window values (stride 2
// between pixels) and its channel's four pixels; the next tile's reads are issued before the current tile's MFMAs.  Timing builds (-DDW_ABL, KITTI x 64): matrix
// instructions alone 0.73-0.90 ms, the reads alone 1.23 ms, together 1.24: the kernel is bound by its gathers -- a lane reads 16-32 bytes of a line of ITS tap / channel
// plane, 2.4 GB arrive at 1.9 TB/s.  Staging coalesced rows through LDS would leave the matrix cores as the limit (~0.75 ms); not built.
#ifndef DW_ABL
#define DW_ABL 0        // timing builds only: 1 = one tile's values for all tiles (no loads in the loop), 2 = no matrix instructions
#endif
#ifndef DW_PX
#define DW_PX 4          // pixels per lane and tile (8: the same load time, two waves per SIMD instead of three, worse overlap: 1.58 vs 1.24 ms)
#endif
constexpr int DW_TILE = 2 * DW_PX;
struct DwTile { float a[3][DW_PX]; float bq[2][DW_PX]; };

template <int NB>
__global__ __launch_bounds__(256, 2) void head_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ gg, const float* __restrict__ gb,
                                                             float* __restrict__ part, const float* __restrict__ part_zero, int C, int c0, int h, int w, int H, int W,
                                                             int tiles, int tiles_w, int hfed, int nwave) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (wave >= nwave) return;
    const int t0 = (int)((long long)tiles * wave / nwave), t1 = (int)((long long)tiles * (wave + 1) / nwave);   // this wave's share of the tiles
    const int half = lane >> 5, id = lane & 31;
    const size_t HWo = (size_t)H * W, hw = (size_t)h * w;
    // what does not change from tile to tile: the lane's taps (one per row block) and channels (one per column block)
    int tr[3], tk[3];
    const float* tsrc[3];          // plane of image 0 (a padding row of the block / no blur head: some valid plane, never used)
    bool tvalid[3];
    size_t tstep[3];               // from one image to the next
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
        const int t = tb * 32 + id;
        const int o = t / 9;
        tr[tb] = (t - o * 9) / 3;
        tk[tb] = t - o * 9 - tr[tb] * 3;
        tvalid[tb] = t < 81 && (o < 8 || gb != nullptr);
        tsrc[tb] = (tvalid[tb] && o == 8) ? gb : gg + (size_t)(o < 8 ? o : 0) * HWo;
        tstep[tb] = (tvalid[tb] && o == 8) ? HWo : 8 * HWo;
    }
    const float* xsrc[NB];
    bool xvalid[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { const int ch = c0 + nb * 32 + id; xvalid[nb] = ch < C; xsrc[nb] = x + (size_t)(ch < C ? ch : 0) * hw; }
    const float* zero = part_zero;
    auto load = [&](DwTile& T, int tw, int i, int b) {       // tile tw of input row i of image b
        const int jb = tw * DW_TILE + DW_PX * half;             // this lane's first pixel
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
            const int Y = 2 * i - 1 + tr[tb], Xb = 2 * jb - 1 + tk[tb];
            const bool rowok = tvalid[tb] && Y >= 0 && Y < H;
            const float* p = tsrc[tb] + (size_t)b * tstep[tb] + (size_t)Y * W;
            if (!rowok || (Xb >= 0 && Xb + 2 * DW_PX - 1 < W)) {   // the common case: 2 DW_PX consecutive floats, every other one is a pixel's (a row outside: zeros)
                float q[2 * DW_PX];
                __builtin_memcpy(q, rowok ? p + Xb : zero, 8 * DW_PX);
#pragma unroll
                for (int s_ = 0; s_ < DW_PX; ++s_) T.a[tb][s_] = q[2 * s_];
            } else {                                             // the first / last tiles of a row: what lies outside reads a zero word
#pragma unroll
                for (int s_ = 0; s_ < DW_PX; ++s_) {
                    const int X = Xb + 2 * s_;
                    T.a[tb][s_] = *((X >= 0 && X < W) ? p + X : zero);
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float* p = xsrc[nb] + (size_t)b * C * hw + (size_t)i * w;
            if (!xvalid[nb] || (jb + DW_PX - 1 < w && 2 * (jb + DW_PX - 1) < W)) {
                __builtin_memcpy(T.bq[nb], xvalid[nb] ? p + jb : zero, 4 * DW_PX);
            } else {
#pragma unroll
                for (int s_ = 0; s_ < DW_PX; ++s_) {
                    const int j = jb + s_;
                    T.bq[nb][s_] = *((j < w && 2 * j < W) ? p + j : zero);      // (beyond the narrowed output: fed nothing)
                }
            }
        }
    };
    f16v acc[3][NB];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[tb][nb][q] = 0.f;
    DwTile nxt;
    int tw = (int)((unsigned)t0 % (unsigned)tiles_w), ti = (int)(((unsigned)t0 / (unsigned)tiles_w) % (unsigned)hfed), tb_ = (int)((unsigned)t0 / ((unsigned)tiles_w * (unsigned)hfed));
    if (t0 < t1) load(nxt, tw, ti, tb_);
    for (int t = t0; t < t1; ++t) {
        const DwTile cur = nxt;
        if (++tw == tiles_w) { tw = 0; if (++ti == hfed) { ti = 0; ++tb_; } }      // the next tile (scalar)
        if (!(DW_ABL & 1) && t + 1 < t1) load(nxt, tw, ti, tb_);
#pragma unroll
        for (int s_ = 0; s_ < DW_PX; ++s_)
#pragma unroll
            for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (DW_ABL & 2) acc[tb][nb][0] += cur.a[tb][s_] * cur.bq[nb][s_];
                    else acc[tb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tb][s_], cur.bq[nb][s_], acc[tb][nb], 0, 0, 0);
                }
    }
    float* dst = part + (size_t)wave * 3 * NB * 16 * 64 + lane;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) dst[((tb * NB + nb) * 16 + q) * 64] = acc[tb][nb][q];
}

// ---- dL/dW, second version: the same GEMM fed through LDS.  A wave reads a tile of 32 pixels COALESCED -- lane = pixel: per channel pair one 256-byte load of x,
// per (plane, window row) one 256-byte load of g (66 columns: + one load for the two left over of all 27 rows) --, drops the values into its OWN LDS region (no
// barrier: nobody else touches it) and reads them back in the matrix cores' layout (lane = tap / channel; odd pitches: conflict-free).  60 loads, 60 LDS writes,
// 80 LDS reads and 96 MFMAs per tile against 8 gathers per 24 MFMAs above (kept behind -DDW_V1): 0.94 ms against 1.24 at KITTI x 64.  Timing builds: the matrix
// instructions with their LDS reads alone 0.66 ms, the loads + LDS writes alone 0.65, together 0.98 with or without the next tile's loads in flight during the MFMAs
// (two waves per SIMD at 254 registers).
constexpr int W2_TP = 32, W2_XP = 33, W2_GP = 67;
template <int NB>
__global__ __launch_bounds__(256, 2) void head_bwd_w2_kernel(const float* __restrict__ x, const float* __restrict__ gg, const float* __restrict__ gb,
                                                              float* __restrict__ part, const float* __restrict__ zero, int C, int c0, int h, int w, int H, int W,
                                                              int tiles, int tiles_w, int hfed, int nwave) {
    constexpr int XT = NB * 32 * W2_XP, GT = 28 * W2_GP;      // x tile [channel][33]; g tile [27 window rows + a row of zeros][67]
    __shared__ float lds[4][XT + GT];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (wave >= nwave) return;
    float* xs = &lds[wv][0];
    float* gs = &lds[wv][XT];
    const int t0 = (int)((long long)tiles * wave / nwave), t1 = (int)((long long)tiles * (wave + 1) / nwave);
    const int half = lane >> 5, id = lane & 31;
    const size_t HWo = (size_t)H * W, hw = (size_t)h * w;
    for (int q = lane; q < W2_GP; q += 64) gs[27 * W2_GP + q] = 0.f;          // the row the padding taps read
    // where this lane's taps / channels sit in the tile (MFMA layout)
    int aoff[3], boff[NB];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
        const int t = tb * 32 + id;
        const int o = t / 9, r = (t - o * 9) / 3, k = t - o * 9 - r * 3;
        aoff[tb] = (t < 81 ? (o * 3 + r) * W2_GP + k : 27 * W2_GP) + 2 * 16 * half;     // + 2 px per step
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) boff[nb] = (nb * 32 + id) * W2_XP + 16 * half;
    f16v acc[3][NB];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[tb][nb][q] = 0.f;
    int tw = (int)((unsigned)t0 % (unsigned)tiles_w), ti = (int)(((unsigned)t0 / (unsigned)tiles_w) % (unsigned)hfed), tb_ = (int)((unsigned)t0 / ((unsigned)tiles_w * (unsigned)hfed));
    // x: a channel pair per load (lanes 0-31: channel 2p, lanes 32-63: channel 2p + 1; lane & 31 = pixel).  The NEXT tile's x is requested before this tile's matrix
    // instructions and waits in registers
    // Every load is (scalar row / plane address) + (one 32-bit lane offset, clamped into the row): no per-load vector address arithmetic; what lies outside is
    // zeroed when the value goes to LDS a tile later (the masks are recomputed there).
    float xv[NB * 16];
    auto ld = [&](const float* sbase, unsigned voff) { return *(const float*)((const char*)sbase + voff); };
    auto load_x = [&](int tw_, int i_, int b_) {
        const int j = tw_ * W2_TP + id;
        const unsigned voff0 = 4u * (unsigned)(j < w ? j : w - 1);
        const unsigned voff = voff0 + 4u * (unsigned)half * (unsigned)hw;      // lanes 32-63: the pair's second channel
        const float* row = x + ((size_t)b_ * C + c0) * hw + (size_t)i_ * w;
        const int nch = C - c0;                                              // channels left in this column block
#pragma unroll
        for (int p = 0; p < NB * 16; ++p) {
            // (uniform: a pair's first channel clamped into the tensor; a pair without its second channel reads the first for both halves: masked later)
            const int cp = 2 * p < nch ? 2 * p : nch - 1;
            xv[p] = ld(row + (size_t)cp * hw, 2 * p + 1 < nch ? voff : voff0);
        }
    };
    float gv[28];
    auto load_g = [&](int tw_, int i_, int b_) {
        const int j0_ = tw_ * W2_TP;
        const int X = 2 * j0_ - 1 + lane;
        const unsigned voff = 4u * (unsigned)(X < 0 ? 0 : (X >= W ? W - 1 : X));
        const int rl = lane >> 1, X2 = 2 * j0_ + 63 + (lane & 1);
        const int ol = rl < 27 ? rl / 3 : 0, Yl = 2 * i_ - 1 + (rl < 27 ? rl - ol * 3 : 0);
        const bool hasb = gb != nullptr;
        const float* gimg = gg + (size_t)b_ * 8 * HWo;
        const float* bimg = hasb ? gb + (size_t)b_ * HWo : gimg;
#pragma unroll
        for (int rr = 0; rr < 27; ++rr) {
            const int o = rr / 3, r = rr - o * 3, Y = 2 * i_ - 1 + r;
            const float* src = o < 8 ? gimg + (size_t)o * HWo : bimg;
            gv[rr] = ld(src + (size_t)(Y < 0 ? 0 : (Y >= H ? H - 1 : Y)) * W, voff);
        }
        const float* src = (ol < 8 || !hasb) ? gimg + (size_t)(ol < 8 ? ol : 0) * HWo : bimg;                  // (per lane: the one gather of the tile)
        gv[27] = src[(size_t)(Yl < 0 ? 0 : (Yl >= H ? H - 1 : Yl)) * W + (X2 < W ? X2 : W - 1)];
    };
    if (t0 < t1) { load_x(tw, ti, tb_); load_g(tw, ti, tb_); }
    for (int t = t0; t < t1; ++t) {
        if (!(DW_ABL & 1) || t == t0) {
        {   // the tile whose values arrived: tw, ti, tb_ still name it
            const int j = tw * W2_TP + id;
            const bool pok = j < w && 2 * j < W;                  // (beyond the narrowed output: fed nothing)
#pragma unroll
            for (int p = 0; p < NB * 16; ++p) xs[(2 * p + half) * W2_XP + id] = (pok && c0 + 2 * p + half < C) ? xv[p] : 0.f;
            const int X = 2 * tw * W2_TP - 1 + lane;
            const bool cok = X >= 0 && X < W, hasb = gb != nullptr;
#pragma unroll
            for (int rr = 0; rr < 27; ++rr) {
                const int o = rr / 3, r = rr - o * 3, Y = 2 * ti - 1 + r;
                gs[rr * W2_GP + lane] = (cok && Y >= 0 && Y < H && (o < 8 || hasb)) ? gv[rr] : 0.f;
            }
            const int rl = lane >> 1, X2 = 2 * tw * W2_TP + 63 + (lane & 1);
            const int ol = rl / 3, Yl = 2 * ti - 1 + (rl - ol * 3);
            if (lane < 54) gs[rl * W2_GP + 64 + (lane & 1)] = (X2 < W && Yl >= 0 && Yl < H && (ol < 8 || hasb)) ? gv[27] : 0.f;
        }
        }   // (DW_ABL & 1: the first tile's values for every tile)
        if (++tw == tiles_w) { tw = 0; if (++ti == hfed) { ti = 0; ++tb_; } }
        if (!(DW_ABL & 1) && t + 1 < t1) { load_x(tw, ti, tb_); load_g(tw, ti, tb_); }
        // ---- 16 steps of two pixels (s, s + 16)
#pragma unroll 4
        for (int s_ = 0; s_ < 16; ++s_) {
            float av[3], bv[NB];
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) av[tb] = gs[aoff[tb] + 2 * s_];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = xs[boff[nb] + s_];
#pragma unroll
            for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (DW_ABL & 2) acc[tb][nb][0] += av[tb] * bv[nb];
                    else acc[tb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tb], bv[nb], acc[tb][nb], 0, 0, 0);
                }
        }
    }
    float* dst = part + (size_t)wave * 3 * NB * 16 * 64 + lane;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) dst[((tb * NB + nb) * 16 + q) * 64] = acc[tb][nb][q];
}

__global__ void head_zero_line_kernel(float* __restrict__ z) { z[threadIdx.x] = 0.f; }

// dW[o][c][ky][kx] = sum over the waves' partial blocks, in wave order (deterministic)
__global__ __launch_bounds__(256) void head_bwd_w_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw6, float* __restrict__ dw5, int C, int c0,
                                                                 int NB, int nwave) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nch = NB * 32;
    if (idx >= 81 * nch) return;
    const int t = idx / nch, cl = idx - t * nch, ch = c0 + cl;
    if (ch >= C) return;
    const int tb = t >> 5, i = t & 31, nb = cl >> 5, jc = cl & 31;
    const int q = (i >> 3) * 4 + (i & 3), l = ((i & 7) >> 2) * 32 + jc;
    const float* p = part + ((size_t)(tb * NB + nb) * 16 + q) * 64 + l;
    const size_t stride = (size_t)3 * NB * 16 * 64;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int wv = 0;
    for (; wv + 3 < nwave; wv += 4) { s0 += p[(size_t)wv * stride]; s1 += p[(size_t)(wv + 1) * stride]; s2 += p[(size_t)(wv + 2) * stride]; s3 += p[(size_t)(wv + 3) * stride]; }
    for (; wv < nwave; ++wv) s0 += p[(size_t)wv * stride];
    const float v = (s0 + s1) + (s2 + s3);
    const int o = t / 9, r = (t - o * 9) / 3, k = t - o * 9 - r * 3;
    const int tap = (2 - r) * 3 + (2 - k);
    if (o < 8) { if (dw6) dw6[((size_t)o * C + ch) * 9 + tap] = v; }
    else if (dw5) dw5[(size_t)ch * 9 + tap] = v;
}

}  // namespace

constexpr int DW_MAX_WAVES = 2048;
static size_t bwdx_bytes(int C) { return (((size_t)C * XREC * sizeof(float)) + 255) & ~(size_t)255; }
size_t head_backward_workspace(int B, int C, int h, int w) {
    (void)B; (void)C; (void)h; (void)w;
    return bwdx_bytes(C) + 256 + (size_t)DW_MAX_WAVES * 3 * 2 * 16 * 64 * sizeof(float);   // + a line of zeros + the waves' partial blocks of dL/dW (64 channels at a time)
}

int head_backward(const float* x, const float* w6, const float* w5, const float* gg, const float* gb, float* dx, float* dw6, float* dw5, int B, int C, int h,
                  int w, int H, int W, void* ws, hipStream_t st) {
    if (dx) {
        float* wq = (float*)ws;
        hipLaunchKernelGGL(head_bwdx_pack_kernel, dim3((C * XREC + 255) / 256), dim3(256), 0, st, w6, w5, wq, C);
        const long long units = (long long)B * h * ((w + 62) / 63);
        const long long groups = ((units + 3) / 4 + 7) / 8 * 8;
        hipLaunchKernelGGL(head_bwd_x_kernel, dim3((unsigned)groups), dim3(256), 0, st, gg, gb, wq, dx, C, h, w, H, W, B);
        if (int e = check_launch("head_bwd_x_kernel")) return e;
    }
    if (dw6 || dw5) {
        float* zero = (float*)((char*)ws + bwdx_bytes(C));        // what a tile reads for positions outside the tensors
        float* part = zero + 64;
        hipLaunchKernelGGL(head_zero_line_kernel, dim3(1), dim3(64), 0, st, zero);
        const int hfed = (H + 1) / 2 < h ? (H + 1) / 2 : h;     // input rows whose unpooled row lies inside the (narrowed) output
        const int wfed = (W + 1) / 2 < w ? (W + 1) / 2 : w;
#ifdef DW_V1
        const int tiles_w = (wfed + DW_TILE - 1) / DW_TILE;
#else
        const int tiles_w = (wfed + W2_TP - 1) / W2_TP;
#endif
        const long long tiles_ll = (long long)B * hfed * tiles_w;
        if (tiles_ll >= (1ll << 31)) { set_error("cspn_guidance_head_backward_f32: too many pixels"); return CSPN_E_UNSUPPORTED; }
        const int tiles = (int)tiles_ll;
        const int nwave = tiles < DW_MAX_WAVES ? tiles : DW_MAX_WAVES;
        for (int c0 = 0; c0 < C; c0 += 64) {                     // 64 channels at a time (two column blocks of the matrix core)
            const int NB = C - c0 > 32 ? 2 : 1;
#ifndef DW_V1
            if (NB == 2) hipLaunchKernelGGL(head_bwd_w2_kernel<2>, dim3((nwave + 3) / 4), dim3(256), 0, st, x, gg, gb, part, zero, C, c0, h, w, H, W, tiles, tiles_w, hfed, nwave);
            else hipLaunchKernelGGL(head_bwd_w2_kernel<1>, dim3((nwave + 3) / 4), dim3(256), 0, st, x, gg, gb, part, zero, C, c0, h, w, H, W, tiles, tiles_w, hfed, nwave);
#else
            if (NB == 2) hipLaunchKernelGGL(head_bwd_w_kernel<2>, dim3((nwave + 3) / 4), dim3(256), 0, st, x, gg, gb, part, zero, C, c0, h, w, H, W, tiles, tiles_w, hfed, nwave);
            else hipLaunchKernelGGL(head_bwd_w_kernel<1>, dim3((nwave + 3) / 4), dim3(256), 0, st, x, gg, gb, part, zero, C, c0, h, w, H, W, tiles, tiles_w, hfed, nwave);
#endif
            hipLaunchKernelGGL(head_bwd_w_reduce_kernel, dim3((81 * NB * 32 + 255) / 256), dim3(256), 0, st, part, dw6, dw5, C, c0, NB, nwave);
        }
        if (int e = check_launch("head_bwd_w_kernel")) return e;
    }
    return 0;
}

}  // namespace cspn
