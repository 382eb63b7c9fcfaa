// cspn2d_fused.hip -- all n_iter propagation steps of Affinity_Propagate.forward
// (reference cspn_pytorch/models/cspn.py:42-83, with affinity_normalization :85-144,
// pad_blur_depth :147-172, sum_conv :44-53 and the tail :70-81) in ONE launch for gfx950.
//
// "Time-skewed wave ring" (executable specification: tools/tsw_model.py):
//   * a workgroup = 8 waves streams a 256-column band of the image top -> bottom;
//   * every lane owns 4 adjacent columns, held as two float2 pairs (c0,c2),(c1,c3) so that the
//     inner loop is v_pk_fma_f32 (2 FMAs per VALU slot; x+-1 inside a lane is the OTHER pair,
//     across lanes it is a DPP wave_shr/wave_shl move);
//   * every wave keeps 4 consecutive image rows resident: 9 folded coefficients per pixel
//     (8 normalised, mask-folded affinities + the centre/mask constant) and two partial
//     accumulators -- 176 VGPRs of state, weights never leave registers;
//   * the 32 resident rows form a ring; row q enters at step phi(q) = 3*(q/4) + q%4 and
//     advances one CSPN iteration per step, so rows inside a wave sit on a 1-level staircase
//     and the first/last row of neighbouring waves sit at the SAME level ("flat spot"): the
//     only cross-wave traffic is two 1 KB boundary rows per wave per step through LDS, consumed
//     one step later (one s_barrier per step, double buffered);
//   * "push" form: a completed row value V immediately adds its three contributions
//     (below / self / above taps) to the accumulators of rows r-1, r, r+1, so shifted copies of
//     V are transient and no second copy of the depth is stored;
//   * affinity normalisation, centre term and sparse-mask folding ("cooking") are done by all
//     512 threads, one pixel each, two events every three steps, from global loads issued one
//     event earlier; cooked rows reach the owning wave through LDS (9 planes + H0 ring);
//   * all 24 iterations need each input byte once: 40 B/pixel (44 with sparse) of HBM traffic.
#include <type_traits>

#include "cspn_common.h"

namespace cspn {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int NW = 8;        // waves per workgroup
constexpr int R = 4;         // resident rows (slots) per wave
constexpr int LV = 24;       // NW*(R-1): iterations fused per pass
constexpr int BW = 256;      // band width in pixels: 64 lanes x 4 columns
constexpr int NT = NW * 64;  // threads per workgroup
constexpr int MIN_ROWS_PER_WG = 16;  // a workgroup pays ~2*n_iter halo rows + a 24-step drain; more workgroups still win

struct Lds {
    float bnd[2][NW][2][BW];  // [step parity][wave][0: top row (slot 0) | 1: bottom row (slot 3)]
    float cook[8][9][BW];     // folded coefficients of stream row q in cook[q & 7], (c0,c2,c1,c3) per lane
    float h0[8][BW];          // level-0 value of stream row q in h0[q & 7], same layout
    int hdr[8][4];            // per cooked row: active, out offset (or -1), own lo, own hi (band relative)
    int meta[NW][R][4];       // the same record for the row currently held by (wave, slot)
    int rinfo[16][8];         // descriptor of stream row q in rinfo[q & 15] (written by wave 0 only)
};

// ---- lane-crossing moves (DPP, whole-wave shift by one lane; edge lanes read 0) -------------
__device__ __forceinline__ float dpp_shr1(float v) {  // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {  // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// A row value as seen by its consumers: p0 = (c0,c2), p1 = (c1,c3) and the two lane-crossing
// pairs xl = (c3 of lane-1, c1), xr = (c2, c0 of lane+1).
struct Shift { f2 p0, p1, xl, xr; };
__device__ __forceinline__ Shift mk_shift(f2 p0, f2 p1) {
    Shift s;
    s.p0 = p0;
    s.p1 = p1;
    s.xl = f2{dpp_shr1(p1.y), p1.x};
    s.xr = f2{p0.y, dpp_shl1(p0.x)};
    return s;
}
// three taps KR (dx=+1), KM (dx=0), KL (dx=-1) of one neighbour row
template <int KR, int KM, int KL>
__device__ __forceinline__ void push3(const f2 (&w)[9][2], const Shift& s, f2& a0, f2& a1) {
    a0 = pkfma(w[KR][0], s.p1, a0);
    a1 = pkfma(w[KR][1], s.xr, a1);
    a0 = pkfma(w[KM][0], s.p0, a0);
    a1 = pkfma(w[KM][1], s.p1, a1);
    a0 = pkfma(w[KL][0], s.xl, a0);
    a1 = pkfma(w[KL][1], s.p0, a1);
}
// channel k <-> (dy,dx): 0 (+1,+1) 1 (+1,0) 2 (+1,-1) | 3 (0,+1) 4 (0,-1) | 5 (-1,+1) 6 (-1,0) 7 (-1,-1)
__device__ __forceinline__ void push_below(const f2 (&w)[9][2], const Shift& s, f2& a0, f2& a1) { push3<0, 1, 2>(w, s, a0, a1); }
__device__ __forceinline__ void push_above(const f2 (&w)[9][2], const Shift& s, f2& a0, f2& a1) { push3<5, 6, 7>(w, s, a0, a1); }
__device__ __forceinline__ void push_self(const f2 (&w)[9][2], const Shift& s, f2& a0, f2& a1) {
    a0 = pkfma(w[3][0], s.p1, a0);
    a1 = pkfma(w[3][1], s.xr, a1);
    a0 = pkfma(w[4][0], s.xl, a0);
    a1 = pkfma(w[4][1], s.p0, a1);
}


__device__ __forceinline__ void wg_barrier() {
#if defined(CSPN_DBG_SYNCTHREADS)
    __syncthreads();
#elif defined(CSPN_DBG_FENCE_BARRIER)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    // LDS-only barrier: must not drain the in-flight global prefetch (vmcnt) like __syncthreads() would
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

#ifdef CSPN_DBG_TIMING
__device__ long long g_tim[8 * 8];  // [wave][phase] cycle totals of block 0
#define TIM_DECL long long tim_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tim_t0 = clock64();
#define TIM(i) do { const long long t_ = clock64(); tim_[i] += t_ - tim_t0; tim_t0 = t_; } while (0)
#define TIM_FLUSH() do { if (blockIdx.x == 0 && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) g_tim[wv * 8 + i_] = tim_[i_]; } } while (0)
#else
#define TIM_DECL
#define TIM(i) do { } while (0)
#define TIM_FLUSH() do { } while (0)
#endif

// ---- the stream of rows a workgroup processes ------------------------------------------------
struct Geo {
    int B, H, W, n_iter, nb, halo;  // nb bands per image, horizontal halo (multiple of 4)
};
struct RowInfo {
    int active;   // 0: separator / past-the-end row (stays exactly zero)
    int y, p0;    // image row, first physical column of the band
    int img;      // b*H*W: element offset of the image inside a 1-channel tensor
    int outoff;   // element offset of out[b][0][y][p0], or -1 when the row is only halo
    int lo, hi;   // owned columns, band relative
};
__device__ __forceinline__ void band_of(const Geo& g, int bi, int& p0, int& lo, int& hi) {
    if (g.W <= BW) { p0 = 0; lo = 0; hi = g.W; return; }
    if (bi == 0) { p0 = 0; lo = 0; }
    else { lo = (BW - g.halo) + (bi - 1) * (BW - 2 * g.halo); p0 = lo - g.halo; }
    hi = (p0 + BW >= g.W) ? g.W : p0 + BW - g.halo;
}
struct Cursor {
    int r_next, r_end;  // [r_next, r_end): still unopened part of this workgroup's share of B*nb*H rows
    int in_seg, pending;
    int img, p0, lo, hi, y0, y1, ye, y;
    __device__ __forceinline__ void init(int r0, int r1) {
        r_next = r0; r_end = r1; in_seg = 0; pending = r0 < r1;
        img = p0 = lo = hi = y0 = y1 = ye = y = 0;
    }
    __device__ __forceinline__ void open(const Geo& g) {
        const int u = r_next / g.H;
        y0 = r_next - u * g.H;
        y1 = min(g.H, y0 + (r_end - r_next));
        r_next += y1 - y0;
        const int b = u / g.nb;
        int plo, phi;
        band_of(g, u - b * g.nb, p0, plo, phi);
        lo = plo - p0; hi = phi - p0;
        img = b * g.H * g.W;
        y = max(0, y0 - g.n_iter);
        ye = min(g.H, y1 + g.n_iter);
        in_seg = 1; pending = 0;
    }
    __device__ __forceinline__ RowInfo next(const Geo& g) {
        RowInfo r;
        if (__builtin_expect(in_seg && y < ye, 1)) {
            r.active = 1; r.y = y; r.p0 = p0; r.img = img; r.lo = lo; r.hi = hi;
            r.outoff = (y >= y0 && y < y1) ? (img + y * g.W + p0) : -1;
            ++y;
            return r;
        }
        r.active = 0; r.y = 0; r.p0 = 0; r.img = 0; r.outoff = -1; r.lo = 0; r.hi = 0;
        if (in_seg) {  // segment exhausted: emit a separator (or idle rows at the very end)
            in_seg = 0; pending = r_next < r_end;
            return r;
        }
        if (!pending) return r;
        open(g);
        r.active = 1; r.y = y; r.p0 = p0; r.img = img; r.lo = lo; r.hi = hi;
        r.outoff = (y >= y0 && y < y1) ? (img + y * g.W + p0) : -1;
        ++y;
        return r;
    }
};
// number of stream rows of a share (segments + separators between them)
__device__ __forceinline__ int stream_length(const Geo& g, int r0, int r1) {
    int q = 0, r = r0;
    while (r < r1) {
        const int u = r / g.H, y0 = r - u * g.H, y1 = min(g.H, y0 + (r1 - r));
        q += (min(g.H, y1 + g.n_iter) - max(0, y0 - g.n_iter)) + (r > r0 ? 1 : 0);
        r += y1 - y0;
    }
    return q;
}

// ---- cooking: per pixel normalise + fold (cspn.py:85-144, :76, :81) ---------------------------
// ---- cooking: a TASK = half a row (128 pixels), 2 adjacent pixels per lane, done by one wave -------------
// (normalise + fold of cspn.py:85-144, :76, :81).  Per pixel this costs a third of the instructions of a
// one-pixel-per-thread formulation, and instructions per SIMD are what bounds this kernel.
struct Pend {  // raw inputs of the lane's two pixels, loaded one step ahead (8-byte loads, 4-byte aligned)
    f2 g[8], blur, hin, sp;
};

__device__ __forceinline__ f2 ld_row2(const float* row, unsigned byte_off) {  // scalar row base + per-lane byte offset
    return *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(row) + byte_off);
}

// Straight-line, branch-free: all loads come from clamped (always valid) addresses so they stay in flight until the
// next step; whatever is outside the image (or belongs to a separator row) is zeroed when the task is cooked.
// A +-1 pixel shift stays inside the tensor: the element before a row start / after a row end belongs to the
// neighbouring row or channel for every plane that is read shifted (channels 2,4,7 shift left, 0,3,5 right).
template <int NORM, bool SPARSE, bool HIN>
__device__ __forceinline__ void issue_loads(Pend& p, const RowInfo& ri, int xb, const Geo& g, const float* __restrict__ gd,
                                            const float* __restrict__ blur, const float* __restrict__ hin,
                                            const float* __restrict__ sparse) {
    const unsigned HW4 = 4u * (unsigned)(g.H * g.W), W4 = 4u * (unsigned)g.W;
    const unsigned oc = 4u * (unsigned)min(ri.p0 + xb, g.W - 2);     // lanes right of the image re-read its last pair
    const int ro = ri.img + ri.y * g.W;                               // element offset of the row in a 1-channel tensor
    const float* grow = gd + ((size_t)ri.img * 8 + (size_t)(ri.y * g.W));  // channel 0, this row
    const unsigned up = (ri.y + 1 < g.H) ? W4 : 0u;                   // row below / above, clamped into the image
    const unsigned dn = (ri.y >= 1) ? W4 : 0u;
    constexpr bool GIVEN = NORM == CSPN_NORM_NONE || NORM == CSPN_NORM_PRENORM;   // coefficients used as given, centre-sited
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (GIVEN) {
            p.g[k] = ld_row2(grow, oc + (unsigned)k * HW4);
        } else {
            const unsigned chan = (unsigned)k * HW4 + (dy2(k) > 0 ? up : 0u) - (dy2(k) < 0 ? dn : 0u);
            p.g[k] = ld_row2(grow, oc + chan + (unsigned)(4 * dx2(k)));
        }
    }
    p.blur = ld_row2(blur + ro, oc);
    p.hin = HIN ? ld_row2(hin + ro, oc) : f2{0.f, 0.f};
    p.sp = SPARSE ? ld_row2(sparse + ro, oc) : f2{0.f, 0.f};
}

__device__ __forceinline__ float fast_rcp(float s) {  // ~1 ulp; rcp(0) = inf so 0 * inf = NaN like torch.div's 0/0
    float r = __builtin_amdgcn_rcpf(s);
    const float e = fmaf(-s, r, 1.f);
    return fmaf(e, r, r);
}

// Wave-uniform specialisations: YINT = the rows above and below are inside the image; XFULL = the whole 256-column band
// is inside the image and the row is a real one (not a separator).
template <int NORM, bool SPARSE, bool HIN, bool YINT, bool XFULL>
__device__ __forceinline__ void cook_task(const Pend& p, const RowInfo& ri, int q, int xb, const Geo& g, Lds& lds) {
    const int x0 = ri.p0 + xb;                             // image column of the lane's first pixel (even)
    const bool pv = XFULL || (ri.active && x0 < g.W);      // both pixels inside the image (W % 4 == 0)
    const bool el = x0 == 0;                               // pixel 0 has no left neighbour
    const bool er = x0 + 2 == g.W;                         // pixel 1 has no right neighbour
    const bool ru = YINT || ri.y + 1 < g.H, rd = YINT || ri.y >= 1;  // row below / above inside the image
    constexpr bool GIVEN = NORM == CSPN_NORM_NONE || NORM == CSPN_NORM_PRENORM;
    f2 gv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f2 t = p.g[k];
        if (!GIVEN) {
            const bool rok = dy2(k) > 0 ? ru : (dy2(k) < 0 ? rd : true);  // wave-uniform
            if (!rok) t = f2{0.f, 0.f};
            if (dx2(k) < 0) t.x = el ? 0.f : t.x;
            if (dx2(k) > 0) t.y = er ? 0.f : t.y;
        }
        if (!XFULL && !pv) t = f2{0.f, 0.f};  // whatever the clamped load fetched: never let it in
        if (NORM == CSPN_NORM_8SUM_ABS) t = f2{fabsf(t.x), fabsf(t.y)};
        gv[k] = t;
    }
    f2 h0 = p.blur, hv = HIN ? p.hin : p.blur;
    f2 scale = f2{1.f, 1.f}, c = f2{0.f, 0.f};
    if (NORM == CSPN_NORM_PRENORM) {   // the planes are the w_k(p) of cspn.py:138 already: only the centre term is left (cspn.py:76)
        const f2 T = ((gv[0] + gv[1]) + (gv[2] + gv[3])) + ((gv[4] + gv[5]) + (gv[6] + gv[7]));
        c = __builtin_elementwise_fma(-T, h0, h0);
    } else if (!GIVEN) {
        f2 S = f2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) S += f2{fabsf(gv[k].x), fabsf(gv[k].y)};
        const f2 T = ((gv[0] + gv[1]) + (gv[2] + gv[3])) + ((gv[4] + gv[5]) + (gv[6] + gv[7]));
        scale = f2{fast_rcp(S.x), fast_rcp(S.y)};
        c = __builtin_elementwise_fma(-(T * scale), h0, h0);  // (1 - sigma) * H0, cspn.py:76
    }
    if (SPARSE) {  // cspn.py:64,81: mask pins to H0; folded into the coefficients
        const f2 m = f2{signf(p.sp.x), signf(p.sp.y)}, om = f2{1.f, 1.f} - m;
        scale *= om;
        c = __builtin_elementwise_fma(om, c, m * h0);
    }
    if (!XFULL && !pv) { scale = c = hv = f2{0.f, 0.f}; }  // separator rows / columns right of the image stay exactly zero
    // LDS layout per owner lane (4 columns): (c0,c2,c1,c3); this lane's pixels are (c0,c1) or (c2,c3) of one group
    const int pos = (xb & ~3) | ((xb >> 1) & 1);
    const int cb = q & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f2 w = (GIVEN && !SPARSE) ? gv[k] : gv[k] * scale;
        lds.cook[cb][k][pos] = w.x;
        lds.cook[cb][k][pos + 2] = w.y;
    }
    lds.cook[cb][8][pos] = c.x;
    lds.cook[cb][8][pos + 2] = c.y;
    lds.h0[cb][pos] = hv.x;
    lds.h0[cb][pos + 2] = hv.y;
}

// ---- the kernel -------------------------------------------------------------------------------
template <int NORM, bool SPARSE, bool HIN>
__global__ __launch_bounds__(NT, 2) void cspn2d_fused_kernel(const float* __restrict__ gd, const float* __restrict__ blur,
                                                              const float* __restrict__ hin,
                                                              const float* __restrict__ sparse, float* __restrict__ out,
                                                              Geo geo) {
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // this workgroup's share of the B x bands x H row space
    const long long total = (long long)geo.B * geo.nb * geo.H;
    const int r0 = (int)(total * blockIdx.x / gridDim.x), r1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int Q = stream_length(geo, r0, r1);
    if (Q == 0) return;
    const int last_step = 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + geo.n_iter;

    // ---- register-resident state ----
    f2 Wt[R][9][2];  // folded coefficients of the 4 resident rows (Wt[j][8] = c')
    f2 S[2][R][2];   // accumulators, S[step parity][slot][pair]
    int act[R] = {0, 0, 0, 0};  // 0: the slot holds a separator / nothing -> its value is pinned to 0
    int allact = 0;             // all four slots hold real rows (then the event-free fast step is legal)
    int cnt0 = (LV * 4 - 3 * wv) % LV;               // slot j is at phase (cnt0 - j) mod 24; phase 0 = injection step
    int rcnt = (cnt0 + 2 * LV - geo.n_iter) % LV;    // slot j retires (level n_iter complete) when (rcnt - j) mod 24 == 0
    int qgen = 0;    // generation: slot j takes stream row 4*(wv + 8*qgen) + j next
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { Wt[j][k][0] = f2{0.f, 0.f}; Wt[j][k][1] = f2{0.f, 0.f}; }
        S[0][j][0] = S[0][j][1] = S[1][j][0] = S[1][j][1] = f2{0.f, 0.f};
    }
    for (int i = tid; i < 2 * NW * 2 * BW; i += NT) (&lds.bnd[0][0][0][0])[i] = 0.f;

    // ---- cooking pipeline.  Rows enter at 4 per 3 steps, so every third step (tau % 3 == 2) all eight waves cook
    // one TASK each: group gamma = (tau + 1) / 3 = stream rows 4*gamma-1 .. 4*gamma+2 (they enter at steps 3*gamma,
    // 3*gamma, 3*gamma+1, 3*gamma+2), wave w takes half (w & 1) of row 4*gamma - 1 + (w >> 1).  Identical work for
    // every wave (a step is as slow as its busiest wave).  The raw inputs of a task are loaded by its wave one event
    // (3 steps) earlier; wave 0 alone walks the stream and publishes row descriptors (lds.rinfo) one event before that
    // (one scalar unit per CU: scalar work is scarce).
    Cursor cur;
    cur.init(r0, r1);
    int qpub = 0;  // next stream row to publish (wave 0)
    auto publish_upto = [&](int qlast) {
        if (wv != 0) return;
        while (qpub <= qlast) {
            const RowInfo r = cur.next(geo);
            if (lane == 0) {
                int* d = &lds.rinfo[qpub & 15][0];
                *reinterpret_cast<int4*>(d) = make_int4(r.active, r.y, r.p0, r.img);
                *reinterpret_cast<int4*>(d + 4) = make_int4(r.outoff, r.lo, r.hi, 0);
            }
            ++qpub;
        }
    };
    auto fetch_info = [&](int q) -> RowInfo {
        const int* d = &lds.rinfo[q & 15][0];
        const int4 a4 = *reinterpret_cast<const int4*>(d);
        const int4 b4 = *reinterpret_cast<const int4*>(d + 4);
        RowInfo r;
        r.active = __builtin_amdgcn_readfirstlane(a4.x);
        r.y = __builtin_amdgcn_readfirstlane(a4.y);
        r.p0 = __builtin_amdgcn_readfirstlane(a4.z);
        r.img = __builtin_amdgcn_readfirstlane(a4.w);
        r.outoff = __builtin_amdgcn_readfirstlane(b4.x);
        r.lo = __builtin_amdgcn_readfirstlane(b4.y);
        r.hi = __builtin_amdgcn_readfirstlane(b4.z);
        return r;
    };
    Pend pend;
    RowInfo pinfo;  // row of the pending task
    int pq = -1;    // its stream row (-1: none), inputs in flight, cooked at the next event
    const int xb = 128 * (wv & 1) + 2 * lane;  // band column of this lane's first pixel in its task
    auto load_group = [&](int gamma) {
        pq = 4 * gamma - 1 + (wv >> 1);
        if (pq < 0) return;
        pinfo = fetch_info(pq);
        issue_loads<NORM, SPARSE, HIN>(pend, pinfo, xb, geo, gd, blur, hin, sparse);
    };
    auto cook_pending = [&]() {
        if (pq < 0) return;
        const bool yint = pinfo.y >= 1 && pinfo.y + 1 < geo.H;
        const bool xfull = pinfo.active && pinfo.p0 + BW <= geo.W;
        if (yint && xfull) cook_task<NORM, SPARSE, HIN, true, true>(pend, pinfo, pq, xb, geo, lds);
        else if (xfull) cook_task<NORM, SPARSE, HIN, false, true>(pend, pinfo, pq, xb, geo, lds);
        else cook_task<NORM, SPARSE, HIN, false, false>(pend, pinfo, pq, xb, geo, lds);
        if ((wv & 1) == 0 && lane == 0)
            *reinterpret_cast<int4*>(&lds.hdr[pq & 7][0]) = make_int4(pinfo.active, pinfo.outoff, pinfo.lo, pinfo.hi);
    };
    // prologue: descriptors of groups 0..2, group 0 cooked, group 1 requested
    publish_upto(10);
    wg_barrier();
    load_group(0);
    cook_pending();
    load_group(1);
    wg_barrier();

    int tau3 = 0;  // tau mod 3
    int tau_cur = 0;
    // ---- per-slot events: zero inactive rows, retire (write level n_iter), inject the next stream row
    auto slot_events = [&](auto JT, f2& v0, f2& v1, f2& n20, f2& n21) -> bool {
        constexpr int j = decltype(JT)::value;
        if (!act[j]) { v0 = f2{0.f, 0.f}; v1 = f2{0.f, 0.f}; }
        const bool ret = rcnt == j && act[j];
        // phase 0 = the slot's row completed level 24 and the next stream row enters.  (wave 7, slot 3) has
        // phi = 24 == 0 (mod 24): its counter is also 0 at step 0, before its first row exists.
        const bool inj = cnt0 == j && tau_cur >= 3 * wv + j;
        if (ret | inj) {
            // every LDS read of this event is issued before the first one is consumed: one round trip, not four
            const int q = 4 * (wv + NW * qgen) + j;
            const int cb = q & 7;
            int4 md = make_int4(0, -1, 0, 0), hd = md;
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f), ha = h;
            const f2 o0 = v0, o1 = v1;  // the completed value (written below if the row retires)
            if (ret) md = *reinterpret_cast<const int4*>(&lds.meta[wv][j][0]);
            if (inj) {
                hd = *reinterpret_cast<const int4*>(&lds.hdr[cb][0]);
                h = *reinterpret_cast<const float4*>(&lds.h0[q & 7][4 * lane]);
                if (j > 0) ha = *reinterpret_cast<const float4*>(&lds.h0[(q - 1) & 7][4 * lane]);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float4 t = *reinterpret_cast<const float4*>(&lds.cook[cb][k][4 * lane]);
                    Wt[j][k][0] = f2{t.x, t.y};
                    Wt[j][k][1] = f2{t.z, t.w};
                }
            }
            if (ret) {  // this row just completed level n_iter: write it
                const int outoffj = __builtin_amdgcn_readfirstlane(md.y);
                const int oloj = __builtin_amdgcn_readfirstlane(md.z);
                const int ohij = __builtin_amdgcn_readfirstlane(md.w);
                const int xb = 4 * lane;
                if (outoffj >= 0 && xb >= oloj && xb < ohij)
                    *reinterpret_cast<float4*>(out + (size_t)outoffj + xb) = make_float4(o0.x, o1.x, o0.y, o1.y);
            }
            if (inj) {
                *reinterpret_cast<int4*>(&lds.meta[wv][j][0]) = hd;  // kept for this row's retirement (same wave: in order)
                act[j] = __builtin_amdgcn_readfirstlane(hd.x);
                v0 = f2{h.x, h.y};
                v1 = f2{h.z, h.w};
                // accumulator of level 1: c' + self taps of H0(q) (+ above taps of H0(q-1), same wave)
                n20 = Wt[j][8][0];
                n21 = Wt[j][8][1];
                const Shift s = mk_shift(v0, v1);
                push_self(Wt[j], s, n20, n21);
                if (j > 0) {
                    const Shift sa = mk_shift(f2{ha.x, ha.y}, f2{ha.z, ha.w});
                    push_above(Wt[j], sa, n20, n21);
                }
                if (j == R - 1) ++qgen;
            }
        }
        return inj;
    };

    // One propagation step of this wave's four rows.  EV = false is the event-free fast path: all
    // four slots hold real rows and none retires or is replaced this step.
    auto step = [&](auto PT, auto ET) {
        constexpr int PAR = decltype(PT)::value;
        constexpr bool EV = decltype(ET)::value;
        f2 (&N1)[R][2] = S[PAR];
        f2 (&N2)[R][2] = S[PAR ^ 1];
        // boundary rows published by the neighbouring waves in the previous step
        const float4 tq = *reinterpret_cast<const float4*>(&lds.bnd[PAR ^ 1][(wv + NW - 1) & (NW - 1)][1][4 * lane]);
        const float4 bq = *reinterpret_cast<const float4*>(&lds.bnd[PAR ^ 1][(wv + 1) & (NW - 1)][0][4 * lane]);
        Shift s;
        bool inj = false;
        // received rows: below taps for slot 3 (next block's top row), above taps for slot 0 (prev block's bottom row)
        s = mk_shift(f2{bq.x, bq.y}, f2{bq.z, bq.w});
        push_below(Wt[3], s, N1[3][0], N1[3][1]);
        s = mk_shift(f2{tq.x, tq.y}, f2{tq.z, tq.w});
        push_above(Wt[0], s, N1[0][0], N1[0][1]);
        // slot 3 completes
        if (EV) inj = slot_events(std::integral_constant<int, 3>{}, N1[3][0], N1[3][1], N2[3][0], N2[3][1]);
        *reinterpret_cast<float4*>(&lds.bnd[PAR][wv][1][4 * lane]) = make_float4(N1[3][0].x, N1[3][0].y, N1[3][1].x, N1[3][1].y);
        s = mk_shift(N1[3][0], N1[3][1]);
        push_below(Wt[2], s, N1[2][0], N1[2][1]);
        if (!inj) push_self(Wt[3], s, N2[3][0], N2[3][1]);
        // slot 2 completes
        if (EV) inj = slot_events(std::integral_constant<int, 2>{}, N1[2][0], N1[2][1], N2[2][0], N2[2][1]);
        s = mk_shift(N1[2][0], N1[2][1]);
        push_below(Wt[1], s, N1[1][0], N1[1][1]);
        if (!inj) push_self(Wt[2], s, N2[2][0], N2[2][1]);
        N1[3][0] = Wt[3][8][0]; N1[3][1] = Wt[3][8][1];
        push_above(Wt[3], s, N1[3][0], N1[3][1]);
        // slot 1 completes
        if (EV) inj = slot_events(std::integral_constant<int, 1>{}, N1[1][0], N1[1][1], N2[1][0], N2[1][1]);
        s = mk_shift(N1[1][0], N1[1][1]);
        push_below(Wt[0], s, N1[0][0], N1[0][1]);
        if (!inj) push_self(Wt[1], s, N2[1][0], N2[1][1]);
        N1[2][0] = Wt[2][8][0]; N1[2][1] = Wt[2][8][1];
        push_above(Wt[2], s, N1[2][0], N1[2][1]);
        // slot 0 completes
        if (EV) inj = slot_events(std::integral_constant<int, 0>{}, N1[0][0], N1[0][1], N2[0][0], N2[0][1]);
        *reinterpret_cast<float4*>(&lds.bnd[PAR][wv][0][4 * lane]) = make_float4(N1[0][0].x, N1[0][0].y, N1[0][1].x, N1[0][1].y);
        s = mk_shift(N1[0][0], N1[0][1]);
        if (!inj) {
            N2[0][0] = Wt[0][8][0]; N2[0][1] = Wt[0][8][1];
            push_self(Wt[0], s, N2[0][0], N2[0][1]);
        }
        N1[1][0] = Wt[1][8][0]; N1[1][1] = Wt[1][8][1];
        push_above(Wt[1], s, N1[1][0], N1[1][1]);
        if (EV) allact = act[0] & act[1] & act[2] & act[3];
    };

    auto do_step = [&](auto PT, int tau) {
        tau_cur = tau;
        TIM(0);
        const bool events = (cnt0 < R && tau >= 3 * wv + cnt0) || rcnt < R || !allact;
        if (events) { step(PT, std::true_type{}); TIM(3); }
        else { step(PT, std::false_type{}); TIM(2); }
        cnt0 = (cnt0 + 1 == LV) ? 0 : cnt0 + 1;
        rcnt = (rcnt + 1 == LV) ? 0 : rcnt + 1;
        if (tau3 == 2) {                  // the group entering from the next step on
            const int gamma = (tau + 1) / 3;
            cook_pending();               // its inputs were requested one event (3 steps) ago
            load_group(gamma + 1);        // request the next group's
            publish_upto(4 * gamma + 10); // descriptors up to group gamma + 2 (ring of 16)
        }
        tau3 = tau3 == 2 ? 0 : tau3 + 1;
        TIM(1);
        wg_barrier();
        TIM(4);
    };
    for (int tau = 0; tau <= last_step; tau += 2) {
        do_step(std::integral_constant<int, 0>{}, tau);
        if (tau + 1 > last_step) break;
        do_step(std::integral_constant<int, 1>{}, tau + 1);
    }
    TIM_FLUSH();
}

#ifdef CSPN_DBG_TIMING
}  // namespace
}  // namespace cspn
extern "C" int cspn_debug_timing(long long* dst) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(cspn::g_tim), sizeof(long long) * 64);
}
namespace cspn {
namespace {
#endif

int bands_of(int W, int halo) {
    if (W <= BW) return 1;
    int nb = 0, lo = 0;
    while (lo < W) {
        const int p0 = lo == 0 ? 0 : lo - halo;
        lo = (p0 + BW >= W) ? W : p0 + BW - halo;
        ++nb;
    }
    return nb;
}

template <int NORM, bool SPARSE>
void launch_pass(bool hin_differs, int grid, hipStream_t st, const float* g, const float* blur, const float* hin,
                 const float* sparse, float* out, const Geo& geo) {
    if (hin_differs)
        hipLaunchKernelGGL((cspn2d_fused_kernel<NORM, SPARSE, true>), dim3(grid), dim3(NT), 0, st, g, blur, hin, sparse, out, geo);
    else
        hipLaunchKernelGGL((cspn2d_fused_kernel<NORM, SPARSE, false>), dim3(grid), dim3(NT), 0, st, g, blur, hin, sparse, out, geo);
}

}  // namespace

bool fused2d_supported(int B, int H, int W, int n_iter) {
    return B > 0 && H > 0 && W > 0 && n_iter > 0 && (W % 4) == 0;
}

static size_t ping_bytes(int B, int H, int W, int n_iter) {
    // one ping buffer for n_iter > 24 (passes alternate between it and `out`)
    return n_iter > LV ? (((size_t)B * H * W * sizeof(float) + 255) & ~(size_t)255) : 0;
}

size_t fused2d_workspace(int B, int H, int W, int n_iter) {
    return ping_bytes(B, H, W, n_iter);   // the assembly passes plan their streams in the kernel: no table in HBM
}

int fused2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W,
                    int n_iter, int norm, void* ws, hipStream_t st, bool use_asm, int plan_mode) {
    if (((uintptr_t)out & 15u) != 0) { set_error("fused kernel needs a 16-byte aligned output"); return CSPN_E_UNSUPPORTED; }
    const int passes = (n_iter + LV - 1) / LV;
    float* pingpong = (float*)ws;
    const bool asm_ok = use_asm && tsw2d_supported(B, H, W);
    const float* hin = blur;
    int done = 0;
    for (int p = 0; p < passes; ++p) {
        // the assembly loop (round 5: every n_iter) runs the remainder n_iter % 24 FIRST -- its short pass is a first pass, the full passes
        // continue from any level --; the compiler-generated kernel below (narrow images, experiment switches) runs it last, as before
        const int rem = n_iter % LV;
        const int n = asm_ok ? ((p == 0 && rem) ? rem : LV) : ((n_iter - done) < LV ? (n_iter - done) : LV);
        // the last pass writes `out`; earlier passes alternate so that no pass reads what it writes
        float* dst = ((passes - 1 - p) % 2 == 0) ? out : pingpong;
        if (asm_ok && n < LV) {
            if (int e = tsw2d_pass(g, blur, blur, sparse, dst, B, H, W, norm, st, nullptr, plan_mode & 7, n)) return e;
            hin = dst;
            done += n;
            continue;
        }
        if (asm_ok && n == LV) {
#ifdef CSPN_EXPERIMENTS
            if ((plan_mode & 7) == 3) {   // the round-3 loop (experiment builds only)
                if (!tsw3_supported(B, H, W, sparse != nullptr, hin != blur)) { set_error("round-3 loop: unsupported call"); return CSPN_E_UNSUPPORTED; }
                if (int e = tsw3_pass(g, blur, hin, sparse, dst, B, H, W, norm, st)) return e;
            } else
#endif
            if (hin == blur && !(plan_mode & 8) && ((plan_mode & 16) ? tsw4_supported(B, H, W) : tsw4_preferred(B, H, W, sparse != nullptr))) {   // round 6: 12 waves x 3 rows, three waves per SIMD
                if (int e = tsw4_pass(g, blur, sparse, dst, B, H, W, norm, st, plan_mode)) return e;
            } else
            if (int e = tsw2d_pass(g, blur, hin, sparse, dst, B, H, W, norm, st, nullptr, plan_mode & 7)) return e;
            hin = dst;
            done += n;
            continue;
        }
        Geo geo;
        geo.B = B; geo.H = H; geo.W = W; geo.n_iter = n;
        geo.halo = 4 * ((n + 3) / 4);
        geo.nb = bands_of(W, geo.halo);
        const long long total = (long long)B * geo.nb * H;
        long long grid = total / MIN_ROWS_PER_WG;
        if (grid < 1) grid = 1;
        if (grid > num_cus()) grid = num_cus();
        const bool hd = hin != blur;
        const bool sp = sparse != nullptr;
        switch (norm * 2 + (sp ? 1 : 0)) {
            case 0: launch_pass<0, false>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 1: launch_pass<0, true>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 2: launch_pass<1, false>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 3: launch_pass<1, true>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 4: launch_pass<2, false>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 5: launch_pass<2, true>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            case 6: launch_pass<3, false>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
            default: launch_pass<3, true>(hd, (int)grid, st, g, blur, hin, sparse, dst, geo); break;
        }
        if (int e = check_launch("cspn2d_fused_kernel")) return e;
        hin = dst;
        done += n;
    }
    return 0;
}

}  // namespace cspn
