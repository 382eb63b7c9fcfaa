// cspn2d_tsw_plan.h -- how the time-skewed wave ring kernel (cspn2d_tsw.hip: forward, history and adjoint variants; the
// round-3 experiment cspn2d_tsw3.hip) cuts B x H x W into 256-column bands and per-workgroup row streams.
// tools/tswgen/plan.py is the numpy twin.  Every workgroup builds its row-descriptor table from these functions, straight
// into LDS.
//
// Two kinds of plan:
//   kind 0 "band groups": a group of nb workgroups owns an equal range of the B*H image rows, one band each (a workgroup's
//          band is a constant: the history / adjoint variants of the loop and the round-3 experiment need that).
//   kind 1 "linear" (round 4; the forward passes): the nb * B * H band rows are put in ONE order -- chunks of kimg images,
//          inside a chunk band after band -- and cut into one contiguous piece per CU so that the longest STREAM (rows +
//          warm-up / cool-down rows + separators) is as short as possible (binary search over the length, greedy cuts:
//          optimal, because a stream's length is monotone in both of its ends).  A cut at an image edge costs no warm-up
//          rows, and the optimiser finds them: BASELINE config 3 (64 x 304 x 1216, 6 bands, 256 CUs) becomes 256 pieces of
//          exactly 1.5 (image, band) units = 481 stream rows, against 513 with 42 groups x 6 bands (252 CUs, two mid-image
//          cuts per share).  kimg is chosen so that a chunk holds a whole number of pieces per band: the pieces of
//          neighbouring bands then stream the same rows at the same time and are placed on the same XCD (the columns both
//          read come from HBM once).  A piece may continue in the next band (the 4 CUs that 42 groups left idle take the 64th
//          image that way): descriptors carry the band per row, the loop re-derives its owned-lane mask when it changes.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "cspn_common.h"

namespace cspn {
namespace tswplan {

constexpr int BW = 256;
constexpr int LV = 24;
constexpr int NT = 512;
constexpr int MIN_ROWS_PER_WG = 16;
constexpr int MAX_CUT = 256;   // pieces of a linear plan (one per CU); more pieces -> band groups

struct PlanGeo {
    int B, H, W, n_iter, nb, halo, n_wg, stride;  // stride: descriptors per workgroup (PADF + max stream + PADB)
    // kind 0, XCD-aware placement (workgroup id -> XCD is round robin: id = xcd + 8 * slot): the nb workgroups of a group sit
    // on one XCD, so the columns two neighbouring bands both read are fetched from HBM once and hit in that XCD's L2 afterwards.
    // gpx groups per XCD fill gpx * nb of its per_xcd slots; the left-over slots of all XCDs form `extra` more groups.
    // kind 1: xcd != 0 -> piece p runs as workgroup (p % per_xcd) * 8 + p / per_xcd (runs of per_xcd pieces share an XCD).
    int xcd, ng, per_xcd, gpx, extra;
    int kind, kimg;         // kind 1: images per chunk
    int cut[MAX_CUT + 1];   // kind 1: piece p = positions [cut[p], cut[p + 1]) of the linear order
};

__host__ __device__ inline void band_of(const PlanGeo& g, int bi, int& p0, int& lo, int& hi) {
    if (bi == 0) { p0 = 0; lo = 0; }
    else { lo = (BW - g.halo) + (bi - 1) * (BW - 2 * g.halo); p0 = lo - g.halo; }
    if (p0 + BW >= g.W) { p0 = g.W - BW; hi = g.W; }  // the last band ends exactly at the image edge
    else hi = p0 + BW - g.halo;
}

inline int bands_of(int W, int halo) {
    PlanGeo g{};
    g.W = W; g.halo = halo;
    for (int bi = 0;; ++bi) {
        int p0, lo, hi;
        band_of(g, bi, p0, lo, hi);
        if (hi >= W) return bi + 1;
    }
}

// ---- kind 0: workgroup wg = (group G, band bi); group G owns the image rows [r0, r1) ----------------------------------------
__host__ __device__ __forceinline__ bool tsw_wg_share(const PlanGeo& g, int wg, int& bi, int& r0, int& r1) {
    int G = wg / g.nb;
    bi = wg - G * g.nb;
    if (g.xcd) {
        const int x = wg & 7, sl = wg >> 3;
        if (sl < g.gpx * g.nb) {
            G = x * g.gpx + sl / g.nb;
            bi = sl % g.nb;
        } else {
            const int t = (sl - g.gpx * g.nb) * 8 + x;  // left-over slots, all XCDs
            G = t < g.extra * g.nb ? 8 * g.gpx + t / g.nb : g.ng;  // G == ng: idle workgroup
            bi = t % g.nb;
        }
    }
    if (G >= g.ng) { r0 = r1 = 0; return false; }
    const long long total = (long long)g.B * g.H;
    r0 = (int)(total * G / g.ng);
    r1 = (int)(total * (G + 1) / g.ng);
    return r1 > r0;
}

// ---- kind 1: the linear order.  Position x -> chunk c = x / (kimg * H * nb), inside the chunk band after band ----------------
struct Run { int bi, ra, rb; };   // image rows [ra, rb) (global: image * H + y) of band bi

// the run that starts at position x of a piece ending at position xe -> the position behind it
__host__ __device__ __forceinline__ int lin_next_run(const PlanGeo& g, int x, int xe, Run& r) {
    const int Rc = g.kimg * g.H, total = g.B * g.H;
    const int c = x / (Rc * g.nb);
    const int r0 = c * Rc;
    const int rows_c = (Rc < total - r0) ? Rc : total - r0;   // the last chunk may be shorter
    const int xp = x - c * Rc * g.nb;
    r.bi = xp / rows_c;
    r.ra = r0 + xp - r.bi * rows_c;
    int n = r0 + rows_c - r.ra;
    if (n > xe - x) n = xe - x;
    r.rb = r.ra + n;
    return x + n;
}

__host__ __device__ __forceinline__ bool lin_wg_piece(const PlanGeo& g, int wg, int& xa, int& xe) {
    int p = wg;
    if (g.xcd) p = (wg & 7) * g.per_xcd + (wg >> 3);
    if (p >= g.n_wg) { xa = xe = 0; return false; }
    xa = g.cut[p];
    xe = g.cut[p + 1];
    return xe > xa;
}

// ---- streams.  A share is cut into segments at image ends (and, kind 1, where the band changes), every segment is extended
// by n_iter halo rows on both sides (clipped to the image) and segments are separated by one inactive row. -------------------
struct StreamRow { int b, y, bi; bool owned; };

// segments of the rows [ra, rb) of one band: stream row q -> (image, y, owned) if it falls into them; qq: running stream length
__host__ __device__ __forceinline__ bool run_stream_row(const PlanGeo& g, const Run& run, int q, int& qq, bool& first, StreamRow& out) {
    bool found = false;
    int b = run.ra / g.H, y0 = run.ra - b * g.H;
    for (int r = run.ra; r < run.rb; ++b, y0 = 0) {
        const int rem = run.rb - r;
        const int y1 = (g.H < y0 + rem) ? g.H : y0 + rem;
        const int ys = (y0 - g.n_iter > 0) ? y0 - g.n_iter : 0, ye = (g.H < y1 + g.n_iter) ? g.H : y1 + g.n_iter;
        if (!first) ++qq;  // separator
        first = false;
        if (q >= qq && q < qq + (ye - ys)) {
            out.b = b; out.y = ys + (q - qq); out.bi = run.bi;
            out.owned = (out.y >= y0 && out.y < y1);
            found = true;
        }
        qq += ye - ys;
        r += y1 - y0;
    }
    return found;
}

// stream row q of workgroup wg (either kind) -> found?; *Q = rows of the stream (q = -1: only the length)
__host__ __device__ __forceinline__ bool wg_stream_row(const PlanGeo& g, int wg, int q, StreamRow& out, int* Q) {
    bool found = false, first = true;
    int qq = 0;
    if (g.kind == 0) {
        Run run;
        if (tsw_wg_share(g, wg, run.bi, run.ra, run.rb)) found = run_stream_row(g, run, q, qq, first, out);
    } else {
        int x, xe;
        if (lin_wg_piece(g, wg, x, xe)) {
            while (x < xe) {
                Run run;
                x = lin_next_run(g, x, xe, run);
                StreamRow o;
                if (run_stream_row(g, run, q, qq, first, o)) { out = o; found = true; }
            }
        }
    }
    *Q = qq;
    return found;
}

// the band of a workgroup whose stream stays in one band (kind 0: always); -1: the stream visits more than one band
__host__ __device__ __forceinline__ int wg_single_band(const PlanGeo& g, int wg) {
    if (g.kind == 0) {
        int bi, r0, r1;
        tsw_wg_share(g, wg, bi, r0, r1);
        return bi;
    }
    int x, xe, bi = -2;
    if (!lin_wg_piece(g, wg, x, xe)) return 0;
    while (x < xe) {
        Run run;
        x = lin_next_run(g, x, xe, run);
        if (bi == -2) bi = run.bi;
        else if (bi != run.bi) return -1;
    }
    return bi;
}

// (kind 0, kept for the round-3 experiment) stream row q of the share [r0, r1) -> image b, row y, owned?; *Q = stream rows
__device__ __forceinline__ bool tsw_stream_row(const PlanGeo& g, int r0, int r1, int q, int& b_out, int& y_out, bool& owned, int* Q) {
    Run run{0, r0, r1};
    StreamRow o{};
    bool first = true;
    int qq = 0;
    const bool found = run_stream_row(g, run, q, qq, first, o);
    if (found) { b_out = o.b; y_out = o.y; owned = o.owned; }
    *Q = qq;
    return found;
}

// ---- kind 0 --------------------------------------------------------------------------------------------------------------------
// padf / padb: inactive descriptors in front of / behind a stream; tab_max: descriptors that fit in the kernel's LDS table
inline PlanGeo make_geo(int B, int H, int W, int padf, int padb, int tab_max, int ncu = 0, bool allow_xcd = true) {
    PlanGeo g;
    memset(&g, 0, sizeof(g));
    if (ncu <= 0) ncu = num_cus();
    g.B = B; g.H = H; g.W = W; g.n_iter = LV;
    g.halo = 4 * ((LV + 3) / 4);
    g.nb = bands_of(W, g.halo);
    const long long total = (long long)B * H;  // image rows; every group of nb workgroups takes an equal share of them
    long long ng = total / MIN_ROWS_PER_WG;
    if (ng > ncu / g.nb) ng = ncu / g.nb;
    if (ng < 1) ng = 1;
    const long long ng_cu = ng;
    for (;;) {  // a share's descriptor table must fit in the LDS left over by the ring (tools/tswgen/plan.py plan_geo)
        const long long share = (total + ng - 1) / ng;
        const long long stride = padf + share + (share / H + 2) * (2 * LV + 1) + padb;
        if (stride <= tab_max) { g.stride = (int)stride; break; }
        ng += ng / 8 > 1 ? ng / 8 : 1;
    }
    g.ng = (int)ng;
    g.n_wg = (int)(ng * g.nb);
    // every CU busy with whole groups (the usual case for full batches): place the groups XCD by XCD
    if (allow_xcd && ng == ng_cu && ng == ncu / g.nb && ncu % 8 == 0 && (ncu / 8) / g.nb >= 1) {
        g.xcd = 1;
        g.per_xcd = ncu / 8;
        g.gpx = g.per_xcd / g.nb;
        g.extra = (8 * (g.per_xcd - g.gpx * g.nb)) / g.nb;
        g.ng = 8 * g.gpx + g.extra;
        g.n_wg = 8 * g.per_xcd;  // the few slots that belong to no group get an empty stream
        const long long share = (total + g.ng - 1) / g.ng;
        g.stride = (int)(padf + share + (share / H + 2) * (2 * LV + 1) + padb);
        if (g.stride > tab_max) { g.xcd = 0; g.per_xcd = g.gpx = g.extra = 0; g.ng = (int)ng; g.n_wg = (int)(ng * g.nb); g.stride = 0; }
    }
    if (!g.xcd) {
        const long long share = (total + g.ng - 1) / g.ng;
        g.stride = (int)(padf + share + (share / H + 2) * (2 * LV + 1) + padb);
    }
    return g;
}

// ---- kind 1 --------------------------------------------------------------------------------------------------------------------
// stream rows of the piece [xa, xe)
inline int lin_stream_len(const PlanGeo& g, int xa, int xe) {
    bool first = true;
    int qq = 0;
    StreamRow o;
    for (int x = xa; x < xe;) {
        Run run;
        x = lin_next_run(g, x, xe, run);
        run_stream_row(g, run, -1, qq, first, o);
    }
    return qq;
}

// greedy cuts for stream length <= L: every piece as long as it may be.  -> pieces used (n + 1: L is too short for n pieces)
inline int lin_greedy(PlanGeo& g, int n, int total, int L) {
    int xa = 0, p = 0;
    g.cut[0] = 0;
    while (xa < total) {
        if (p == n) return n + 1;
        int lo = xa, hi = (xa + L < total) ? xa + L : total;   // a stream is never shorter than its share
        while (lo < hi) {
            const int m = lo + (hi - lo + 1) / 2;
            if (lin_stream_len(g, xa, m) <= L) lo = m; else hi = m - 1;
        }
        if (lo == xa) return n + 1;
        xa = lo;
        g.cut[++p] = xa;
    }
    for (int i = p + 1; i <= n; ++i) g.cut[i] = total;
    return p;
}

inline bool make_geo_linear_uncached(PlanGeo& g, int B, int H, int W, int padf, int padb, int tab_max, int ncu, bool allow_xcd) {
    memset(&g, 0, sizeof(g));
    g.B = B; g.H = H; g.W = W; g.n_iter = LV;
    g.halo = 4 * ((LV + 3) / 4);
    g.nb = bands_of(W, g.halo);
    g.kind = 1;
    const long long total_ll = (long long)B * H * g.nb;
    if (total_ll >= (1ll << 30)) return false;
    const int total = (int)total_ll;
    int n = total / MIN_ROWS_PER_WG;
    if (n > ncu) n = ncu;
    if (n > MAX_CUT) n = MAX_CUT;
    if (n < 1) n = 1;
    // chunks of kimg images such that a chunk holds (nearly) a whole number of pieces per band: the pieces of neighbouring bands
    // then cover the same rows at the same time
    {
        const double rho = (double)total / n;   // band rows per piece
        double best = 1e30;
        g.kimg = 1;
        for (int k = 1; k <= 8 && k <= B; ++k) {
            const double m = (double)k * H / rho;
            double mr = (double)(long long)(m + 0.5);
            if (mr < 1) mr = 1;
            const double d = (m > mr ? m - mr : mr - m) / mr;
            if (d < best - 1e-9) { best = d; g.kimg = k; }
        }
    }
    const int longest = tab_max - padf - padb;   // stream rows a table holds
    if (total / n > longest) return false;       // huge batches: more pieces than CUs -> band groups
    int lo = total / n, hi = total / n + 1 + (total / n / H + 2) * (2 * LV + 1);
    if (lo < 1) lo = 1;
    while (lin_greedy(g, n, total, hi) > n) hi += hi;   // (cannot happen for the bound above; cheap to be safe)
    while (lo < hi) {
        const int m = lo + (hi - lo) / 2;
        if (lin_greedy(g, n, total, m) <= n) hi = m; else lo = m + 1;
    }
    if (lo > longest) return false;   // huge batches: more pieces than CUs -> band groups
    const int used = lin_greedy(g, n, total, lo);
    (void)used;
    g.n_wg = n;
    g.stride = padf + lo + padb;
    if (allow_xcd && n == ncu && ncu % 8 == 0) {
        g.xcd = 1;
        g.per_xcd = ncu / 8;
    }
    return true;
}

// The forward passes' plan.  Falls back to band groups when the linear plan does not apply (n_wg would exceed MAX_CUT).
// The optimiser costs ~1 ms of host time: the last plans are kept per thread (no shared mutable state).
inline const PlanGeo& make_geo_linear(int B, int H, int W, int padf, int padb, int tab_max, int mode /*0: default, 1: no XCD placement, 2: band groups*/) {
    struct Entry { int key[8]; PlanGeo g; };
    constexpr int N = 4;
    thread_local Entry cache[N];
    thread_local int next = 0;
    const int ncu = num_cus();
    const int key[8] = {1, B, H, W, padf, padb, tab_max, ncu * 4 + mode};
    for (int i = 0; i < N; ++i)
        if (!memcmp(cache[i].key, key, sizeof(key))) return cache[i].g;
    Entry& e = cache[next];
    next = (next + 1) % N;
    memcpy(e.key, key, sizeof(key));
    if (mode == 2 || !make_geo_linear_uncached(e.g, B, H, W, padf, padb, tab_max, ncu, mode != 1))
        e.g = make_geo(B, H, W, padf, padb, tab_max, ncu, mode != 1);
    return e.g;
}

}  // namespace tswplan
}  // namespace cspn
