// cspn2d_tsw_plan.h -- how the time-skewed wave ring kernels (cspn2d_tsw.hip: round-2 loop, history / adjoint variants;
// cspn2d_tsw3.hip: round-3 loop) cut B x H x W into bands, band groups and per-workgroup row streams.  tools/tswgen/plan.py is
// the numpy twin.  Both kernels build their row-descriptor table from these functions, straight into LDS.
#pragma once
#include <cstdlib>

#include "cspn_common.h"

namespace cspn {
namespace tswplan {

constexpr int BW = 256;
constexpr int LV = 24;
constexpr int NT = 512;
constexpr int MIN_ROWS_PER_WG = 16;

struct PlanGeo {
    int B, H, W, n_iter, nb, halo, n_wg, stride;  // stride: descriptors per workgroup (PADF + max stream + PADB)
    // XCD-aware placement (workgroup id -> XCD is round robin: id = xcd + 8 * slot): the nb workgroups of a group sit on one
    // XCD, so the columns two neighbouring bands both read are fetched from HBM once and hit in that XCD's L2 afterwards.
    // gpx groups per XCD fill gpx * nb of its per_xcd slots; the left-over slots of all XCDs form `extra` more groups.
    int xcd, ng, per_xcd, gpx, extra;
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

// ---- the row-descriptor table of a workgroup (tools/tswgen/plan.py is the numpy twin, tests compare the two) ----------
// Workgroup wg = (group G, band bi): group G owns a contiguous range [r0, r1) of the B*H image rows and its nb workgroups take
// one 256-column band each, so the workgroups that read overlapping columns of the same rows run side by side and the halo
// re-reads hit in cache.  Stream row q of the workgroup: the share is cut into segments at image ends, every segment is
// extended by n_iter halo rows on both sides (clipped to the image) and segments are separated by one inactive row.
// The table is built by the kernel itself, straight into LDS (tsw_fill_table): no planning launch, no table in HBM.
__device__ __forceinline__ bool tsw_wg_share(const PlanGeo& g, int wg, int& bi, int& r0, int& r1) {
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


// stream row q of the share [r0, r1) -> image b, row y, owned? (false: separator / padding row); *Q = rows of the stream.
// The share is cut into segments at image ends, every segment is extended by n_iter halo rows on both sides (clipped to the
// image) and segments are separated by one inactive row.
__device__ __forceinline__ bool tsw_stream_row(const PlanGeo& g, int r0, int r1, int q, int& b_out, int& y_out, bool& owned, int* Q) {
    bool found = false;
    int qq = 0, b = r0 / g.H, y0 = r0 - b * g.H;
    for (int r = r0; r < r1; ++b, y0 = 0) {
        const int y1 = min(g.H, y0 + (r1 - r));
        const int ys = max(0, y0 - g.n_iter), ye = min(g.H, y1 + g.n_iter);
        if (r > r0) ++qq;  // separator
        if (q >= qq && q < qq + (ye - ys)) {
            const int y = ys + (q - qq);
            b_out = b; y_out = y; owned = (y >= y0 && y < y1);
            found = true;
        }
        qq += ye - ys;
        r += y1 - y0;
    }
    *Q = qq;
    return found;
}

inline int num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

// padf / padb: inactive descriptors in front of / behind a stream; tab_max: descriptors that fit in the kernel's LDS table
inline PlanGeo make_geo(int B, int H, int W, int padf, int padb, int tab_max) {
    PlanGeo g;
    g.B = B; g.H = H; g.W = W; g.n_iter = LV;
    g.halo = 4 * ((LV + 3) / 4);
    g.nb = bands_of(W, g.halo);
    const long long total = (long long)B * H;  // image rows; every group of nb workgroups takes an equal share of them
    long long ng = total / MIN_ROWS_PER_WG;
    if (ng > num_cus() / g.nb) ng = num_cus() / g.nb;
    if (ng < 1) ng = 1;
    static const int no_xcd = getenv("CSPN_TSW_NO_XCD") ? atoi(getenv("CSPN_TSW_NO_XCD")) : 0;  // A/B switch for tests
    const long long ng_cu = ng;
    for (;;) {  // a share's descriptor table must fit in the LDS left over by the ring (tools/tswgen/plan.py plan_geo)
        const long long share = (total + ng - 1) / ng;
        const long long stride = padf + share + (share / H + 2) * (2 * LV + 1) + padb;
        if (stride <= tab_max) { g.stride = (int)stride; break; }
        ng += ng / 8 > 1 ? ng / 8 : 1;
    }
    g.xcd = 0; g.per_xcd = g.gpx = g.extra = 0;
    g.ng = (int)ng;
    g.n_wg = (int)(ng * g.nb);
    // every CU busy with whole groups (the usual case for full batches): place the groups XCD by XCD
    if (!no_xcd && ng == ng_cu && ng == num_cus() / g.nb && num_cus() % 8 == 0 && (num_cus() / 8) / g.nb >= 1) {
        g.xcd = 1;
        g.per_xcd = num_cus() / 8;
        g.gpx = g.per_xcd / g.nb;
        g.extra = (8 * (g.per_xcd - g.gpx * g.nb)) / g.nb;
        g.ng = 8 * g.gpx + g.extra;
        g.n_wg = 8 * g.per_xcd;  // the few slots that belong to no group get an empty stream
        const long long share = (total + g.ng - 1) / g.ng;
        g.stride = (int)(padf + share + (share / H + 2) * (2 * LV + 1) + padb);
        if (g.stride > tab_max) { g.xcd = 0; g.ng = (int)ng; g.n_wg = (int)(ng * g.nb); g.stride = 0; }
    }
    if (!g.xcd) {
        const long long share = (total + g.ng - 1) / g.ng;
        g.stride = (int)(padf + share + (share / H + 2) * (2 * LV + 1) + padb);
    }
    return g;
}


}  // namespace tswplan
}  // namespace cspn
