// cspn2d_tsw_desc.h -- the 16-byte row descriptors and the workgroup header of the assembly loop (cspn2d_tsw.hip), built by
// every workgroup for itself from the plan (cspn2d_tsw_plan.h); tools/tswgen/plan.py is the numpy twin, the test-hook library
// dumps the device version for the comparison (csrc/cspn_test_hooks.hip).
#pragma once
#include "cspn2d_tsw_plan.h"

namespace cspn {
namespace tswplan {

// descriptor flags (tools/tswgen/kernel.py F_*)
enum { F_ACTIVE = 0, F_UP = 1, F_DN = 2, F_FIRST = 3, F_LAST = 4, F_OWNED = 5, F_PLAIN = 6 };

// descriptor of stream row q of workgroup wg (zeros: separator / padding row); *Q = number of stream rows of the share
// dword 0:1 byte offset of (image, channel 0, y, p0) in the guidance tensor, 2 the same in a 1-channel tensor,
// 3 flags | (lo - p0) << 8 | (hi - p0) << 20: the band's owned columns travel with every row (a linear plan's piece may
// continue in the next band)
__device__ __forceinline__ uint4 tsw_desc(const PlanGeo& g, int wg, int q, int* Q) {
    unsigned d[4] = {0, 0, 0, 0};
    StreamRow s;
    if (wg_stream_row(g, wg, q, s, Q)) {
        int p0, lo, hi;
        band_of(g, s.bi, p0, lo, hi);
        const unsigned long long goff = 4ull * ((unsigned long long)s.b * 8ull * g.H * g.W + (unsigned long long)s.y * g.W + p0);
        d[0] = (unsigned)goff;
        d[1] = (unsigned)(goff >> 32);
        d[2] = 4u * (unsigned)(s.b * g.H * g.W + s.y * g.W + p0);
        d[3] = (1u << F_ACTIVE) | ((unsigned)(s.y + 1 < g.H) << F_UP) | ((unsigned)(s.y >= 1) << F_DN) |
               ((unsigned)(p0 == 0) << F_FIRST) | ((unsigned)(p0 + BW == g.W) << F_LAST) |
               ((unsigned)s.owned << F_OWNED) | ((unsigned)(lo - p0) << 8) | ((unsigned)(hi - p0) << 20) |
               ((unsigned)(s.y >= 1 && s.y + 1 < g.H && p0 > 0 && p0 + BW < g.W) << F_PLAIN);
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
}

// header of a workgroup: Q, last step (-1: nothing to do), owned columns lo | hi << 16 (band relative) of the workgroup's band
// -- -1 for a linear plan: the loop takes them from the descriptor of the row it retires
__device__ __forceinline__ int4 tsw_header(const PlanGeo& g, int wg, int Q) {
    int lohi = -1;
    if (g.kind == 0) {
        int p0, lo, hi;
        band_of(g, wg_single_band(g, wg), p0, lo, hi);
        lohi = (lo - p0) | ((hi - p0) << 16);
    }
    return make_int4(Q, Q > 0 ? 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + g.n_iter : -1, lohi, 0);
}

}  // namespace tswplan
}  // namespace cspn
