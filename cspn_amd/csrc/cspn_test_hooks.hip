// cspn_test_hooks.hip -> libcspn_amd_hooks.so: everything the tests and the measuring tools need that is NOT part of the
// product's ABI.  The product library (libcspn_amd.so) exports no cspn_debug_* symbol, reads no environment variable and keeps no
// test state; this library links against it and reaches the same code through internal entry points that take the test's choice
// as an ARGUMENT (plan mode, muted workgroup, launch kind, one launch per step), so one copy of the product code runs in the
// process and nothing global is switched.  Built by the same Makefile; loaded by cspn_amd._lib.load_hooks().
#include "cspn_common.h"
#include "cspn2d_tsw_desc.h"
#include "cspn2d_tsw_gen.inc"   // only the TSW_PADF / TSW_PADB / TSW_TAB_MAX_ROWS constants are used here

using namespace cspn;
using namespace cspn::tswplan;

namespace {

// the same descriptor functions every workgroup of cspn2d_tsw_kernel runs, one thread per (workgroup, table entry), into global memory
__global__ void cspn2d_plan_dump_kernel(int4* __restrict__ hdr, uint4* __restrict__ tab, PlanGeo g) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= g.n_wg * g.stride) return;
    const int wg = gid / g.stride, e = gid - wg * g.stride;
    int Q = 0;
    tab[(size_t)wg * g.stride + e] = tsw_desc(g, wg, e - TSW_PADF, &Q);
    if (e == 0) hdr[wg] = tsw_header(g, wg, Q);
}

PlanGeo geo_of(int B, int H, int W, int plan_mode, int hist) {
    if (hist) return make_geo(B, H, W, TSW_PADF, TSW_PADB, TSW_TAB_MAX_ROWS, 0, plan_mode != 1);
    return make_geo_linear(B, H, W, TSW_PADF, TSW_PADB, TSW_TAB_MAX_ROWS, plan_mode);
}

}  // namespace

extern "C" {

// the plan a 24-iteration pass of this shape gets (hist != 0: the history / adjoint variants' band groups).
// info[8] = kind, n_wg, stride, kimg, xcd, per_xcd, ng, nb
int cspn_debug_tsw_plan_geo(int B, int H, int W, int plan_mode, int hist, int* info) {
    const PlanGeo g = geo_of(B, H, W, plan_mode, hist);
    const int v[8] = {g.kind, g.n_wg, g.stride, g.kimg, g.xcd, g.kind ? g.per_xcd : (g.gpx | (g.extra << 8) | (g.per_xcd << 16)), g.ng, g.nb};
    for (int i = 0; i < 8; ++i) info[i] = v[i];
    return 0;
}

// the linear plan's cuts, computed for a given CU count WITHOUT a device (CPU tests hold the C++ optimiser to tools/tswgen/plan.py).
// cut: n_wg + 1 ints; -> n_wg (0: the linear plan does not apply), *kimg, *stride
int cspn_debug_tsw_plan_cuts(int B, int H, int W, int ncu, int xcd, int* cut, int* kimg, int* stride) {
    PlanGeo g;
    if (!make_geo_linear_uncached(g, B, H, W, TSW_PADF, TSW_PADB, TSW_TAB_MAX_ROWS, ncu, xcd != 0)) return 0;
    for (int i = 0; i <= g.n_wg; ++i) cut[i] = g.cut[i];
    *kimg = g.kimg;
    *stride = g.stride;
    return g.n_wg;
}

// the descriptor tables every workgroup would build for itself (hdr: n_wg x int4, tab: n_wg x stride x uint4, device pointers)
int cspn_debug_tsw_dump_plan(int B, int H, int W, int plan_mode, int hist, void* hdr, void* tab, void* stream) {
    const PlanGeo g = geo_of(B, H, W, plan_mode, hist);
    const int n = g.n_wg * g.stride;
    hipLaunchKernelGGL(cspn2d_plan_dump_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int4*)hdr, (uint4*)tab, g);
    return check_launch("cspn2d_plan_dump_kernel");
}

// cspn2d_forward_f32 (fused algo) with the plan chosen by the caller: 0 the product's, 1 without XCD-aware placement, 2 band groups
// (the round-1..3 plan), 3 (experiment builds) the round-3 loop
int cspn_debug_forward2d_plan(const float* guidance, const float* blur, const float* sparse, float* out, int B, int H, int W, int n_iter,
                              int norm_type, int plan_mode, void* ws, void* stream) {
    if (!fused2d_supported(B, H, W, n_iter)) return CSPN_E_UNSUPPORTED;
    return fused2d_forward(guidance, blur, sparse, out, B, H, W, n_iter, norm_type, ws, (hipStream_t)stream, true, plan_mode);
}

// which ring a full first pass (24 iterations) of this shape runs on: 12 (round 6, cspn2d_tsw4.hip) or 8 (cspn2d_tsw.hip); 0: not the assembly path
int cspn_debug_fused2d_ring(int B, int H, int W, int sparse) {
    if (!tsw2d_supported(B, H, W)) return 0;
    return tsw4_preferred(B, H, W, sparse != 0) ? 12 : 8;
}

// the plan of a persistent 3D launch: info[9] = tz, ty, cx, tiles, workgroups launched, bz, by, bx (block of the XCD-aware placement; 0: off), chunks
int cspn_debug_3d_geo(int B, int D, int H, int W, int n_iter, int* info) {
    persistent3d_geo(B, D, H, W, n_iter, info);
    return 0;
}

// error word of the last persistent 3D run in this workspace (0 ok, 2 neighbour-quad timeout); synchronises
int cspn_debug_3d_persistent_error(const void* ws, int B, int D, int H, int W) { return persistent3d_error_word(ws, B, D, H, W); }

// the Paddle-contract persistent launch with tile `mute` never publishing its boundary (-1: nobody); coop bit 0: cooperative launch,
// bit 1: tiles in plain workgroup order instead of the XCD-aware placement (A/B); bit 2: every published row write-through (no L2-resident stores)
int cspn_debug_3d_persistent_forward(const float* gate, const float* feat, float* out, int B, int D, int H, int W, int n_iter, int mute,
                                     int coop, void* ws, void* stream) {
    if (!persistent3d_supported(B, D, H, W, n_iter)) return CSPN_E_UNSUPPORTED;
    P3Options opt;
    opt.mute = mute;
    opt.coop = (coop & 1) != 0;
    opt.placement = (coop & 2) == 0;
    opt.write_through = (coop & 4) != 0;
    return persistent3d_run(gate, feat, out, nullptr, 0, 0, false, B, D, H, W, n_iter, ws, (hipStream_t)stream, opt);
}

// cspn3d_backward_f32 with one launch per step (A/B against the fused sweeps)
int cspn_debug_3d_backward_stepwise(const float* gate, const float* feat, const float* gout, float* gg, float* gf, int B, int D, int H, int W,
                                    int n_iter, void* ws, void* stream) {
    return backward3d(gate, feat, gout, gg, gf, B, D, H, W, n_iter, ws, (hipStream_t)stream, true);
}

// ---- SURVEY 8f-2, FIRST alternative (closed experiment; experiment builds only: make EXPERIMENTS=1): the guidance as
// 32 contiguous bytes per pixel pair record, gathered by the producer -- [B][H][W/2][8][2] floats, record of the pixel pair
// (x, x+1) = (G_0(x), G_0(x+1), G_1(x), ...), G_k(p) = g_k(p + off_k), zero outside the image.  Measured 9 % slower than the planar
// contract; these three were part of the ABI until round 4.
int cspn_debug_sited8_supported(int B, int H, int W, int n_iter) {
#ifdef CSPN_EXPERIMENTS
    return n_iter == 24 && tsw2d_supported(B, H, W) ? 1 : 0;   // (W >= 256, W % 4 == 0)
#else
    (void)B; (void)H; (void)W; (void)n_iter;
    return 0;
#endif
}

int cspn_debug_guidance_to_sited8(const float* guidance, float* guidance_s8, int B, int H, int W, int norm_type, void* stream) {
#ifdef CSPN_EXPERIMENTS
    if (!guidance || !guidance_s8 || B <= 0 || H <= 0 || W <= 0 || (W % 2) != 0 || ((uintptr_t)guidance_s8 & 15u) != 0) return CSPN_E_BADARG;
    return guidance_to_sited8(guidance, guidance_s8, B, H, W, norm_type, (hipStream_t)stream);
#else
    (void)guidance; (void)guidance_s8; (void)B; (void)H; (void)W; (void)norm_type; (void)stream;
    return CSPN_E_UNSUPPORTED;
#endif
}

int cspn_debug_forward_sited8(const float* guidance_s8, const float* blur, const float* sparse, float* out, int B, int H, int W,
                              int n_iter, int norm_type, void* stream) {
#ifdef CSPN_EXPERIMENTS
    if (!guidance_s8 || !blur || !out) return CSPN_E_BADARG;
    if (!cspn_debug_sited8_supported(B, H, W, n_iter) || (((uintptr_t)guidance_s8 | (uintptr_t)out) & 15u) != 0) return CSPN_E_UNSUPPORTED;
    return tsw2d_pass_sited8(guidance_s8, blur, sparse, out, B, H, W, norm_type, (hipStream_t)stream);
#else
    (void)guidance_s8; (void)blur; (void)sparse; (void)out; (void)B; (void)H; (void)W; (void)n_iter; (void)norm_type; (void)stream;
    return CSPN_E_UNSUPPORTED;
#endif
}

}  // extern "C"
