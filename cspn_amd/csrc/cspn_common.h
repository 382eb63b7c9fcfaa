// cspn_common.h -- shared declarations of libcspn_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/cspn_amd.h"

namespace cspn {

// Neighbour offsets of the eight affinity channels, derived from the ZeroPad2d
// tuples at reference cspn_pytorch/models/cspn.py:105-128 (dy = 1-top, dx = 1-left):
// channel k couples output pixel p with neighbour p + (DY[k], DX[k]), and -- because
// the reference pads the affinity planes with the SAME tuples as the depth
// (cspn.py:149-167) -- its weight is read AT THE NEIGHBOUR ("neighbour-sited").
__host__ __device__ constexpr int dy2(int k) { return k < 3 ? 1 : (k < 5 ? 0 : -1); }
__host__ __device__ constexpr int dx2(int k) {
    return (k == 0 || k == 3 || k == 5) ? 1 : ((k == 1 || k == 6) ? 0 : -1);
}

// sign() of reference cspn.py:64 (NaN stays NaN like torch.sign -> here NaN compares false -> s itself)
__device__ __forceinline__ float signf(float s) { return s > 0.f ? 1.f : (s < 0.f ? -1.f : s); }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// CUs of the current device (queried once per device and kept: idempotent memoisation, no other state)
int num_cus();

// ---- stepwise path (one launch per iteration; general shapes) ----
size_t stepwise2d_workspace(int B, int H, int W, int n_iter);
int stepwise2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W,
                       int n_iter, int norm, void* ws, hipStream_t st);
// reference affinity_normalization (cspn.py:85-144) as a stand-alone kernel: g [B,8,H,W] -> gate_wb [B,8,H,W] (norm 8SUM / 8SUM_ABS)
int normalize2d(const float* g, float* wb, int B, int H, int W, int norm, hipStream_t st);
// W % 4 != 0: the fused path on rows padded to a multiple of 4 columns (cspn2d_stepwise.hip)
bool padded2d_supported(int B, int H, int W, int n_iter);
size_t padded2d_workspace(int B, int H, int W, int n_iter);
int padded2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W, int n_iter, int norm, void* ws,
                     hipStream_t st);
size_t stepwise3d_workspace(int B, int D, int H, int W, int n_iter);
int stepwise3d_forward(const float* g, const float* feat, const float* sparse, float* out, int B, int D, int H,
                       int W, int n_iter, int norm, void* ws, hipStream_t st, int algo = 0);
size_t forward3d_workspace(int B, int D, int H, int W, int n_iter, int norm, bool has_sparse);

// ---- 3D, gates resident in registers for all steps (cspn3d_persistent.hip); Paddle contract only ----
bool persistent3d_supported(int B, int D, int H, int W, int n_iter);
size_t persistent3d_workspace(int B, int D, int H, int W);
int persistent3d_forward(const float* gate, const float* feat, float* out, int B, int D, int H, int W, int n_iter, void* ws,
                         hipStream_t st);
// the folded form of the normalising / masked modes: wf = 26 planes w' + the constant term c' ([27][B*V], fold3d_kernel)
int persistent3d_forward_folded(const float* wf, const float* feat, float* out, int B, int D, int H, int W, int n_iter, void* ws,
                                hipStream_t st);
// launch options only the test-hook library sets (csrc/cspn_test_hooks.hip): mute = the workgroup that never publishes its
// boundary (its neighbours then run into the poll timeout), coop = hipLaunchCooperativeKernel instead of the event chain
struct P3Options { int mute = -1; bool coop = false; bool placement = true; /* false: tiles in plain workgroup order (A/B of the XCD-aware placement) */
                   bool write_through = false; /* true: no L2-resident stores, every published row goes write-through (A/B, tests) */ };
// the same run for the backward: adjoint = transposed operator; levels + (lv0 + it * lvs) volumes receive step it < n_iter
// C > 1: feat / out / the level volumes hold C value channels per volume ([B][C][V]) on shared gates (the MULTI instantiations)
int persistent3d_run(const float* gate, const float* feat, float* out, float* levels, int lv0, int lvs, bool adjoint, int B, int D,
                     int H, int W, int n_iter, void* ws, hipStream_t st, const P3Options& opt = P3Options(), int C = 1);
int persistent3d_error_word(const void* ws, int B, int D, int H, int W);
void persistent3d_geo(int B, int D, int H, int W, int n_iter, int* info);   // (test-hook library)
// C value channels per volume that share the gates ([B][C][V] value tensors, [B][26][V] gates used as given)
bool persistent3d_multi_supported(int B, int C, int D, int H, int W, int n_iter);
int persistent3d_forward_multi(const float* gate, const float* feat, float* out, int B, int C, int D, int H, int W, int n_iter, void* ws,
                               hipStream_t st);

// sticky per-device status of the persistent launches: != 0 once after a launch gave up (a workgroup waited in vain for a
// neighbour: not all workgroups resident); read without synchronisation from a pinned host word, cleared by the read
int persistent3d_take_status();

// ---- backward of the 3D op, Paddle contract only (cspn3d_backward.hip) ----
// C > 1: feat / gout / gf are [B][C][V] on shared gates; gg [B][26][V] is the sum over the channels
size_t backward3d_workspace(int B, int D, int H, int W, int n_iter, int C = 1);
int backward3d(const float* g, const float* feat, const float* gout, float* gg, float* gf, int B, int D, int H, int W, int n_iter,
               void* ws, hipStream_t st, bool stepwise_only = false /* test-hook library: one launch per step */, int C = 1);

// ---- the producer of the path's inputs (cspn_head.hip): Unpool + 3x3 conv C -> 8 (guidance) and C -> 1 (blur) as one kernel; mode 0 raw guidance,
// 1 / 2 gate_wb of '8sum' / '8sum_abs' (the normalisation fused behind the conv) ----
size_t head_workspace(int C);
size_t head_backward_workspace(int B, int C, int h, int w);
int head_backward(const float* x, const float* w6, const float* w5, const float* gg, const float* gb, float* dx, float* dw6, float* dw5, int B, int C, int h,
                  int w, int H, int W, void* ws, hipStream_t st);
int head_forward(const float* x, const float* w6, const float* w5, float* gout, float* bout, int B, int C, int h, int w, int H, int W, int mode,
                 void* ws, hipStream_t st);

// ---- fused path (all iterations in one launch; time-skewed wave ring) ----
bool fused2d_supported(int B, int H, int W, int n_iter);
size_t fused2d_workspace(int B, int H, int W, int n_iter);
// plan_mode (test-hook library only; the ABI passes 0): 0 the linear plan, 1 the same without XCD-aware placement, 2 band groups;
// + 8: the 8-wave x 4-row loop of rounds 1-5 (cspn2d_tsw.hip) also for the passes the round-6 loop (cspn2d_tsw4.hip) would take;
// + 16: the round-6 loop for every full first pass it supports (also the short streams on which the dispatcher prefers the other)
int fused2d_forward(const float* g, const float* blur, const float* sparse, float* out, int B, int H, int W,
                    int n_iter, int norm, void* ws, hipStream_t st, bool use_asm = true, int plan_mode = 0);

// ---- the same ring with the main loop in gfx950 assembly (cspn2d_tsw.hip); one pass = 24 iterations, or -- a FIRST pass only
// (hin == blur, no history) -- n_early = 1 .. 23 of them ----
bool tsw2d_supported(int B, int H, int W);
int tsw2d_pass(const float* gd, const float* blur, const float* hin, const float* sparse, float* out, int B, int H,
               int W, int norm, hipStream_t st, float* hist = nullptr, int plan_mode = 0, int n_early = 0);
// ---- round 6: the same ring as 12 waves x 3 rows at 168 VGPRs -- three waves per SIMD (cspn2d_tsw4.hip, tools/tswgen/kernel4.py): FIRST
// passes of exactly 24 iterations ----
bool tsw4_supported(int B, int H, int W);
bool tsw4_preferred(int B, int H, int W, bool sparse);   // long streams on the linear plan: where it is the faster of the two rings
int tsw4_pass(const float* gd, const float* blur, const float* sparse, float* out, int B, int H, int W, int norm, hipStream_t st,
              int plan_mode = 0);
#ifdef CSPN_EXPERIMENTS
// ---- experiments kept out of the default build (make EXPERIMENTS=1): the round-3 loop (cspn2d_tsw3.hip: LDS-DMA row slots;
// ties with the loop above on long streams, slower on short ones: profiles/r03_perf_notes.md) and the sited8 guidance layout ----
bool tsw3_supported(int B, int H, int W, bool sparse, bool hin_differs);
int tsw3_pass(const float* gd, const float* blur, const float* hin, const float* sparse, float* out, int B, int H, int W,
              int norm, hipStream_t st);
#endif
int guidance_to_sited8(const float* g, float* out, int B, int H, int W, int norm, hipStream_t st);
int tsw2d_pass_sited8(const float* g8, const float* blur, const float* sparse, float* out, int B, int H, int W, int norm,
                      hipStream_t st);
int tsw2d_adjoint_pass(const float* wf, const float* a_in, float* a0, int B, int H, int W, hipStream_t st, float* hist);

// ---- backward of the 2D op (cspn2d_backward.hip) ----
size_t backward2d_workspace(int B, int H, int W, int n_iter);
int backward2d(const float* g, const float* blur, const float* sparse, const float* gout, float* gg, float* gb, int B, int H,
               int W, int n_iter, int norm, void* ws, hipStream_t st);
// training mode: the forward keeps its checkpoints (every fourth level + the folded coefficients; 24-iteration passes the assembly
// kernel takes), the backward starts there
size_t history2d_bytes(int B, int H, int W, int n_iter);  // 0: not available for this shape
int forward2d_history(const float* g, const float* blur, const float* sparse, float* out, void* history, int B, int H, int W,
                      int n_iter, int norm, void* ws, hipStream_t st);
size_t backward2d_history_workspace(int B, int H, int W);
int backward2d_history(const float* g, const float* blur, const float* sparse, const float* gout, const void* history, float* gg,
                       float* gb, int B, int H, int W, int n_iter, int norm, void* ws, hipStream_t st);

}  // namespace cspn
