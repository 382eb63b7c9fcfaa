/*
 * cspn_amd.h -- C ABI of the MI355X-native CSPN propagation engine (libcspn_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of XinJCheng/CSPN:
 * Affinity_Propagate (reference: cspn_pytorch/models/cspn.py:14-83) and the
 * 3x3x3 / pre-normalised-gate call site of the Paddle demo
 * (reference: cspn_paddle/demo.py:41-43,50-52).  Plain pointers and sizes only,
 * no torch types.  The binding a reference maintainer adds is ~25 lines of
 * ctypes; cspn_amd/cspn.py is that binding, packaged.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32, contiguous NCHW / NCDHW) valid on
 *     the current HIP device; the call only ENQUEUES work on `stream` and returns
 *     (no hipDeviceSynchronize, no device allocation; safe to call from several
 *     host threads, cf. nn.DataParallel at reference cspn_pytorch/eval.py:117).
 *     State the library keeps between calls -- the complete list:
 *       (1) the persistent 3D kernel's, per device: the event that chains its
 *           launches, one pinned status word and two launch counters (see
 *           cspn3d_check_status);
 *       (2) memoisation with no effect on results: the CU count per device, and
 *           per host thread the last four 2D forward plans (cut positions of the
 *           linear plan, ~1 ms of host arithmetic per new shape).
 *     Nothing else: the library reads no environment variable and exports no test
 *     switch.  What tests and measuring tools need beyond this header (plan dumps,
 *     plan A/B, a persistent launch with a muted workgroup) is a SEPARATE library,
 *     libcspn_amd_hooks.so (csrc/cspn_test_hooks.hip), which links against this
 *     one and passes the test's choice as an argument of internal entry points;
 *   - inputs are never written; `out` must not alias an input;
 *   - return 0 on success, a negative CSPN_E_* code on argument errors, or a
 *     positive hipError_t; cspn_last_error() gives a thread-local message;
 *   - `workspace` must hold cspn{2,3}d_workspace_bytes(...) bytes, 256-B aligned,
 *     and must stay untouched until the enqueued work has finished.
 */
#ifndef CSPN_AMD_H
#define CSPN_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSPN_ABI_VERSION 5   /* 2: cspn3d_check_status, CSPN_E_ASYNC, smaller cspn3d_workspace_bytes_ex; 3: cspn3d_forward_multi_f32;
                              * 4: CSPN_NORM_PRENORM, cspn2d_normalize_f32, cspn2d_forward_prenorm_f32, cspn3d_backward_multi_f32; the
                              *    sited8 experiment's three entry points left the ABI (hook library, experiment builds); CSPN_ALGO_FUSED_PADDED
                              *    (what AUTO returns for W % 4 != 0; cspn2d_workspace_bytes grows accordingly for such widths);
                              * 5: cspn_guidance_head_f32 (the producer of the path's inputs) and cspn_guidance_head_backward_f32; CSPN_NORM_PRENORM on the 2D backward entry points */

/* hipStream_t, spelled without the HIP headers. NULL = the null stream. */
typedef void* cspn_stream_t;

/* norm_type: reference cspn_pytorch/models/cspn.py:19,36 ('8sum' | '8sum_abs');
 * NONE = gates are used as given, centre-sited, no centre term -- the contract of
 * fluid.layers.affinity_propagate (reference cspn_paddle/README.md:54: "should be
 * normalized in the channel dimension" by the caller, cspn_paddle/demo.py:47-49). */
enum { CSPN_NORM_8SUM = 0, CSPN_NORM_8SUM_ABS = 1, CSPN_NORM_NONE = 2,
       CSPN_NORM_PRENORM = 3 /* 2D only: `guidance` holds the reference's gate_wb (see cspn2d_normalize_f32 below) */ };

/* algo: AUTO picks the fused kernel whenever the shape allows it (W % 4 == 0, 16-byte aligned output).  FUSED runs images at
 * least 256 columns wide through the assembly main loop for EVERY n_iter (round 5): n_iter = 24 k + r is one short first pass
 * of r iterations (a row is stored when it completes level r) followed by k passes of 24 -- one launch per pass, 40 / 44 B
 * per pixel and pass; narrower images and FUSED_CXX (A/B tests) take the compiler-generated version of the same kernel. */
enum { CSPN_ALGO_AUTO = 0, CSPN_ALGO_STEPWISE = 1, CSPN_ALGO_FUSED = 2, CSPN_ALGO_FUSED_CXX = 3,
       CSPN_ALGO_FUSED_PADDED = 4 /* W % 4 != 0 (what AUTO picks there, round 5): the inputs are laid out once in the workspace with rows padded to a
                                    * multiple of 4 columns (zeros; 8SUM / 8SUM_ABS are normalised on the way, for the real width), the fused path
                                    * runs on those and the output is copied back: three more passes over the data instead of one launch per iteration */ };

enum {
    CSPN_E_BADARG = -1,   /* null pointer, non-positive size, unknown enum      */
    CSPN_E_WORKSPACE = -2, /* workspace too small or misaligned                  */
    CSPN_E_UNSUPPORTED = -3, /* algo explicitly requested but shape not supported */
    CSPN_E_ASYNC = -4      /* an EARLIER call failed on the device after it had returned (cspn3d_check_status) */
};

int cspn_abi_version(void);
const char* cspn_last_error(void);

/* ---- 2D: replaces Affinity_Propagate.forward, reference cspn.py:42-83 -------
 * (affinity_normalization :85-144, pad_blur_depth :147-172, sum_conv :44-53 and
 * the elementwise tail :70-81 are all inside this one call)
 *   guidance [B,8,H,W]  raw affinities as the backbone emits them (cspn.py:42)
 *   blur     [B,1,H,W]  coarse depth, also the H_0 of the centre term (cspn.py:58,76)
 *   sparse   [B,1,H,W]  or NULL; only its sign is used (cspn.py:64,81)
 *   out      [B,1,H,W]
 *   n_iter   = prop_time (cspn.py:31,66); 0 copies blur to out                  */
size_t cspn2d_workspace_bytes(int B, int H, int W, int n_iter);
int cspn2d_forward_f32(const float* guidance, const float* blur, const float* sparse, float* out,
                       int B, int H, int W, int n_iter, int norm_type,
                       void* workspace, size_t workspace_bytes, cspn_stream_t stream);
/* same, with an explicit kernel choice (tests and bench use it to A/B the paths) */
int cspn2d_forward_f32_algo(const float* guidance, const float* blur, const float* sparse, float* out,
                            int B, int H, int W, int n_iter, int norm_type, int algo,
                            void* workspace, size_t workspace_bytes, cspn_stream_t stream);
/* which kernel AUTO would run for this shape: CSPN_ALGO_FUSED, CSPN_ALGO_FUSED_PADDED (W % 4 != 0: rows padded in the workspace) or
 * CSPN_ALGO_STEPWISE (also AUTO's fallback when `out` / the workspace is not 16-byte aligned) */
int cspn2d_auto_algo(int B, int H, int W, int n_iter);

/* ---- 2D with the normalisation moved to the producer (second alternative: "fuse normalisation into that conv's
 * epilogue", the conv being gud_up_proj_layer6 at cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206,318-319,372-373).
 * The contract is the reference's own intermediate: `gate_wb`, what affinity_normalization returns (cspn.py:85-144, used at
 * :69-76) -- wb [B,8,H,W] with wb_k(p) = G_k(p) / sum_j |G_j(p)|, G_k(p) = g~_k(p + off_k), zero outside the image: normalised AND
 * consumer-sited, the same 32 B/pixel as the raw guidance.  norm_type CSPN_NORM_PRENORM on cspn2d_forward_f32 / _algo takes such a
 * tensor in the `guidance` argument: what is left of the fold is sigma = sum_k wb_k, the centre term (1 - sigma) H_0 (cspn.py:76)
 * and the mask (cspn.py:81) -- no abs-sum, no reciprocal, no edge patching, aligned loads.  Results: those of CSPN_NORM_8SUM /
 * _8SUM_ABS on the raw guidance up to the rounding of the division (<= 1e-6 relative; NaN where sum |G| = 0, as the reference).
 * Every shape and n_iter the forward takes.  Backward (ABI 5): cspn2d_backward_f32 / _history_f32 with CSPN_NORM_PRENORM return dL/d(wb) in grad_guidance --
 * dL/dwb_k(p) = (1 - m)(dW'_k - dC H_0)(p), no normalisation chain, no scatter -- and dL/d(blur_depth); chaining dL/d(wb) into the head is the producer's.
 * cspn2d_normalize_f32 is that producer epilogue as a stand-alone kernel (norm_type 8SUM or 8SUM_ABS; tests, A/B timing):
 * 36 B read + 32 B written per pixel.  cspn2d_forward_prenorm_f32 = cspn2d_forward_f32 with norm_type CSPN_NORM_PRENORM. */
int cspn2d_normalize_f32(const float* guidance, float* wb, int B, int H, int W, int norm_type, cspn_stream_t stream);
int cspn2d_forward_prenorm_f32(const float* wb, const float* blur, const float* sparse, float* out,
                               int B, int H, int W, int n_iter, void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* ---- 2D backward: the gradient torch autograd computes through Affinity_Propagate.forward (reference cspn.py:42-83),
 * i.e. what reference cspn_pytorch/train.py:196-198 back-propagates through.
 *   grad_out      [B,1,H,W]  dL/d(out)
 *   grad_guidance [B,8,H,W]  dL/d(guidance), or NULL to skip
 *   grad_blur     [B,1,H,W]  dL/d(blur_depth) (as level-0 value and as H_0 of the centre / mask terms), or NULL to skip
 * sparse_depth gets no gradient (only its sign is used, cspn.py:64).  n_iter >= 1.  norm_type: 8SUM, 8SUM_ABS, NONE, or PRENORM (then `guidance` = gate_wb and
 * grad_guidance = dL/d(gate_wb), see above).  Fast path (two sweeps of the assembly ring that keep every fourth
 * level + one recomputing final pass): W >= 256, W % 4 == 0, n_iter = 4, 8 .. 24 (24 only until round 5); everything else runs one launch per step.
 * For n_iter < 24 both sweeps still run the full 24-level ring and discard the levels beyond n_iter (the checkpoint of level n_iter is stored when the row passes it;
 * `out` briefly holds level 24 before the level-n_iter plane overwrites it): correct, but their cost does not shrink with n_iter (profiles/r05_backward_niter.jsonl),
 * and the discarded levels may hold Inf / NaN -- never read `scratch` or `out` of a call that has not completed. */
size_t cspn2d_backward_workspace_bytes(int B, int H, int W, int n_iter);
int cspn2d_backward_f32(const float* guidance, const float* blur, const float* sparse, const float* grad_out,
                        float* grad_guidance, float* grad_blur, int B, int H, int W, int n_iter, int norm_type,
                        void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* Training mode (optional, faster): the forward also keeps what the backward needs -- every fourth intermediate level (H_4, H_8 ..
 * H_20; the backward recomputes the three in between) and the folded coefficients, 13 planes of B*H*W floats, cspn2d_history_bytes() bytes, 256-B aligned; 0 = not available for this shape / n_iter (available: W >= 256, W % 4 == 0, n_iter = 4, 8 .. 24; else use
 * cspn2d_forward_f32 + cspn2d_backward_f32, which recomputes the history).  This is what torch autograd does for the
 * reference by saving ~27 temporaries per iteration.
 *   cspn2d_forward_history_f32: same result as cspn2d_forward_f32, plus `history`; workspace cspn2d_workspace_bytes().
 *   cspn2d_backward_history_f32: same result as cspn2d_backward_f32 from that history; workspace
 *   cspn2d_backward_history_workspace_bytes(). */
size_t cspn2d_history_bytes(int B, int H, int W, int n_iter);
int cspn2d_forward_history_f32(const float* guidance, const float* blur, const float* sparse, float* out, void* history,
                               size_t history_bytes, int B, int H, int W, int n_iter, int norm_type,
                               void* workspace, size_t workspace_bytes, cspn_stream_t stream);
size_t cspn2d_backward_history_workspace_bytes(int B, int H, int W, int n_iter);
int cspn2d_backward_history_f32(const float* guidance, const float* blur, const float* sparse, const float* grad_out,
                                const void* history, size_t history_bytes, float* grad_guidance, float* grad_blur,
                                int B, int H, int W, int n_iter, int norm_type,
                                void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* ---- 3D: replaces n_iter chained fluid.layers.affinity_propagate calls,
 * reference cspn_paddle/demo.py:41-43,50-52 (kernel_size == 3 only, demo.py:90)
 *   gate [B,26,D,H,W], feat [B,1,D,H,W], sparse [B,1,D,H,W] or NULL, out [B,1,D,H,W] */
size_t cspn3d_workspace_bytes(int B, int D, int H, int W, int n_iter);   /* enough for every mode */
/* what this particular call needs with 16-byte aligned tensors (norm NONE without a mask: two value volumes + the exchange
 * buffers of the persistent kernel, ~14 MB; the folding modes, and misaligned tensors: 27 planes more = cspn3d_workspace_bytes) */
size_t cspn3d_workspace_bytes_ex(int B, int D, int H, int W, int n_iter, int norm_type, int has_sparse);
int cspn3d_forward_f32(const float* gate, const float* feat, const float* sparse, float* out,
                       int B, int D, int H, int W, int n_iter, int norm_type,
                       void* workspace, size_t workspace_bytes, cspn_stream_t stream);
/* algo: AUTO keeps the 26 gates of every voxel in registers across all n_iter steps (persistent kernel, one pass over the
 * gate tensor per forward) when W % 4 == 0, the tensors are 16-byte aligned and 2 <= n_iter <= 60 -- the Paddle contract
 * (norm NONE, no mask) directly, the normalising / masked modes after one folding pass; STEPWISE = one launch and one pass
 * over the gates (or the 27 folded planes) per step.  The persistent kernel needs all of its workgroups resident at once;
 * launches of one process are chained so that two of them never share the device: one process per GPU. */
enum { CSPN_ALGO3D_AUTO = 0, CSPN_ALGO3D_STEPWISE = 1, CSPN_ALGO3D_PERSISTENT = 2 };
/* Failures that only show on the device.  The workgroups of the persistent kernel wait for each other; if some of them never get
 * a compute unit (another process, a CU mask, a long kernel of another library on the device), the waiting ones give up after
 * ~0.5 s, fill the voxels they own with NaN -- the call's `out` then never passes for a result -- and raise a sticky per-device
 * status word.  The NEXT cspn3d_* call of the process on that device (forward or backward, any stream) finds it without a
 * synchronisation, returns CSPN_E_ASYNC instead of enqueuing anything (the reporting call itself is NOT run: call again), and
 * remembers that this launch has been reported: the status word holds the NUMBER of the launch that gave up, so the same
 * launch's other workgroups, which run into their own timeouts later, do not produce a second report.  (A launch captured into a
 * HIP graph replays with frozen arguments: it raises a fixed marker instead, which is cleared when reported, so EVERY failing
 * replay is reported -- possibly twice, if more of its workgroups time out after the report.)  cspn3d_check_status
 * synchronises `stream` first, so it also reports the call just made: 0, CSPN_E_ASYNC or a hipError_t.
 * Pre-flight: the persistent kernel is only chosen when hipOccupancyMaxActiveBlocksPerMultiprocessor x the CU count says all of
 * its workgroups fit the device at once; otherwise AUTO runs the per-step kernels (and PERSISTENT returns CSPN_E_UNSUPPORTED). */
int cspn3d_check_status(cspn_stream_t stream);
int cspn3d_forward_f32_algo(const float* gate, const float* feat, const float* sparse, float* out,
                            int B, int D, int H, int W, int n_iter, int norm_type, int algo,
                            void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* C input channels on SHARED gates (reference cspn_paddle/README.md:56: "gate_weight would be shared in the channel dimension for
 * input when C>1"; call site demo.py:41-43): feat, out [B,C,D,H,W], gate [B,26,D,H,W] used as given (norm NONE, no mask), n_iter
 * chained steps.  The gates of a chunk are read ONCE and stay in the registers while the steps run for channel after channel
 * (a per-channel loop over cspn3d_forward_f32 reads the 104 B/voxel of gates C times).  cspn3d_multi_supported() != 0 where the
 * persistent kernel takes the call (W % 4 == 0, 2 <= n_iter <= 60, 16-byte aligned tensors, the volume fits the device);
 * elsewhere the entry point returns CSPN_E_UNSUPPORTED and the caller loops over the channels.  Workspace:
 * cspn3d_workspace_bytes_ex(B, D, H, W, n_iter, CSPN_NORM_NONE, 0).  Same failure reporting as above (CSPN_E_ASYNC). */
int cspn3d_multi_supported(int B, int C, int D, int H, int W, int n_iter);
int cspn3d_forward_multi_f32(const float* gate, const float* feat, float* out, int B, int C, int D, int H, int W, int n_iter,
                             void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* Backward of the 3D op under the Paddle contract (norm_type CSPN_NORM_NONE, no mask): the reference op is differentiated by
 * the demo's optimiser (cspn_paddle/demo.py:65-75, `feat` has stop_gradient=False).  grad_out [B,1,D,H,W];
 * grad_gate [B,26,D,H,W] and grad_feat [B,1,D,H,W] are outputs, either may be NULL.  n_iter chained steps with the same
 * gates are differentiated as one op (n_iter = 1 is the single fluid.layers.affinity_propagate call). */
size_t cspn3d_backward_workspace_bytes(int B, int D, int H, int W, int n_iter);
int cspn3d_backward_f32(const float* gate, const float* feat, const float* grad_out, float* grad_gate, float* grad_feat,
                        int B, int D, int H, int W, int n_iter, int norm_type,
                        void* workspace, size_t workspace_bytes, cspn_stream_t stream);
/* The same for C input channels on SHARED gates (reference cspn_paddle/README.md:56; the demo's optimiser differentiates the op,
 * demo.py:65-75): feat, grad_out, grad_feat [B,C,D,H,W]; grad_gate [B,26,D,H,W] = the gate gradient SUMMED over the channels (what
 * autograd accumulates into a shared tensor).  Any shape and n_iter >= 1; with n_iter >= 3 where cspn3d_multi_supported() holds, the
 * level-keeping forward and the transposed sweep are ONE persistent launch each for all channels (gates resident in the registers
 * across the channels) and the gate planes are written once.  Outputs may be NULL.  C = 1 is cspn3d_backward_f32. */
size_t cspn3d_backward_multi_workspace_bytes(int B, int C, int D, int H, int W, int n_iter);
int cspn3d_backward_multi_f32(const float* gate, const float* feat, const float* grad_out, float* grad_gate, float* grad_feat,
                              int B, int C, int D, int H, int W, int n_iter,
                              void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* ---- the steps right next to the path, on the device ----
 * cspn_metrics_f32: reference cspn_pytorch/utils.py:19-47 (evaluate_error) and loss.py:16-23 (Wighted_L1_Loss = MAE
 * over gt > 1e-4) as one fused masked reduction over n elements.  out12 (device, 12 floats):
 *   [0] n_valid  [1] MSE  [2] RMSE  [3] ABS_REL  [4] LG10 (the reference never fills it: 0)  [5] MAE
 *   [6..11] DELTA1.02, 1.05, 1.10, 1.25, 1.25^2, 1.25^3.   All zero when no element is valid (utils.py:23-26).
 * cspn_l1_backward_f32: d(Wighted_L1_Loss)/d(pred) = grad_scale[0] * sign(pred - label) / n_valid on label > 1e-4;
 *   stats12 = the out12 of cspn_metrics_f32(label, pred), grad_scale = 1 device float.
 * cspn_unpool_f32: reference models/torch_resnet_cspn_nyu.py:41-54 (Unpool: conv_transpose2d with a one-hot
 *   stride x stride kernel): out[nc][y*stride][x*stride] = x[nc][y][x], zeros elsewhere; x [NC,H,W] -> out [NC,H*s,W*s]. */
size_t cspn_metrics_workspace_bytes(size_t n);
int cspn_metrics_f32(const float* gt, const float* pred, size_t n, float* out12, void* workspace, size_t workspace_bytes,
                     cspn_stream_t stream);
int cspn_l1_backward_f32(const float* pred, const float* label, const float* stats12, const float* grad_scale,
                         float* grad_pred, size_t n, cspn_stream_t stream);
int cspn_unpool_f32(const float* x, float* out, size_t NC, int H, int W, int stride, cspn_stream_t stream);
int cspn_unpool_backward_f32(const float* grad_out, float* grad_x, size_t NC, int H, int W, int stride, cspn_stream_t stream);

/* cspn_sparse_sample_f32: reference createSparseDepthImage on the device -- sparse = depth * bernoulli(p), p = n_sample /
 * (pixels per image) for mode 0 (cspn_pytorch/nyu_dataset_loader.py:135-144) or n_sample / (pixels of that image with
 * depth > 1e-4) for mode 1 (cspn_pytorch/kitti_dataset_loader.py:138-148).  depth, sparse_out: [n_images][hw] floats.
 * Counter-based generator keyed by (seed, image, pixel): reproducible, independent of launch geometry; NOT the bit stream
 * of torch.bernoulli.  workspace: cspn_sparse_sample_workspace_bytes(n_images) (only used by mode 1). */
size_t cspn_sparse_sample_workspace_bytes(size_t n_images);
int cspn_sparse_sample_f32(const float* depth, float* sparse_out, size_t n_images, size_t hw, int n_sample, int mode,
                           unsigned long long seed, void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* ---- the producer of the path's inputs (producer half): the two heads Simple_Gudi_UpConv_Block_Last_Layer of the reference backbone
 * (cspn_pytorch/models/torch_resnet_cspn_nyu.py:187-206; gud_up_proj_layer6 64 -> 8 = guidance and gud_up_proj_layer5 64 -> 1 = blur depth, :318-319,
 * called :372-373) as ONE kernel: Unpool (:41-54, narrowed to H x W :196-201) + 3x3 conv, padding 1, no bias (:190) -- the three quarters of the
 * unpooled taps that are structurally zero are never multiplied.
 *   x [B,C,h,w];  w_guidance [8,C,3,3];  w_blur [1,C,3,3] or NULL;  guidance_out [B,8,H,W];  blur_out [B,1,H,W] or NULL;  H <= 2h, W <= 2w (the reference: exactly 2x).
 *   norm_type CSPN_NORM_NONE: guidance_out = the raw guidance, what gud_up_proj_layer6 returns (feed it to cspn2d_forward_f32 with '8sum' / '8sum_abs');
 *   CSPN_NORM_8SUM / CSPN_NORM_8SUM_ABS: guidance_out = gate_wb = affinity_normalization (cspn.py:85-144) of that guidance, fused behind the conv (IEEE
 *   division: 0 / 0 = NaN as in the reference) -- the input contract of cspn2d_forward_f32 with CSPN_NORM_PRENORM; no stand-alone normalisation pass.
 * workspace: cspn_guidance_head_workspace_bytes(C) (the packed weights). */
size_t cspn_guidance_head_workspace_bytes(int C);
int cspn_guidance_head_f32(const float* x, const float* w_guidance, const float* w_blur, float* guidance_out, float* blur_out,
                           int B, int C, int h, int w, int H, int W, int norm_type,
                           void* workspace, size_t workspace_bytes, cspn_stream_t stream);

/* The gradient of the raw heads (norm_type NONE above): what torch autograd computes through the two reference layers when the training loop back-propagates
 * through torch_resnet_cspn_nyu.py:372-373.  grad_guidance [B,8,H,W] and grad_blur [B,1,H,W] (with w_blur; both or neither) are dL/d(outputs);
 *   grad_x          [B,C,h,w]   dL/dx = sum_{o,ky,kx} W[o][c][ky][kx] g[o][2i + 1 - ky][2j + 1 - kx]   (inputs beyond the narrowed output: 0), or NULL to skip
 *   grad_w_guidance [8,C,3,3], grad_w_blur [1,C,3,3]   dL/dW = sum over the batch and all input pixels of x g -- on the matrix cores (fp32 MFMA), the waves'
 *                   partial sums added in a fixed order: deterministic; either or both may be NULL
 * workspace: cspn_guidance_head_backward_workspace_bytes(B, C, h, w) bytes, 256-byte aligned. */
size_t cspn_guidance_head_backward_workspace_bytes(int B, int C, int h, int w);
int cspn_guidance_head_backward_f32(const float* x, const float* w_guidance, const float* w_blur, const float* grad_guidance, const float* grad_blur,
                                    float* grad_x, float* grad_w_guidance, float* grad_w_blur, int B, int C, int h, int w, int H, int W,
                                    void* workspace, size_t workspace_bytes, cspn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSPN_AMD_H */
