/*
 * oracle/cspn_oracle.c -- CPU restatement of the CSPN propagation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * What it restates (all paths relative to /root/reference):
 *   cspn_pytorch/models/cspn.py:42-83    Affinity_Propagate.forward
 *   cspn_pytorch/models/cspn.py:85-144   affinity_normalization
 *   cspn_pytorch/models/cspn.py:147-172  pad_blur_depth
 *   cspn_pytorch/models/cspn.py:44-53    sum_conv (Conv3d 8->1, 1x1x1, weight ones)
 *   cspn_paddle/demo.py:24,41-52         3D / "pre-normalised gate" call site
 *
 * Parity status
 *   2D ('8sum', '8sum_abs'): PINNED.  the tests/golden/ .npz files were produced by the
 *     unmodified reference module (tests/golden/make_golden.py, run in the
 *     authoring container) and tests/test_oracle.py checks this file against
 *     them.
 *     'prenorm' (norm_type 3) and cspn2d_oracle_gate_wb_f32: PINNED as well --
 *     tests/golden/cspn2d_norm_golden.npz holds gate_wb / gate_sum as the unmodified
 *     reference's affinity_normalization returned them (tests/golden/make_norm_golden.py);
 *     the export reproduces them and the 'prenorm' mode, fed the GOLDEN gate_wb,
 *     reproduces the golden outputs (tests/test_oracle.py).
 *   3D and norm_type==2 (Paddle style): PARITY UNPINNED.  The arithmetic of
 *     fluid.layers.affinity_propagate lives in a custom PaddlePaddle 1.5.2
 *     wheel that is not in the reference tree (cspn_paddle/README.md:24,30-35)
 *     and the reference holds no test vector for it.  The 3D code below is the
 *     direct 3x3x3 generalisation of the pinned 2D semantics.
 *
 * The code keeps the reference's structure on purpose: one zero-padded canvas
 * per affinity channel (cspn.py:105-132), an 8->1 channel sum standing in for
 * sum_conv, a padded depth canvas rebuilt every iteration (cspn.py:69) and the
 * elementwise tail of cspn.py:76,81 evaluated in the reference's operation
 * order.  The one liberty: pad_blur_depth's eight shifted copies of the SAME
 * depth plane are read as eight shifted windows of one padded plane, which is
 * the identical set of values.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ZeroPad2d((left,right,top,bottom)) tuples of cspn.py:105-128 (and :149-167),
 * reduced to (top,left): padded[y][x] = plane[y-top][x-left].               */
static const int PAD_T[8] = {0, 0, 0, 1, 1, 2, 2, 2};
static const int PAD_L[8] = {0, 1, 2, 0, 2, 0, 1, 2};

int cspn_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void cspn_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* norm_type: 0 = '8sum', 1 = '8sum_abs' (cspn.py:36), 2 = gates used as given,
 * centre-sited, no centre term (Paddle contract, cspn_paddle/README.md:54),
 * 3 = 'prenorm': g IS the reference's gate_wb (what affinity_normalization,
 * cspn.py:85-144, returns, cropped to the image): the loop of cspn.py:66-81 runs
 * on it as it stands, gate_sum = its channel sum (cspn.py:139).
 * wb_out != NULL: ALSO store the cropped gate_wb [8][H][W] (pinned by
 * tests/golden/cspn2d_norm_golden.npz, vectors of the unmodified reference).  */
static int cspn2d_one(const float* g, const float* blur, const float* sparse, float* out,
                      int H, int W, int n_iter, int norm_type, float* wb_out) {
    const int PH = H + 2, PW = W + 2;
    const size_t pn = (size_t)PH * PW, n = (size_t)H * W;
    float* gate_wb = (float*)calloc(8 * pn, sizeof(float)); /* [8][H+2][W+2] */
    float* gate_sum = (float*)malloc(n * sizeof(float));    /* [H][W]        */
    float* pad = (float*)calloc(pn, sizeof(float));          /* padded depth  */
    float* res = (float*)malloc(n * sizeof(float));
    if (!gate_wb || !gate_sum || !pad || !res) {
        free(gate_wb); free(gate_sum); free(pad); free(res);
        return -1;
    }

    if (norm_type == 2 || norm_type == 3) {
        /* centre-sited: weight of neighbour k of pixel p is gate[k][p] itself.
         * Stored in the same "value used at padded coord (i+1,j+1)" layout.  */
        for (int k = 0; k < 8; ++k)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j)
                    gate_wb[k * pn + (size_t)(i + 1) * PW + (j + 1)] = g[k * n + (size_t)i * W + j];
        for (size_t i = 0; i < n; ++i) {
            float gs = 0.0f;                               /* cspn.py:139 on the given gate_wb */
            for (int k = 0; k < 8; ++k) gs += g[k * n + i];
            gate_sum[i] = (norm_type == 2) ? 1.0f : gs;    /* 2: no centre term */
        }
    } else {
        /* cspn.py:88-89 abs; :105-132 eight differently padded canvases      */
        for (int k = 0; k < 8; ++k) {
            float* dst = gate_wb + k * pn;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    float v = g[k * n + (size_t)i * W + j];
                    if (norm_type == 1) v = fabsf(v);
                    dst[(size_t)(i + PAD_T[k]) * PW + (j + PAD_L[k])] = v;
                }
        }
        /* cspn.py:135-138: abs_weight = sum_conv(|gate_wb|); gate_wb /= abs_weight.
         * Only the cropped interior [1:-1,1:-1] is ever used (:72,:142).     */
        for (int i = 1; i <= H; ++i)
            for (int j = 1; j <= W; ++j) {
                const size_t q = (size_t)i * PW + j;
                float s = 0.0f;
                for (int k = 0; k < 8; ++k) s += fabsf(gate_wb[k * pn + q]);
                float gs = 0.0f;
                for (int k = 0; k < 8; ++k) {
                    const float w = gate_wb[k * pn + q] / s; /* IEEE: 0/0 -> NaN */
                    gate_wb[k * pn + q] = w;
                    gs += w;                                  /* cspn.py:139    */
                }
                gate_sum[(size_t)(i - 1) * W + (j - 1)] = gs;
            }
    }

    if (wb_out)
        for (int k = 0; k < 8; ++k)
            for (int i = 0; i < H; ++i)
                memcpy(wb_out + k * n + (size_t)i * W, gate_wb + k * pn + (size_t)(i + 1) * PW + 1, (size_t)W * sizeof(float));
    memcpy(res, blur, n * sizeof(float)); /* cspn.py:58,61 */
    for (int it = 0; it < n_iter; ++it) {
        /* cspn.py:69 pad_blur_depth: plane k is the depth padded by tuple k, so
         * padded_k[i+1][j+1] = depth[i+1-t_k][j+1-l_k] = P[i+2-t_k][j+2-l_k]
         * with P the depth padded by one zero on every side.                  */
        for (int i = 0; i < H; ++i)
            memcpy(pad + (size_t)(i + 1) * PW + 1, res + (size_t)i * W, (size_t)W * sizeof(float));
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const size_t q = (size_t)(i + 1) * PW + (j + 1);
                float acc = 0.0f; /* cspn.py:70 sum_conv(gate_wb * result_depth) */
                for (int k = 0; k < 8; ++k)
                    acc += gate_wb[k * pn + q] * pad[(size_t)(i + 2 - PAD_T[k]) * PW + (j + 2 - PAD_L[k])];
                const size_t p = (size_t)i * W + j;
                float r = acc;
                if (norm_type != 2) r = (1.0f - gate_sum[p]) * blur[p] + acc; /* cspn.py:76 */
                if (sparse) {                                                  /* cspn.py:64,81 */
                    const float s = sparse[p];
                    const float m = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : s); /* sign(); keeps NaN */
                    r = (1.0f - m) * r + m * blur[p];
                }
                res[p] = r;
            }
    }
    memcpy(out, res, n * sizeof(float));
    free(gate_wb); free(gate_sum); free(pad); free(res);
    return 0;
}

/* guidance [B,8,H,W], blur [B,1,H,W], sparse [B,1,H,W] or NULL, out [B,1,H,W] */
int cspn2d_oracle_f32(const float* guidance, const float* blur, const float* sparse, float* out,
                      int B, int H, int W, int n_iter, int norm_type) {
    if (B < 0 || H <= 0 || W <= 0 || n_iter < 0 || norm_type < 0 || norm_type > 3) return -2;
    const size_t n = (size_t)H * W;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        int e = cspn2d_one(guidance + (size_t)b * 8 * n, blur + (size_t)b * n,
                           sparse ? sparse + (size_t)b * n : NULL, out + (size_t)b * n,
                           H, W, n_iter, norm_type, NULL);
        if (e) {
#pragma omp atomic write
            err = e;
        }
    }
    return err;
}

/* gate_wb of reference affinity_normalization (cspn.py:85-144), cropped: guidance [B,8,H,W] -> wb [B,8,H,W] */
int cspn2d_oracle_gate_wb_f32(const float* guidance, float* wb, int B, int H, int W, int norm_type) {
    if (B < 0 || H <= 0 || W <= 0 || norm_type < 0 || norm_type > 1) return -2;
    const size_t n = (size_t)H * W;
    int err = 0;
    float* blur = (float*)calloc(n, sizeof(float));
    float* out = (float*)malloc(n * sizeof(float));
    if (!blur || !out) { free(blur); free(out); return -1; }
    for (int b = 0; b < B; ++b)
        if (cspn2d_one(guidance + (size_t)b * 8 * n, blur, NULL, out, H, W, 0, norm_type, wb + (size_t)b * 8 * n)) err = -1;
    free(blur); free(out);
    return err;
}

/* ---- 3D: 26 neighbours, PARITY UNPINNED (see header) --------------------- */
/* Channel order: raster over (f,t,l) in {0,1,2}^3 skipping (1,1,1); the canvas
 * of channel c is the plane padded by (front=f, top=t, left=l), i.e. neighbour
 * offset (1-f, 1-t, 1-l) -- the same rule that turns cspn.py:105-128's tuples
 * into the eight 2D offsets.                                                  */
/* par_inner != 0: the voxel loops of THIS volume run on all threads (z-planes in parallel; every voxel's arithmetic and
 * its order are unchanged, so the result is bit-identical) -- used when there are fewer volumes than threads, so that
 * one full-size volume (32x160x608) finishes in about a second. */
static int cspn3d_one(const float* g, const float* feat, const float* sparse, float* out,
                      int D, int H, int W, int n_iter, int norm_type, int par_inner) {
    const int PD = D + 2, PH = H + 2, PW = W + 2;
    const size_t pn = (size_t)PD * PH * PW, n = (size_t)D * H * W;
    int pf[26], pt[26], pl[26], c = 0;
    for (int f = 0; f < 3; ++f)
        for (int t = 0; t < 3; ++t)
            for (int l = 0; l < 3; ++l) {
                if (f == 1 && t == 1 && l == 1) continue;
                pf[c] = f; pt[c] = t; pl[c] = l; ++c;
            }
    float* w = (float*)malloc(26 * n * sizeof(float)); /* normalised weights at p */
    float* gs = (float*)malloc(n * sizeof(float));
    float* pad = (float*)calloc(pn, sizeof(float));
    float* res = (float*)malloc(n * sizeof(float));
    if (!w || !gs || !pad || !res) { free(w); free(gs); free(pad); free(res); return -1; }

#pragma omp parallel for collapse(2) schedule(static) if (par_inner)
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = ((size_t)z * H + y) * W + x;
                float G[26], s = 0.0f;
                for (int k = 0; k < 26; ++k) {
                    float v;
                    if (norm_type == 2) {
                        v = g[k * n + p];
                    } else {
                        const int zz = z + 1 - pf[k], yy = y + 1 - pt[k], xx = x + 1 - pl[k];
                        v = 0.0f;
                        if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            v = g[k * n + ((size_t)zz * H + yy) * W + xx];
                            if (norm_type == 1) v = fabsf(v);
                        }
                    }
                    G[k] = v;
                    s += fabsf(v);
                }
                float sum = 0.0f;
                for (int k = 0; k < 26; ++k) {
                    const float wk = (norm_type == 2) ? G[k] : G[k] / s;
                    w[k * n + p] = wk;
                    sum += wk;
                }
                gs[p] = (norm_type == 2) ? 1.0f : sum;
            }

    memcpy(res, feat, n * sizeof(float));
    for (int it = 0; it < n_iter; ++it) {
        for (int z = 0; z < D; ++z)
            for (int y = 0; y < H; ++y)
                memcpy(pad + ((size_t)(z + 1) * PH + (y + 1)) * PW + 1,
                       res + ((size_t)z * H + y) * W, (size_t)W * sizeof(float));
#pragma omp parallel for collapse(2) schedule(static) if (par_inner)
        for (int z = 0; z < D; ++z)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const size_t p = ((size_t)z * H + y) * W + x;
                    float acc = 0.0f;
                    for (int k = 0; k < 26; ++k)
                        acc += w[k * n + p] *
                               pad[((size_t)(z + 2 - pf[k]) * PH + (y + 2 - pt[k])) * PW + (x + 2 - pl[k])];
                    float r = acc;
                    if (norm_type != 2) r = (1.0f - gs[p]) * feat[p] + acc;
                    if (sparse) {
                        const float sv = sparse[p];
                        const float m = (sv > 0.0f) ? 1.0f : ((sv < 0.0f) ? -1.0f : sv);
                        r = (1.0f - m) * r + m * feat[p];
                    }
                    res[p] = r;
                }
    }
    memcpy(out, res, n * sizeof(float));
    free(w); free(gs); free(pad); free(res);
    return 0;
}

/* gate [B,26,D,H,W], feat [B,1,D,H,W], sparse [B,1,D,H,W] or NULL, out [B,1,D,H,W] */
int cspn3d_oracle_f32(const float* gate, const float* feat, const float* sparse, float* out,
                      int B, int D, int H, int W, int n_iter, int norm_type) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0 || n_iter < 0 || norm_type < 0 || norm_type > 2) return -2;
    const size_t n = (size_t)D * H * W;
    int err = 0;
    /* few volumes: one after the other, each on all threads; many: one thread per volume */
    const int inner = 2 * B <= omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) if (!inner)
    for (int b = 0; b < B; ++b) {
        int e = cspn3d_one(gate + (size_t)b * 26 * n, feat + (size_t)b * n,
                           sparse ? sparse + (size_t)b * n : NULL, out + (size_t)b * n,
                           D, H, W, n_iter, norm_type, inner);
        if (e) {
#pragma omp atomic write
            err = e;
        }
    }
    return err;
}
