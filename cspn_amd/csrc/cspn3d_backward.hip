// cspn3d_backward.hip -- backward of the 3x3x3 propagation under the Paddle contract (gates used as given, centre-sited,
// no centre term, no mask: reference cspn_paddle/README.md:54-56; the op is differentiated by the demo's optimiser,
// cspn_paddle/demo.py:65-75, `feat` has stop_gradient=False).  Kernel source of the reference op is NOT in the tree
// (parity unpinned): the arithmetic is the adjoint of oracle/cspn_oracle.c's 3D forward, checked against
// autograd of the same recurrence in tests/.
//
//   forward   H_{t+1}(p) = sum_k g_k(p) H_t(p + off_k),  t = 0 .. n-1,  H_0 = feat,  out = H_n      (zero outside the volume)
//   adjoint   A_n = dL/dout,   A_t(q) = sum_k g_k(q - off_k) A_{t+1}(q - off_k)                       dL/dfeat = A_0
//   gates     dL/dg_k(p) = sum_t A_{t+1}(p) H_t(p + off_k)
//
// Three kinds of launches, each HBM-bound streaming: the forward steps that keep H_1 .. H_{n-1} (step3d_direct_kernel of
// cspn3d_stepwise.hip, only when the gate gradient is wanted), n adjoint steps (26 gate planes + A in, A out = 112 B/voxel
// each, like a forward step), and ONE gate-gradient pass that reads the 2n value volumes and writes the 26 planes once
// (26 x 4 accumulators per thread).  A single chained call (n = 1, how the Paddle graph uses the op) therefore moves
// 28 + 28 planes: the adjoint step and the gate-gradient pass, nothing else.
#include <cstdlib>

#include "cspn_common.h"

namespace cspn {

namespace {

__host__ __device__ constexpr int ch3(int k) { return k < 13 ? k : k + 1; }  // skip the centre (index 13)

__device__ __forceinline__ float4 ld4u(const float* p) {   // 16 bytes, 4-byte aligned
    float4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// voxel index -> (b, z, y, x) of a thread's VEC consecutive voxels
template <int VEC>
struct Pos {
    int b, z, y, x;
    size_t r;   // offset inside the volume
    bool ok;
    __device__ Pos(int B, int D, int H, int W) {
        const size_t HW = (size_t)H * W, V = (size_t)D * HW, n = (size_t)B * V / VEC;
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        ok = i < n;
        const size_t idx = (ok ? i : 0) * VEC;
        b = (int)(idx / V);
        r = idx - (size_t)b * V;
        z = (int)(r / HW);
        const int r2 = (int)(r - (size_t)z * HW);
        y = r2 / W;
        x = r2 - y * W;
    }
};

// one adjoint step: aout(q) = sum_k g_k(q - off_k) ain(q - off_k); VEC = 4 (W % 4 == 0, 16-byte aligned tensors) or 1
// fbs: floats from one volume of ain / aout to the next (V, or C V when the call's C value channels share the gates: the caller
// passes the channel's first volume)
template <int VEC>
__global__ __launch_bounds__(256) void adjoint3d_kernel(const float* __restrict__ g, const float* __restrict__ ain,
                                                         float* __restrict__ aout, int B, int D, int H, int W, size_t fbs) {
    const Pos<VEC> p(B, D, H, W);
    if (!p.ok) return;
    const size_t HW = (size_t)H * W, V = (size_t)D * HW;
    const float* ab = ain + (size_t)p.b * fbs;
    const float* gb = g + (size_t)p.b * 26 * V;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) {   // source rows (z - dz, y - dy)
        const int dz = 1 - n / 3, dy = 1 - n % 3;
        const int zz = p.z - dz, yy = p.y - dy;
        if (zz < 0 || zz >= D || yy < 0 || yy >= H) continue;
        const size_t ro = ((size_t)zz * H + yy) * W;
        float a[VEC + 2];   // ain at columns x-1 .. x+VEC of the source row
        if (VEC == 4) {
            const float4 c = *reinterpret_cast<const float4*>(ab + ro + p.x);
            a[1] = c.x; a[2] = c.y; a[3] = c.z; a[4] = c.w;
        } else {
            a[1] = ab[ro + p.x];
        }
        a[0] = p.x > 0 ? ab[ro + p.x - 1] : 0.f;
        a[VEC + 1] = p.x + VEC < W ? ab[ro + p.x + VEC] : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {   // dx = 1 - t: the source voxel of column x + i is x + i - dx
            const int c27 = n * 3 + t;
            if (c27 == 13) continue;
            const int k = c27 < 13 ? c27 : c27 - 1;
            const int dx = 1 - t;
            const float* gp = gb + (size_t)k * V + ro + p.x - dx;
            float w[VEC];
            if (VEC == 4 && p.x - dx >= 0 && p.x - dx + 3 < W) {
                const float4 q = ld4u(gp);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int xs = p.x + i - dx;
                    w[i] = (xs >= 0 && xs < W) ? gp[i] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[i], a[1 + i - dx], acc[i]);
        }
    }
    float* o = aout + (size_t)p.b * fbs + p.r;
    if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else o[0] = acc[0];
}

// dL/dg_k(p) = sum_c sum_t A^c_{t+1}(p) H^c_t(p + off_k) over the C value channels that share the gates (reference
// cspn_paddle/README.md:56; C = 1: the plain op).  Every value tensor is [B][C][V]; level t of channel c: H_0 = feat,
// H_t = hist + (t-1) total; A_n = gout, A_t = ahist + (t-1) total (t = 1 .. n-1), total = B C V
template <int VEC>
__global__ __launch_bounds__(256) void gate_grad3d_kernel(const float* __restrict__ feat, const float* __restrict__ hist,
                                                           const float* __restrict__ ahist, const float* __restrict__ gout,
                                                           float* __restrict__ gg, int B, int D, int H, int W, int n_iter, int C) {
    const Pos<VEC> p(B, D, H, W);
    if (!p.ok) return;
    const size_t HW = (size_t)H * W, V = (size_t)D * HW, total = (size_t)B * C * V;
    float acc[26][VEC];
#pragma unroll
    for (int k = 0; k < 26; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;
#pragma unroll 1
    for (int ct = 0; ct < C * n_iter; ++ct) {
        const int cch = ct / n_iter, t = ct - cch * n_iter;
        const size_t vol = ((size_t)p.b * C + cch) * V, vox = vol + p.r;
        const float* ht = (t == 0 ? feat : hist + (size_t)(t - 1) * total) + vol;
        const float* at = t == n_iter - 1 ? gout : ahist + (size_t)t * total;   // A_{t+1}
        float a[VEC];
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(at + vox);
            a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w;
        } else {
            a[0] = at[vox];
        }
#pragma unroll
        for (int n = 0; n < 9; ++n) {   // neighbour rows (z + dz, y + dy)
            const int dz = 1 - n / 3, dy = 1 - n % 3;
            const int zz = p.z + dz, yy = p.y + dy;
            float h[VEC + 2];
#pragma unroll
            for (int i = 0; i < VEC + 2; ++i) h[i] = 0.f;
            if (zz >= 0 && zz < D && yy >= 0 && yy < H) {
                const float* row = ht + ((size_t)zz * H + yy) * W;
                if (VEC == 4) {
                    const float4 c = *reinterpret_cast<const float4*>(row + p.x);
                    h[1] = c.x; h[2] = c.y; h[3] = c.z; h[4] = c.w;
                } else {
                    h[1] = row[p.x];
                }
                if (p.x > 0) h[0] = row[p.x - 1];
                if (p.x + VEC < W) h[VEC + 1] = row[p.x + VEC];
            }
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int c27 = n * 3 + t3;
                if (c27 == 13) continue;
                const int k = c27 < 13 ? c27 : c27 - 1;
                const int dx = 1 - t3;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[k][i] = fmaf(a[i], h[1 + i + dx], acc[k][i]);
            }
        }
    }
    float* o = gg + (size_t)p.b * 26 * V + p.r;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        if (VEC == 4) *reinterpret_cast<float4*>(o + (size_t)k * V) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        else o[(size_t)k * V] = acc[k][0];
    }
}

// forward step for shapes / alignments the 16-byte kernel of cspn3d_stepwise.hip does not take
__global__ __launch_bounds__(256) void step3d_scalar_kernel(const float* __restrict__ g, const float* __restrict__ hin,
                                                             float* __restrict__ hout, int B, int D, int H, int W, size_t fbs) {
    const Pos<1> p(B, D, H, W);
    if (!p.ok) return;
    const size_t HW = (size_t)H * W, V = (size_t)D * HW;
    const float* hb = hin + (size_t)p.b * fbs;
    const float* gb = g + (size_t)p.b * 26 * V + p.r;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        const int c = ch3(k);
        const int zz = p.z + 1 - c / 9, yy = p.y + 1 - (c / 3) % 3, xx = p.x + 1 - c % 3;
        if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
            acc = fmaf(gb[(size_t)k * V], hb[((size_t)zz * H + yy) * W + xx], acc);
    }
    hout[(size_t)p.b * fbs + p.r] = acc;
}

}  // namespace

// levels kept: H_1 .. H_{n-1} and A_1 .. A_{n-1} (A_0 goes to grad_feat, or to one more volume when the caller does not want
// it), each level B C volumes laid out like feat ([B][C][V]); then the workspace of the persistent kernel (fused sweeps, n >= 3)
static size_t levels_bytes(int B, int D, int H, int W, int n_iter, int C) {
    const size_t total = (size_t)B * C * D * H * W;
    const size_t b = (2 * (size_t)(n_iter > 0 ? n_iter - 1 : 0) + 1) * total * sizeof(float);
    return (b + 255) & ~(size_t)255;
}

size_t backward3d_workspace(int B, int D, int H, int W, int n_iter, int C) {
    return levels_bytes(B, D, H, W, n_iter, C) + persistent3d_workspace(B, D, H, W);
}

int step3d_direct(const float* g, const float* hin, float* hout, int B, int D, int H, int W, hipStream_t st);   // cspn3d_stepwise.hip

// C > 1 (round 5; reference cspn_paddle/README.md:56, trained through at demo.py:65-75): feat / gout / gf hold C value channels per
// volume on shared gates, gg is the gate gradient summed over the channels.  Fused sweeps: ONE level-keeping forward launch and ONE
// transposed launch of the persistent kernel for all channels (its MULTI instantiations: the gates of a chunk stay in the registers
// while the steps run for channel after channel), then one gate-gradient pass that loops over the channels and writes the 26
// planes once.  Otherwise one launch per step and channel.
int backward3d(const float* g, const float* feat, const float* gout, float* gg, float* gf, int B, int D, int H, int W,
               int n_iter, void* ws, hipStream_t st, bool stepwise_only, int C) {
    const size_t V = (size_t)D * H * W, total = (size_t)B * C * V, fbs = (size_t)C * V;
    float* hist = (float*)ws;                                  // H_1 .. H_{n-1}
    float* ahist = hist + (size_t)(n_iter - 1) * total;        // A_1 .. A_{n-1}
    float* a0 = gf ? gf : ahist + (size_t)(n_iter - 1) * total;
    const bool vec = (W % 4) == 0 &&
                     ((((uintptr_t)g | (uintptr_t)feat | (uintptr_t)gout | (uintptr_t)gg | (uintptr_t)gf | (uintptr_t)ws) & 15u) == 0);
    const unsigned blocks = (unsigned)(((size_t)B * V / (vec ? 4 : 1) + 255) / 256);   // threads cover ONE channel's B volumes
    // fused sweeps: the persistent kernel (gates read once per sweep, resident in registers across the steps) in its
    // level-keeping and transposed variants -- forward H_1 .. H_{n-1} (n - 1 steps, the last one "out" = H_{n-1}), adjoint
    // A_{n-1} .. A_0 (n steps).  One launch per step otherwise.
    void* pws = (char*)ws + levels_bytes(B, D, H, W, n_iter, C);
    const bool fused = vec && !stepwise_only && n_iter >= 3 &&
                       (C == 1 ? persistent3d_supported(B, D, H, W, n_iter - 1) && persistent3d_supported(B, D, H, W, n_iter)
                               : persistent3d_multi_supported(B, C, D, H, W, n_iter - 1) && persistent3d_multi_supported(B, C, D, H, W, n_iter));
    if (fused) {
        if (gg)
            if (int e = persistent3d_run(g, feat, hist + (size_t)(n_iter - 2) * total, hist, -1, 1, false, B, D, H, W, n_iter - 1, pws, st, P3Options(), C))
                return e;
        // step it of the adjoint run produces A_{n-it}: volume n - it - 1 of ahist; the last one (A_0) is its "out"
        if (gf || gg)
            if (int e = persistent3d_run(g, gout, a0, ahist, n_iter - 1, -1, true, B, D, H, W, n_iter, pws, st, P3Options(), C)) return e;
        if (gg) {
            hipLaunchKernelGGL(gate_grad3d_kernel<4>, dim3(blocks), dim3(256), 0, st, feat, hist, ahist, gout, gg, B, D, H, W, n_iter, C);
            if (int e = check_launch("gate_grad3d_kernel")) return e;
        }
        return 0;
    }
    for (int c = 0; c < C; ++c) {
        const size_t co = (size_t)c * V;
        if (gg) {   // the value levels the gate gradient multiplies with
            const float* src = feat + co;
            for (int t = 1; t < n_iter; ++t) {
                float* dst = hist + (size_t)(t - 1) * total + co;
                if (vec && C == 1) {
                    if (int e = step3d_direct(g, src, dst, B, D, H, W, st)) return e;
                } else {
                    hipLaunchKernelGGL(step3d_scalar_kernel, dim3((unsigned)(((size_t)B * V + 255) / 256)), dim3(256), 0, st, g, src, dst, B, D, H, W, fbs);
                }
                src = dst;
            }
            if (int e = check_launch("3D forward levels")) return e;
        }
        // adjoint levels A_{n-1} .. A_1 (kept only if the gate gradient needs them: otherwise two volumes would do, but the
        // workspace is sized for the general call) and A_0
        const float* src = gout + co;
        for (int t = n_iter - 1; t >= 0; --t) {
            float* dst = (t == 0 ? a0 : ahist + (size_t)(t - 1) * total) + co;
            if (t == 0 && !gf) break;   // A_0 is only the feature gradient
            if (vec) hipLaunchKernelGGL(adjoint3d_kernel<4>, dim3(blocks), dim3(256), 0, st, g, src, dst, B, D, H, W, fbs);
            else hipLaunchKernelGGL(adjoint3d_kernel<1>, dim3(blocks), dim3(256), 0, st, g, src, dst, B, D, H, W, fbs);
            src = dst;
        }
        if (int e = check_launch("adjoint3d_kernel")) return e;
    }
    if (gg) {
        if (vec)
            hipLaunchKernelGGL(gate_grad3d_kernel<4>, dim3(blocks), dim3(256), 0, st, feat, hist, ahist, gout, gg, B, D, H, W, n_iter, C);
        else
            hipLaunchKernelGGL(gate_grad3d_kernel<1>, dim3(blocks), dim3(256), 0, st, feat, hist, ahist, gout, gg, B, D, H, W, n_iter, C);
        if (int e = check_launch("gate_grad3d_kernel")) return e;
    }
    return 0;
}

}  // namespace cspn
