// tools/ubench.hip -- design-validation micro-benchmarks for the fused CSPN kernel (gfx950).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench.hip -o tools/ubench
// Measures, with 512-thread workgroups (2 waves/SIMD), one WG per CU:
//   1. DPP wave_shr:1 / wave_shl:1 semantics (lane i <- lane i-1 / i+1, edge lanes read 0)
//   2. v_fmac_f32 vs v_pk_fma_f32 issue rate
//   3. the register-resident "4 slots x 4 columns" propagation step (the fused kernel's core),
//      alone and with the per-step LDS boundary exchange + s_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float dpp_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__global__ void k_dpp(const float* in, float* o_shr, float* o_shl) {
    int t = threadIdx.x + blockIdx.x * blockDim.x;
    float v = in[t];
    o_shr[t] = dpp_shr1(v);
    o_shl[t] = dpp_shl1(v);
}

// ---- raw FMA issue rate ----
__global__ __launch_bounds__(512, 2) void k_fmac(float* out, float a, float b, int iters) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512, 2) void k_pkfma(float* out, float a, float b, int iters) {
    f2 acc[8];
    f2 av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// ---- propagation step core ----
constexpr int C = 4;
__device__ __forceinline__ void upd(const float (&w)[9][C], const float (&ab)[C], const float (&se)[C],
                                    const float (&be)[C], float (&o)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float acc = w[8][c];
        float bR = c < C - 1 ? be[c + 1] : dpp_shl1(be[0]);
        float bL = c > 0 ? be[c - 1] : dpp_shr1(be[C - 1]);
        float sR = c < C - 1 ? se[c + 1] : dpp_shl1(se[0]);
        float sL = c > 0 ? se[c - 1] : dpp_shr1(se[C - 1]);
        float aR = c < C - 1 ? ab[c + 1] : dpp_shl1(ab[0]);
        float aL = c > 0 ? ab[c - 1] : dpp_shr1(ab[C - 1]);
        acc = fmaf(w[0][c], bR, acc); acc = fmaf(w[1][c], be[c], acc); acc = fmaf(w[2][c], bL, acc);
        acc = fmaf(w[3][c], sR, acc); acc = fmaf(w[4][c], sL, acc);
        acc = fmaf(w[5][c], aR, acc); acc = fmaf(w[6][c], ab[c], acc); acc = fmaf(w[7][c], aL, acc);
        o[c] = acc;
    }
}

template <bool EXCH>
__global__ __launch_bounds__(512, 2) void k_step(const float* __restrict__ in, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) float bnd[2][8][2][256];
    float w[4][9][C], h[2][4][C], top[C], bot[C];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) w[j][k][c] = in[((j * 9 + k) * C + c) * 512 + t] * 0.1f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) { h[0][j][c] = in[t + j + c]; h[1][j][c] = in[t + 7 * j + c]; }
#pragma unroll
    for (int c = 0; c < C; ++c) { top[c] = in[c]; bot[c] = in[c + 9]; }
    if (EXCH) {
        for (int i = t; i < 2 * 8 * 2 * 256; i += 512) (&bnd[0][0][0][0])[i] = 0.f;
        __syncthreads();
    }
    const int up = (wv + 7) & 7, dn = (wv + 1) & 7;
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (EXCH) {
                const float4 a = *reinterpret_cast<const float4*>(&bnd[par ^ 1][up][1][lane * 4]);
                const float4 b = *reinterpret_cast<const float4*>(&bnd[par ^ 1][dn][0][lane * 4]);
                top[0] = a.x; top[1] = a.y; top[2] = a.z; top[3] = a.w;
                bot[0] = b.x; bot[1] = b.y; bot[2] = b.z; bot[3] = b.w;
            }
            const int cu = par, pv = par ^ 1;
            upd(w[3], h[pv][2], h[cu][3], bot, h[pv][3]);
            upd(w[2], h[pv][1], h[cu][2], h[pv][3], h[pv][2]);
            upd(w[1], h[pv][0], h[cu][1], h[pv][2], h[pv][1]);
            upd(w[0], top, h[cu][0], h[pv][1], h[pv][0]);
            if (EXCH) {
                *reinterpret_cast<float4*>(&bnd[par][wv][0][lane * 4]) = make_float4(h[pv][0][0], h[pv][0][1], h[pv][0][2], h[pv][0][3]);
                *reinterpret_cast<float4*>(&bnd[par][wv][1][lane * 4]) = make_float4(h[pv][3][0], h[pv][3][1], h[pv][3][2], h[pv][3][3]);
                __syncthreads();
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) r += h[0][j][c] + h[1][j][c];
    out[blockIdx.x * 512 + t] = r;
}

template <class F>
static float time_ms(F&& launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz, LDS/block %zu\n", p.gcnArchName, ncu, p.clockRate / 1000, p.sharedMemPerBlock);
    // 1. DPP semantics
    {
        const int n = 128;
        std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 100.f + i;
        float *d_in, *d_a, *d_b; CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4));
        CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_dpp, dim3(1), dim3(n), 0, 0, d_in, d_a, d_b);
        std::vector<float> a(n), b(n);
        CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            float es = (i % 64 == 0) ? 0.f : h[i - 1], el = (i % 64 == 63) ? 0.f : h[i + 1];
            if (a[i] != es || b[i] != el) ++bad;
        }
        printf("dpp: wave_shr:1 lane1<-%g (expect %g) lane0=%g lane64=%g | wave_shl:1 lane0<-%g lane63=%g ; mismatches=%d\n",
               a[1], h[0], a[0], a[64], b[0], b[63], bad);
    }
    float* d_out; CK(hipMalloc(&d_out, (size_t)ncu * 4 * 512 * 4));
    float* d_in; CK(hipMalloc(&d_in, 1 << 20));
    {
        std::vector<float> h(1 << 18); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
        CK(hipMemcpy(d_in, h.data(), 1 << 20, hipMemcpyHostToDevice));
    }
    // 2. FMA issue rate
    for (int wgs_per_cu = 1; wgs_per_cu <= 1; ++wgs_per_cu) {
        const int iters = 4000;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fmac, dim3(ncu * wgs_per_cu), dim3(512), 0, 0, d_out, 0.999f, 0.001f, iters); });
        double fma = (double)ncu * wgs_per_cu * 512 * iters * 128.0;
        printf("v_fmac_f32 : %.3f ms, %.1f T FMA/s (%.1f TFLOP/s), %.1f FMA/clk/CU @2.4GHz\n", ms, fma / ms / 1e9, 2 * fma / ms / 1e9, fma / (ms * 1e-3) / ncu / 2.4e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(ncu * wgs_per_cu), dim3(512), 0, 0, d_out, 0.999f, 0.001f, iters); });
        fma = (double)ncu * wgs_per_cu * 512 * iters * 128.0;  // 64 pk instr x 2
        printf("v_pk_fma_f32: %.3f ms, %.1f T FMA/s (%.1f TFLOP/s), %.1f FMA/clk/CU @2.4GHz\n", ms, fma / ms / 1e9, 2 * fma / ms / 1e9, fma / (ms * 1e-3) / ncu / 2.4e9);
    }
    // 3. propagation step core
    {
        const int steps = 2000;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_step<false>, dim3(ncu), dim3(512), 0, 0, d_in, d_out, steps); });
        double pxit = (double)ncu * 512 * steps * 16.0;
        printf("step core (no exchange): %.3f ms, %.3f us/step, %.2f T px-iter/s, %.1f FMA/clk/CU @2.4GHz\n", ms, ms * 1e3 / steps, pxit / ms / 1e9, pxit * 8 / (ms * 1e-3) / ncu / 2.4e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_step<true>, dim3(ncu), dim3(512), 0, 0, d_in, d_out, steps); });
        printf("step core (+LDS exchange+barrier): %.3f ms, %.3f us/step, %.2f T px-iter/s, %.1f FMA/clk/CU @2.4GHz\n", ms, ms * 1e3 / steps, pxit / ms / 1e9, pxit * 8 / (ms * 1e-3) / ncu / 2.4e9);
    }
    return 0;
}
