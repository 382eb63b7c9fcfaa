// tools/ubench_mem.hip -- does streaming global-load traffic slow down a VALU-bound, barrier-stepped loop, and by how much?
// Mimics the fused CSPN loop: 8 waves per workgroup (2 per SIMD), per "step" ~88 packed FMAs per wave + one s_barrier; every
// third step each wave issues 9 8-byte-per-lane loads (512 B per wave-instruction, rows 4864 B apart, planes 1.48 MB apart)
// that are consumed three steps later.  Modes: 0 no loads, 1 streaming from HBM, 2 same instructions over a 1 MB region
// (cache hits), 3 streaming with 16-byte loads (half the instructions), 4 streaming, no barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const float* __restrict__ src,
                                                                                    float* __restrict__ dst, int steps,
                                                                                    size_t plane, size_t pitch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f2 acc[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = f2{0.f, 1.f * i}; w[i] = f2{1.0001f, 0.9999f}; }
    f2 pend[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) pend[i] = f2{0.f, 0.f};
    // each workgroup streams its own rows; each wave its own half-row
    size_t row = (size_t)blockIdx.x * 480 + (wv >> 1);
    const size_t col = (size_t)(wv & 1) * 128 + 2 * lane;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(w[i], acc[(i + 1) & 15], acc[i]);
        if (MODE != 0 && (s % 3) == 2) {
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] += pend[i];
            const size_t r = (MODE == 2) ? (row & 31) : row;
            if (MODE == 3) {
                // 16-byte loads: 5 instructions carry (more than) the same bytes
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)i * plane + r * pitch + 2 * col);
                    pend[i] = f2{v.x + v.z, v.y + v.w};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    pend[i] = *reinterpret_cast<const f2*>(src + (size_t)i * plane + r * pitch + col + (i % 3) - 1 + 1);
            }
            row += 4;
        }
        if (MODE != 4) __builtin_amdgcn_s_barrier();
    }
    f2 t = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t.x == 12345.678f) dst[threadIdx.x] = t.y;
}

template <int MODE>
float run(const float* src, float* dst, int steps, size_t plane, size_t pitch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, src, dst, steps, plane, pitch);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, src, dst, steps, plane, pitch);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 50;
}

int main() {
    const size_t pitch = 1216, plane = (size_t)304 * 1216 * 64 / 9 * 0 + (size_t)370000 * 4;  // floats
    const size_t total = 9 * plane + (size_t)256 * 480 * pitch + 4096;
    float *src, *dst;
    hipMalloc(&src, total * sizeof(float));
    hipMalloc(&dst, 4096);
    hipMemset(src, 0, total * sizeof(float));
    const int steps = 400;
    const double bytes = 256.0 * 8 * (steps / 3) * 9 * 512;
    printf("mode0 no loads        %.4f ms\n", run<0>(src, dst, steps, plane, pitch));
    float t1 = run<1>(src, dst, steps, plane, pitch);
    printf("mode1 streaming       %.4f ms  (%.2f GB -> %.2f TB/s)\n", t1, bytes / 1e9, bytes / t1 / 1e9);
    printf("mode2 cache-resident  %.4f ms\n", run<2>(src, dst, steps, plane, pitch));
    printf("mode3 16-byte loads   %.4f ms\n", run<3>(src, dst, steps, plane, pitch));
    printf("mode4 no barrier      %.4f ms\n", run<4>(src, dst, steps, plane, pitch));
    return 0;
}
