// tools/ubench_occ.hip -- what would more waves per SIMD buy a barrier-stepped VALU loop?  Same packed-FMA work per SIMD and
// step (160 v_pk_fma_f32), split over 2, 3 or 4 waves per SIMD, one s_barrier per step, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WAVES, int REPS>  // REPS x 16 packed FMAs per wave and step
__global__ __launch_bounds__(WAVES * 64) void k(float* dst, int steps) {
    f2 acc[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = f2{0.f, 1.f * i}; w[i] = f2{1.0001f, 0.9999f}; }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int r = 0; r < REPS; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(w[i], acc[(i + 1) & 15], acc[i]);
        __builtin_amdgcn_s_barrier();
    }
    f2 t = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t.x == 12345.678f) dst[threadIdx.x] = t.y;
}

template <int WAVES, int REPS>
void run(float* dst, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 1200;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<WAVES, REPS>), dim3(256), dim3(WAVES * 64), 0, 0, dst, steps);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((k<WAVES, REPS>), dim3(256), dim3(WAVES * 64), 0, 0, dst, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_step_ns = ms / 30 / steps * 1e6;
    printf("%s: %d waves/WG x %d pk_fma per wave-step: %.1f ns per step (%.2f cycles per pk_fma per SIMD at 2.4 GHz)\n", name, WAVES,
           REPS * 16, per_step_ns, per_step_ns * 2.4 / (WAVES / 4 * REPS * 16));
}

int main() {
    float* dst;
    (void)hipMalloc(&dst, 4096);
    run<8, 6>(dst, "2 waves/SIMD");    // 96 per wave -> 192 per SIMD
    run<12, 4>(dst, "3 waves/SIMD");   // 64 per wave -> 192 per SIMD
    run<16, 3>(dst, "4 waves/SIMD");   // 48 per wave -> 192 per SIMD
    run<4, 12>(dst, "1 wave/SIMD");    // 192 per wave
    return 0;
}
