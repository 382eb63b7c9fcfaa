// tools/ubench_issue.hip -- does a wave's SALU / LDS instruction issue for free beside its SIMD partner's VALU work?
// 8 waves per workgroup (2 per SIMD, 256 VGPRs each as in the fused CSPN loop), 256 workgroups, one s_barrier per "step".
// A step = 64 independent v_pk_fma_f32 per wave + K extra instructions of one kind (s_add_u32 / s_mul_i32 / ds_read_b32 /
// v_mov_b32), either interleaved one by one between the FMAs or clustered at the top of the step.  If the extra instructions
// co-issued with the partner wave's VALU, time would not grow with K until K ~ 64.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND, int K, int CLUSTER>
__global__ __launch_bounds__(512) void k(float* out, int steps) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned a = (threadIdx.x & 63) * 4;
    asm volatile("s_mov_b32 s41, 1\n\ts_mov_b32 s43, 1" ::: "s41", "s43");
    for (int s = 0; s < steps; ++s) {
        // 64 FMAs on v[64:191] with K extras; the asm text is built from pieces so that the extras sit between FMAs
#define FMA(i) "v_pk_fma_f32 v[" #i ":" #i "+1], v[200:201], v[202:203], v[" #i ":" #i "+1]\n\t"
#define EXTRA_S "s_add_u32 s40, s41, 3\n\t"
#define EXTRA_M "s_mul_i32 s42, s43, 5\n\t"
#define EXTRA_L "ds_read_b32 v210, %0\n\t"
#define EXTRA_V "v_mov_b32 v211, v212\n\t"
#define EXTRA (KIND == 0 ? EXTRA_S : KIND == 1 ? EXTRA_M : KIND == 2 ? EXTRA_L : EXTRA_V)
        if (CLUSTER) {
            for (int i = 0; i < K; ++i) {
                if (KIND == 0) asm volatile(EXTRA_S ::: "s40", "scc");
                else if (KIND == 1) asm volatile(EXTRA_M ::: "s42", "scc");
                else if (KIND == 2) asm volatile(EXTRA_L :: "v"(a) : "v210");
                else asm volatile(EXTRA_V ::: "v211");
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            asm volatile("v_pk_fma_f32 v[64:65], v[200:201], v[202:203], v[64:65]\n\t"
                         "v_pk_fma_f32 v[66:67], v[200:201], v[202:203], v[66:67]\n\t"
                         "v_pk_fma_f32 v[68:69], v[200:201], v[202:203], v[68:69]\n\t"
                         "v_pk_fma_f32 v[70:71], v[200:201], v[202:203], v[70:71]" ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v200", "v201", "v202", "v203", "v255");
            if (!CLUSTER && g * K / 16 != (g + 1) * K / 16) {
                for (int i = g * K / 16; i < (g + 1) * K / 16; ++i) {
                    if (KIND == 0) asm volatile(EXTRA_S ::: "s40", "scc");
                    else if (KIND == 1) asm volatile(EXTRA_M ::: "s42", "scc");
                    else if (KIND == 2) asm volatile(EXTRA_L :: "v"(a) : "v210");
                    else asm volatile(EXTRA_V ::: "v211");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (steps < 0) out[threadIdx.x] = lds[threadIdx.x];
}

template <int KIND, int K, int CLUSTER>
float run(float* out, int steps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<KIND, K, CLUSTER>), dim3(256), dim3(512), 0, 0, out, steps);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<KIND, K, CLUSTER>), dim3(256), dim3(512), 0, 0, out, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    const int steps = 400;
    const char* kinds[] = {"s_add_u32", "s_mul_i32", "ds_read_b32", "v_mov_b32"};
#define ROW(KIND) printf("%-12s interleaved K=0/16/32/64: %.1f %.1f %.1f %.1f ns/step   clustered K=16/32/64: %.1f %.1f %.1f\n", kinds[KIND], \
        run<KIND, 0, 0>(out, steps) * 1e6 / steps, run<KIND, 16, 0>(out, steps) * 1e6 / steps, run<KIND, 32, 0>(out, steps) * 1e6 / steps, \
        run<KIND, 64, 0>(out, steps) * 1e6 / steps, run<KIND, 16, 1>(out, steps) * 1e6 / steps, run<KIND, 32, 1>(out, steps) * 1e6 / steps, \
        run<KIND, 64, 1>(out, steps) * 1e6 / steps);
    ROW(0) ROW(1) ROW(2) ROW(3)
    return 0;
}
