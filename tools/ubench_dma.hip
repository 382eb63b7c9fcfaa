// tools/ubench_dma.hip -- design probe for the round-3 ring: can a VALU-bound, barrier-stepped loop (the fused CSPN core:
// 8 waves per workgroup, 80 v_pk_fma_f32 per wave-step, one s_barrier per step) pull its raw guidance rows with LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR destination) several steps ahead at HBM speed?
//   part 1: correctness of a DMA landing zone above 64 KiB of LDS (is M0 wide enough on gfx950?) and of 4-byte-aligned sources
//   part 2: timing: 12 KiB per workgroup-step (= 4 rows x 9 planes x 1 KiB per 3 steps) issued D steps ahead of its use
//           WHO 0: pieces spread over the eight waves; WHO 1: one wave (rotating) issues all twelve of a step
//           CONS: every wave reads its share of the landed slot back from LDS (the cooking side's raw reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define LDS_TOTAL 163840

__device__ __forceinline__ void dma16(unsigned voff, const float* base, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// ---- part 1 ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void probe(const float* __restrict__ g, float* __restrict__ out, unsigned lds_off, int shift) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x;
    const unsigned ldsb = (unsigned)(uintptr_t)lds;
    // poison the landing zone and the same offset modulo 64 KiB
    unsigned z = 0x7fc00000u;
    unsigned a0 = ldsb + lds_off + lane * 16, a1 = ldsb + (lds_off & 0xffff) + lane * 16;
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:4\n\tds_write_b32 %0, %1 offset:8\n\tds_write_b32 %0, %1 offset:12\n\t"
                 "ds_write_b32 %2, %1\n\tds_write_b32 %2, %1 offset:4\n\tds_write_b32 %2, %1 offset:8\n\tds_write_b32 %2, %1 offset:12\n\t"
                 "s_waitcnt lgkmcnt(0)" :: "v"(a0), "v"(z), "v"(a1) : "memory");
    __builtin_amdgcn_s_barrier();
    dma16((unsigned)(lane * 16 + shift * 4), g, ldsb + lds_off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float v[4], u[4];
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                 "ds_read_b32 %4, %9\n\tds_read_b32 %5, %9 offset:4\n\tds_read_b32 %6, %9 offset:8\n\tds_read_b32 %7, %9 offset:12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                 : "v"(a0), "v"(a1) : "memory");
    for (int i = 0; i < 4; ++i) { out[lane * 4 + i] = v[i]; out[256 + lane * 4 + i] = u[i]; }
}

// does the instruction's immediate offset move the LDS destination as well as the global source?
__global__ __launch_bounds__(64) void probe_off(const float* __restrict__ g, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x;
    const unsigned ldsb = (unsigned)(uintptr_t)lds;
    for (int i = lane; i < 1024; i += 64) ((float*)lds)[i] = -1.f;
    __syncthreads();
    unsigned keep;
    unsigned voff = lane * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:512\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(ldsb + 1024) : "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = ((float*)lds)[i];
}

// ---- part 2 ------------------------------------------------------------------------------------------------------------
constexpr int PIECES = 12;          // 1 KiB pieces per workgroup-step
constexpr int SLOT = PIECES * 1024;

template <int D, int WHO, int CONS, int HI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const float* __restrict__ src,
                                                                                    float* __restrict__ dst, int steps,
                                                                                    unsigned plane_b, unsigned pitch_b) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_TOTAL];
    constexpr int NS = D + 2;   // slots: D in flight / landed, one being read, one spare
    static_assert(NS * SLOT + (HI ? 32768 : 0) <= LDS_TOTAL, "ring too large");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ring = (unsigned)(uintptr_t)lds + (HI ? (LDS_TOTAL - NS * SLOT) : 0);
    f2 acc[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = f2{0.f, 1.f * i}; w[i] = f2{1.0001f, 0.9999f}; }
    const float* wg_src = src + (size_t)blockIdx.x * 480 * (pitch_b / 4);
    int piece = 0;   // running piece number of the workgroup's stream
    for (int s = 0; s < steps; ++s) {
        // ---- issue the pieces of step s (consumed in step s + D)
        if (D > 0) {
            const unsigned slot = ring + (unsigned)(s % NS) * SLOT;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const bool mine = WHO == 0 ? ((i & 7) == wv) : ((s & 7) == wv);
                if (mine) {
                    const int n = piece + i;
                    const unsigned row = (unsigned)n / 9u, pl = (unsigned)n % 9u;
                    const unsigned voff = pl * plane_b + row * pitch_b + (unsigned)((i % 3) * 4) + lane * 16;
                    dma16(voff, wg_src, slot + i * 1024);
                }
            }
            piece += PIECES;
        }
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(w[i], acc[(i + 1) & 15], acc[i]);
        if (D > 0 && CONS && s >= D) {
            // the slot issued in step s - D landed before the barrier that ended step s - 1
            const unsigned a = ring + (unsigned)((s - D) % NS) * SLOT + wv * 1536 + lane * 8;
            f2 x, y, z2;
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:512\n\tds_read_b64 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(x), "=&v"(y), "=&v"(z2) : "v"(a) : "memory");
            acc[0] += x; acc[5] += y; acc[10] += z2;
        }
        if (D > 0) {
            // what this wave issued in step s - D + 1 must have landed before the barrier: leave (D - 1) steps' worth in flight
            if (WHO == 0) {
                if (wv < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (D - 1) > 63 ? 63 : 2 * (D - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
            } else {
                // one wave issues 12 pieces every 8th step: its previous batch is >= 8 steps old; wait for it D - 1 steps after issue
                if (((s - (D - 1)) & 7) == wv) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f2 t = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t.x == 12345.678f) dst[threadIdx.x] = t.y;
}

template <int D, int WHO, int CONS, int HI>
float run(const float* src, float* dst, int steps, unsigned plane_b, unsigned pitch_b) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<D, WHO, CONS, HI>), dim3(256), dim3(512), 0, 0, src, dst, steps, plane_b, pitch_b);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<D, WHO, CONS, HI>), dim3(256), dim3(512), 0, 0, src, dst, steps, plane_b, pitch_b);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  (error %s)\n", hipGetErrorString(e));
    return ms / 50;
}

int main() {
    // ---- part 1
    {
        const int n = 4096;
        std::vector<float> h(n);
        for (int i = 0; i < n; ++i) h[i] = (float)i;
        float *d, *o;
        hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        const unsigned offs[] = {0u, 32768u, 65536u - 1024u, 65536u, 65536u + 4096u, 131072u, 163840u - 1024u};
        for (unsigned off : offs)
            for (int shift = 0; shift < 2; ++shift) {
                hipMemset(o, 0, 512 * 4);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, off, shift);
                std::vector<float> r(512);
                hipError_t e = hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
                int bad = 0, bad_lo = 0;
                for (int i = 0; i < 256; ++i) { if (r[i] != (float)(i + shift)) ++bad; if (r[256 + i] != (float)(i + shift)) ++bad_lo; }
                printf("probe lds_off %6u shift %d: err=%d  mismatches at lds_off: %d  at lds_off mod 64K: %d  (first %g %g)\n", off, shift,
                       (int)e, bad, bad_lo, r[0], r[256]);
            }
    }
    {
        const int n = 4096;
        std::vector<float> h(n);
        for (int i = 0; i < n; ++i) h[i] = (float)i;
        float *d, *o;
        hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_off, dim3(1), dim3(64), 0, 0, d, o);
        std::vector<float> r(1024);
        hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
        int first = -1;
        for (int i = 0; i < 1024; ++i) if (r[i] >= 0.f) { first = i; break; }
        printf("offset:512 probe (M0 = lds + 1024 B): first written dword %d (256 = M0 only, 384 = M0 + offset), value there %g (128 = source + offset)\n",
               first, first >= 0 ? r[first] : -1.f);
        if (getenv("PROBE_ONLY")) return 0;
    }
    // ---- part 2
    const unsigned pitch_b = 1216 * 4, plane_b = 370000 * 4;
    const size_t total = (size_t)9 * plane_b + (size_t)256 * 480 * pitch_b + (1 << 20);
    float *src, *dst;
    hipMalloc(&src, total);
    hipMalloc(&dst, 4096);
    hipMemset(src, 0, total);
    const int steps = 400;
    const double bytes = 256.0 * steps * PIECES * 1024;
    printf("bytes per launch %.3f GB\n", bytes / 1e9);
#define RUN(D, WHO, CONS, HI) { float t = run<D, WHO, CONS, HI>(src, dst, steps, plane_b, pitch_b); \
        printf("D=%d who=%d cons=%d hi=%d : %.4f ms  %.2f TB/s\n", D, WHO, CONS, HI, t, D ? bytes / t / 1e9 : 0.0); }
    RUN(0, 0, 0, 0)
    RUN(1, 0, 0, 0)
    RUN(2, 0, 0, 0)
    RUN(3, 0, 0, 0)
    RUN(4, 0, 0, 0)
    RUN(6, 0, 0, 0)
    RUN(8, 0, 0, 0)
    RUN(10, 0, 0, 0)
    RUN(4, 0, 1, 0)
    RUN(6, 0, 1, 0)
    RUN(8, 0, 1, 0)
    RUN(4, 1, 1, 0)
    RUN(6, 1, 1, 0)
    RUN(8, 1, 1, 0)
    RUN(4, 0, 1, 1)
    RUN(6, 0, 1, 1)
    return 0;
}
