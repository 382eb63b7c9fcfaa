// tools/r05/ubench_xcd_rtt.hip -- one-way latency of a tagged 16-byte hand-off between two workgroups on MI355X, by placement and store flavour:
// what an exchange step between CUs costs when the partner shares the XCD (the line can stay in that XCD's L2) and when it does not.
// 256 workgroups (one per CU, 64 threads); workgroup w ping-pongs with partner w ^ PM: PM = 8 -> ids 8 apart (same XCD by the observed
// round-robin placement), PM = 1 -> neighbouring ids (different XCDs).  The sender stores {payload x3, tag} (plain or sc1), the receiver polls
// with sc1 loads (they bypass its L1) until the tag is the expected one, then answers.  All 128 pairs run at once (a loaded fabric, like the
// kernels that would use it); the XCC ids actually seen are checked.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/r05/ubench_xcd_rtt tools/r05/ubench_xcd_rtt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(float4* p, v4f x, bool sc1) {
    if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ v4f ld16_sc1(const float4* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// box[wg][2]: the slot the PARTNER writes for wg (two parities, 128 bytes apart from anything else)
__global__ __launch_bounds__(64) void k(float4* box, int pm, int sc1, int iters, unsigned* xcc, unsigned* err, long long* cycles) {
    const int wg = blockIdx.x, partner = wg ^ pm;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[wg] = x & 15u;
    if (threadIdx.x != 0) return;
    const bool first = wg < partner;
    float4* out = box + ((size_t)partner * 2) * 8;   // what the partner reads
    const float4* in = box + ((size_t)wg * 2) * 8;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 1; it <= iters; ++it) {
        const unsigned tag = (unsigned)it;
        if (first) st16(out + (it & 1) * 8, v4f{1.f, 2.f, 3.f, __uint_as_float(tag)}, sc1 != 0);
        unsigned spins = 0;
        for (;;) {
            const v4f v = ld16_sc1(in + (it & 1) * 8);
            if (__float_as_uint(v.w) == tag) break;
            if (++spins > (1u << 20)) { *err = 1; return; }
        }
        if (!first) st16(out + (it & 1) * 8, v4f{1.f, 2.f, 3.f, __uint_as_float(tag)}, sc1 != 0);
    }
    cycles[wg] = __builtin_readcyclecounter() - t0;
}
int main() {
    const int nwg = 256, iters = 2000;
    float4* box; unsigned *xcc, *err; long long* cyc;
    CK(hipMalloc(&box, (size_t)nwg * 2 * 8 * sizeof(float4)));
    CK(hipMalloc(&xcc, nwg * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&cyc, nwg * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const struct { int pm, sc1; const char* name; } cases[] = {{8, 0, "same XCD (ids 8 apart), plain store"}, {8, 1, "same XCD (ids 8 apart), sc1 store"},
                                                                {1, 1, "other XCD (neighbouring ids), sc1 store"}, {1, 0, "other XCD, plain store (NOT coherent: expected to time out)"}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(box, 0, (size_t)nwg * 2 * 8 * sizeof(float4))); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, box, c.pm, c.sc1, c.pm == 1 && !c.sc1 ? 50 : iters, xcc, err, cyc);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hx[256], he; CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
            int same = 0; for (int w = 0; w < nwg; ++w) same += hx[w] == hx[w ^ c.pm];
            if (rep == 1) {
                const int n = c.pm == 1 && !c.sc1 ? 50 : iters;
                printf("%-62s: %7.3f us per round trip (2 hops) = %6.3f us per hop; partners on the same XCC: %d / 256; timeout %u\n", c.name, ms * 1e3 / n, ms * 1e3 / n / 2, same, he);
            }
        }
    }
    return 0;
}
