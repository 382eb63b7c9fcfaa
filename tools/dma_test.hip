// probe: does global_load_lds_dwordx4 accept a 4-byte-aligned (not 16-byte-aligned) per-lane source address?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ g, float* __restrict__ out, int shift) {
    __shared__ __attribute__((aligned(16))) float stage[2][256];
    const int lane = threadIdx.x & 63;
    const float* base = g;                                  // uniform
    const unsigned voff = (unsigned)(lane * 16 + shift * 4);  // per-lane byte offset, 4-B aligned
    const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&stage[0][0];
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(m0v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = stage[0][i];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        std::vector<float> r(256);
        hipError_t e = hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (r[i] != (float)(i + shift)) ++bad;
        printf("shift %d: err=%d mismatches=%d  first values %g %g %g %g %g\n", shift, (int)e, bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
