#include <hip/hip_runtime.h>
__device__ __forceinline__ float dpp_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
constexpr int C = 4;
__device__ __forceinline__ void upd(const float (&w)[9][C], const float (&ab)[C], const float (&se)[C], const float (&be)[C], float (&o)[C]) {
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
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ in, float* __restrict__ out, int steps) {
    float w[4][9][C], h[2][4][C], top[C], bot[C];
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) w[j][k][c] = in[((j * 9 + k) * C + c) * 512 + t];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) { h[0][j][c] = in[t + j + c]; h[1][j][c] = in[t + 7 * j + c]; }
#pragma unroll
    for (int c = 0; c < C; ++c) { top[c] = in[c]; bot[c] = in[c + 9]; }
    for (int s = 0; s < steps; s += 2) {
        // parity 0: cur = h[0], prev = h[1]; new -> h[1]
        upd(w[3], h[1][2], h[0][3], bot, h[1][3]);
        upd(w[2], h[1][1], h[0][2], h[1][3], h[1][2]);
        upd(w[1], h[1][0], h[0][1], h[1][2], h[1][1]);
        upd(w[0], top, h[0][0], h[1][1], h[1][0]);
        // parity 1
        upd(w[3], h[0][2], h[1][3], bot, h[0][3]);
        upd(w[2], h[0][1], h[1][2], h[0][3], h[0][2]);
        upd(w[1], h[0][0], h[1][1], h[0][2], h[0][1]);
        upd(w[0], top, h[1][0], h[0][1], h[0][0]);
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) r += h[0][j][c] + h[1][j][c];
    out[blockIdx.x * 512 + t] = r;
}
