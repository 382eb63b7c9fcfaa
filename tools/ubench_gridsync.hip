// tools/ubench_gridsync.hip -- what does one device-wide synchronisation cost on MI355X when 256 resident workgroups
// (1 per CU, 1024 threads) iterate in lock step?  Decides the design of the persistent 3D kernel:
// each propagation step needs the neighbours' boundary voxels of the previous step.
//   mode 0: central counter barrier (atomicAdd + spin), agent-scope release/acquire fences by one lane
//   mode 1: neighbour flags only (each workgroup publishes its step number, waits for 6 neighbours)
//   mode 2: mode 1 + every workgroup writes `face_bytes` of boundary data before the flag and reads as many bytes of its
//           neighbours' data after it (plain stores + agent release / agent acquire + plain loads)
//   mode 3: mode 0 + the same data exchange
//   mode 4: hierarchical barrier: the workgroups of one XCD (workgroup id % 8) count on an XCD-local counter with L2-level
//           atomics (no sc1: all its users share that L2), the last arriver of each XCD counts on a device counter, everybody
//           polls the device counter
//   mode 5: neighbour flags + data exchange WITHOUT fences: 16-byte sc1 stores / sc1 loads for the data (coherent at the
//           memory side, MI355X_MICROARCH.md), s_waitcnt vmcnt(0) before the flag
// Spins are bounded: a broken protocol reports an error instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned SPIN_MAX = 1u << 16;

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sc1(float4* p, float4 v) {
    const v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 ld16_sc1(const float4* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* ctr, unsigned* flags, float4* data, int face_vec, int iters, unsigned* err,
                                          float* sink) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    __shared__ int bail;
    if (tid == 0) bail = 0;
    __syncthreads();
    const int nb[6] = {(wg + 1) % nwg, (wg + nwg - 1) % nwg, (wg + 8) % nwg, (wg + nwg - 8) % nwg, (wg + 40) % nwg, (wg + 4 * nwg - 40) % nwg};
    for (int it = 1; it <= iters; ++it) {
        float4* mine = data + ((size_t)(it & 1) * nwg + wg) * face_vec;
        if (MODE == 5) {
            for (int i = tid; i < face_vec; i += 1024) st16_sc1(mine + i, make_float4(it, wg, i, acc.x));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE >= 2 && MODE != 4) {
            for (int i = tid; i < face_vec; i += 1024) mine[i] = make_float4(it, wg, i, acc.x);
        }
        __syncthreads();
        if (MODE == 4) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const unsigned x = wg & 7, per_x = (nwg + 7 - x) / 8;
                const unsigned old = __hip_atomic_fetch_add(flags + 64 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((old + 1) % per_x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)it * (nwg < 8 ? nwg : 8);
                unsigned n = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++n > SPIN_MAX) { *err = 8; bail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else if (MODE == 5) {
            if (tid == 0) __hip_atomic_store(flags + wg, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 6) {
                unsigned n = 0;
                while (__hip_atomic_load(flags + nb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                    if (++n > SPIN_MAX) { *err = 2; bail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        } else if (MODE == 0 || MODE == 3) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)it * nwg;
                unsigned n = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++n > SPIN_MAX) { *err = 1; bail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(flags + wg, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 6) {
                unsigned n = 0;
                while (__hip_atomic_load(flags + nb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                    if (++n > SPIN_MAX) { *err = 2; bail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (bail) break;   // a broken protocol ends the run instead of spinning through every iteration
        if (MODE >= 2 && MODE != 4) {
            const int per = face_vec / 6;
            for (int i = tid; i < face_vec; i += 1024) {
                const int f = i / (per > 0 ? per : 1);
                const float4* src = data + ((size_t)(it & 1) * nwg + nb[f < 6 ? f : 5]) * face_vec + i;
                const float4 v = MODE == 5 ? ld16_sc1(src) : *src;
                if (v.x != (float)it) atomicOr(err, 4u);  // stale data from the neighbour
                acc.x += v.y; acc.y += v.z;
            }
        }
    }
    if (acc.x == -1.f) sink[tid] = acc.y;
}

template <int MODE>
void run(int nwg, int face_bytes, int iters) {
    unsigned *ctr, *flags, *err;
    float4* data;
    float* sink;
    const int face_vec = face_bytes / 16;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&flags, 4 * (nwg > 512 ? nwg : 512))); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&data, (size_t)2 * nwg * (face_vec > 0 ? face_vec : 1) * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(flags, 0, 4 * (nwg > 512 ? nwg : 512))); CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(1024), 0, 0, ctr, flags, data, face_vec, iters, err, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h;
        CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        herr |= h;
    }
    printf("mode %d  nwg %d  face %6d B  iters %d : %8.3f us / iteration  (err %u)\n", MODE, nwg, face_bytes, iters,
           best * 1e3f / iters, herr);
    fflush(stdout);
    hipFree(ctr); hipFree(flags); hipFree(err); hipFree(data); hipFree(sink);
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", ncu);
    fflush(stdout);
    const int iters = 100;
    for (int nwg : {ncu, ncu / 2, 32}) {
        run<0>(nwg, 0, iters);
        run<1>(nwg, 0, iters);
        run<4>(nwg, 0, iters);
        for (int fb : {6144, 24576}) {
            run<2>(nwg, fb, iters);
            run<3>(nwg, fb, iters);
            run<5>(nwg, fb, iters);
        }
    }
    return 0;
}
