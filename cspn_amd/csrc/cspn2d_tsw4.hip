// cspn2d_tsw4.hip -- 24 propagation steps of Affinity_Propagate.forward (reference cspn_pytorch/models/cspn.py:42-83 incl.
// affinity_normalization :85-144, pad_blur_depth :147-172, sum_conv :44-53, tail :70-81) in one launch: the round-6 loop.
//
// The time-skewed wave ring of cspn2d_tsw.hip re-cut for THREE waves per SIMD: 12 waves x 3 resident rows x 4
// columns per lane at 168 VGPRs instead of 8 x 4 at 256 (tools/tswgen/kernel4.py generates the loop, cspn2d_tsw4_gen.inc;
// tools/tswgen/emu.py runs the same instruction list on the CPU against the oracle, tests/test_tswgen4.py):
//   * raw rows arrive by LDS-DMA (global_load_lds_dwordx4) five steps ahead of their use in a pool of 9 row slots,
//   * are cooked in place by one-pixel-per-lane tasks (every wave has a static role per ring counter),
//   * and injected with ten ds_read_b128.
// The C++ part only (1) builds the descriptor table in LDS (the same 16-byte descriptors and the same plans as cspn2d_tsw.hip),
// (2) hands kernel arguments to the asm block in fixed SGPRs.  Full first passes of 24 iterations only; everything else stays
// on cspn2d_tsw.hip's loop.
#include "cspn_common.h"
#include "cspn2d_tsw_plan.h"
#include "cspn2d_tsw_desc.h"
#ifdef TSW4_GEN_INC
#include TSW4_GEN_INC
#else
#include "cspn2d_tsw4_gen.inc"
#endif

namespace cspn {
namespace {

using namespace tswplan;
constexpr int NT4 = 64 * TSW4_NW;

template <int NORM, int SPARSE>
struct Tsw4Asm;
#define TSW4_VARIANT(N, S)                                                                                              \
    template <> struct Tsw4Asm<N, S> {                                                                                  \
        static __device__ __forceinline__ void run(int lane, const float* gd, const float* blur, const float* sparse,   \
                                                   float* out, const void* aux, int W4, int HW4, int last, int wv,      \
                                                   unsigned ldsb) {                                                     \
            asm volatile(TSW4_ASM_##N##_##S                                                                             \
                         :                                                                                              \
                         : "{v0}"(lane), "{s[16:17]}"(gd), "{s[18:19]}"(blur), "{s[20:21]}"(blur), "{s[22:23]}"(sparse), \
                           "{s[24:25]}"(out), "{s[26:27]}"(aux), "{s28}"(W4), "{s29}"(HW4), "{s30}"(last), "{s31}"(wv), \
                           "{s15}"(ldsb)                                                                                \
                         : TSW4_CLOBBERS);                                                                              \
        }                                                                                                               \
    };

#define TSW4_CLOBBERS                                                                                                  \
    "memory", "vcc", "scc", "m0", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", \
        "s14", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45",             \
        "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",      \
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75",      \
        "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90",      \
        "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", \
        "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",      \
        "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",      \
        "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",      \
        "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",      \
        "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",      \
        "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102",   \
        "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",       \
        "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128",       \
        "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",       \
        "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154",       \
        "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"

#ifndef TSW4_PART
#define TSW4_PART -1   // everything (single-variant timing builds)
#endif
#define TSW4_IN(p) (TSW4_PART == -1 || TSW4_PART == (p))
#if TSW4_IN(0) || defined(TSW4_SINGLE_VARIANT)
TSW4_VARIANT(0, 0)
#endif
#ifdef TSW4_SINGLE_VARIANT  // timing experiments: every variant runs the one generated loop
template <int NORM, int SPARSE>
struct Tsw4Asm : Tsw4Asm<0, 0> {};
#else
#if TSW4_IN(0)
TSW4_VARIANT(0, 1)
#endif
#if TSW4_IN(1)
TSW4_VARIANT(1, 0)
TSW4_VARIANT(1, 1)
#endif
#if TSW4_IN(2)
TSW4_VARIANT(2, 0)
TSW4_VARIANT(2, 1)
#endif
#if TSW4_IN(3)
TSW4_VARIANT(3, 0)
TSW4_VARIANT(3, 1)
#endif
#endif

#ifdef TSW4_TRACE
__device__ char* g_tsw4_trace = nullptr;
constexpr size_t TSW4_TRACE_WG_BYTES = 1024 * TSW4_NW * 32;
#endif

// the step in which stream row Q - 1 retires (tools/tswgen/kernel4.py last_step): row q enters at step 2 (q div 3) + q mod 3
__device__ __forceinline__ int tsw4_last_step(int Q) {
    return Q > 0 ? 2 * ((Q - 1) / TSW4_NSLOT) + (Q - 1) % TSW4_NSLOT + LV : -1;
}

template <int NORM, int SPARSE>
__global__ __launch_bounds__(NT4) void cspn2d_tsw4_kernel(const float* __restrict__ gd, const float* __restrict__ blur,
                                                          const float* __restrict__ sparse, float* __restrict__ out,
                                                          const PlanGeo g, int W4, int HW4) {
    __shared__ __attribute__((aligned(16))) char lds[TSW4_LDS_BYTES];
    int Q = 0, Qe;
    (void)tsw_desc(g, blockIdx.x, -1, &Q);   // every thread: the stream length
    if (Q == 0) return;                      // idle workgroup (uniform)
    uint4* tab = reinterpret_cast<uint4*>(lds + TSW4_LDS_TAB);
    // boundary-row buffers and row slots start out as zeros (the first steps read them before anyone wrote)
    for (int i = threadIdx.x; i < TSW4_LDS_TAB / 16; i += NT4) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
    // the workgroup plans its own stream: its threads write the row descriptors straight into the LDS table
    for (int e = threadIdx.x; e < g.stride; e += NT4) tab[e] = tsw_desc(g, blockIdx.x, e - TSW4_PADF, &Qe);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsb = (unsigned)(uintptr_t)lds;  // LDS address of the block (0 unless the compiler adds more shared data)
#ifdef TSW4_TRACE
    const void* aux = (const void*)(g_tsw4_trace + (size_t)blockIdx.x * TSW4_TRACE_WG_BYTES);
#else
    const void* aux = nullptr;
#endif
    Tsw4Asm<NORM, SPARSE>::run(lane, gd, blur, sparse, out, aux, W4, HW4, __builtin_amdgcn_readfirstlane(tsw4_last_step(Q)), wv, ldsb);
}

template <int NORM>
void launch4(bool sp, const PlanGeo& g, hipStream_t st, const float* gd, const float* blur, const float* sparse, float* out) {
    const int W4 = 4 * g.W, HW4 = 4 * g.H * g.W;
    if (sp) hipLaunchKernelGGL((cspn2d_tsw4_kernel<NORM, 1>), dim3(g.n_wg), dim3(NT4), 0, st, gd, blur, sparse, out, g, W4, HW4);
    else hipLaunchKernelGGL((cspn2d_tsw4_kernel<NORM, 0>), dim3(g.n_wg), dim3(NT4), 0, st, gd, blur, sparse, out, g, W4, HW4);
}

}  // namespace

#define TSW4_DECL(N) void tsw4_launch_norm##N(bool sp, const tswplan::PlanGeo& g, hipStream_t st, const float* gd, const float* blur, const float* sparse, float* out);
TSW4_DECL(0) TSW4_DECL(1) TSW4_DECL(2) TSW4_DECL(3)
#define TSW4_DEF(N)                                                                                                                 \
    void tsw4_launch_norm##N(bool sp, const tswplan::PlanGeo& g, hipStream_t st, const float* gd, const float* blur, const float* sparse, float* out) { \
        launch4<N>(sp, g, st, gd, blur, sparse, out);                                                                                \
    }
#if TSW4_IN(0)
TSW4_DEF(0)
#endif
#if TSW4_IN(1)
TSW4_DEF(1)
#endif
#if TSW4_IN(2)
TSW4_DEF(2)
#endif
#if TSW4_IN(3)
TSW4_DEF(3)
#endif

#if TSW4_IN(0)
// A FIRST pass of exactly 24 iterations (level 0 = blur) over images at least one band wide
bool tsw4_supported(int B, int H, int W) { return tsw2d_supported(B, H, W); }

// Where the 12 x 3 ring pays (profiles/r06_ring_ab_sweep.md, MI355X): its six lead-in steps and its longer prologue cost 5 .. 17 % on
// short streams, from ~190 stream rows per workgroup on it is 1 .. 3 % faster than the 8 x 4 ring (with a mask: from ~420 rows).  Only on
// the linear plan (one piece per CU; a table that does not fit this loop's LDS falls back to band groups: the 8 x 4 ring takes those).
bool tsw4_preferred(int B, int H, int W, bool sparse) {
    if (!tsw4_supported(B, H, W)) return false;
    const PlanGeo& g = tswplan::make_geo_linear(B, H, W, TSW4_PADF, TSW4_PADB, TSW4_TAB_MAX_ROWS, 0);
    if (g.kind != 1) return false;
    const int longest = g.stride - TSW4_PADF - TSW4_PADB;
    return longest >= (sparse ? 420 : 190);
}

int tsw4_pass(const float* gd, const float* blur, const float* sparse, float* out, int B, int H, int W, int norm, hipStream_t st, int plan_mode) {
    // the linear plan of the forward passes (cspn2d_tsw_plan.h: one piece per CU, cuts where the longest stream is shortest); band
    // groups with more workgroups than CUs when a piece's table would not fit this loop's LDS
    const PlanGeo& g = tswplan::make_geo_linear(B, H, W, TSW4_PADF, TSW4_PADB, TSW4_TAB_MAX_ROWS, plan_mode & 3);
    switch (norm) {
        case 0: tsw4_launch_norm0(sparse != nullptr, g, st, gd, blur, sparse, out); break;
        case 1: tsw4_launch_norm1(sparse != nullptr, g, st, gd, blur, sparse, out); break;
        case 2: tsw4_launch_norm2(sparse != nullptr, g, st, gd, blur, sparse, out); break;
        default: tsw4_launch_norm3(sparse != nullptr, g, st, gd, blur, sparse, out); break;   // CSPN_NORM_PRENORM
    }
    return check_launch("cspn2d_tsw4_kernel");
}

#ifdef TSW4_TRACE
extern "C" int cspn_debug_tsw4_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tsw4_trace), &p, sizeof(p)); }
#endif
#endif

}  // namespace cspn
