// cspn2d_tsw3.hip -- 24 propagation steps of Affinity_Propagate.forward (reference cspn_pytorch/models/cspn.py:42-83 incl.
// affinity_normalization :85-144, pad_blur_depth :147-172, sum_conv :44-53, tail :70-81) in one launch: the round-3 loop.
//
// The same time-skewed wave ring as cspn2d_tsw.hip, with the guidance reaching the ring differently
// (tools/tswgen/kernel3.py generates the loop, cspn2d_tsw3_gen.inc; tools/tswgen/emu.py runs the same instruction list on the
// CPU against the oracle, tests/test_tswgen3.py):
//   * raw rows are fetched by LDS-DMA (global_load_lds_dwordx4) six steps ahead of their use into a pool of 12 row slots,
//   * cooked in place by X / Y tasks (columns c0,c3 / c1,c2 of every lane) spread over the three steps of a group,
//   * injected with ten ds_read_b128,
//   * row descriptors are 4 bytes: the table is 8 KB instead of 44 KB of LDS.
// The C++ part only (1) builds the descriptor table in LDS, (2) hands kernel arguments to the asm block in fixed SGPRs.

#include "cspn_common.h"
#include "cspn2d_tsw_plan.h"
#ifdef TSW3_GEN_INC
#include TSW3_GEN_INC
#else
#include "cspn2d_tsw3_gen.inc"
#endif

namespace cspn {
namespace {

using namespace tswplan;

// descriptor flags / geometry word (tools/tswgen/kernel3.py F_*, G_*)
enum { F3_ACTIVE = 0, F3_UP = 1, F3_DN = 2, F3_OWNED = 3, G3_FIRST = 8, G3_LAST = 9 };

__host__ __device__ inline int ybits_of(int H) {
    int b = 1;
    while ((1 << b) < H) ++b;
    return b;
}

// compact descriptor of stream row q: flags | (image << ybits | y) << 4 (zero: separator / padding row)
__device__ __forceinline__ unsigned tsw3_desc(const PlanGeo& g, int r0, int r1, int q, int yb, int* Q) {
    int b = 0, y = 0;
    bool owned = false;
    if (!tsw_stream_row(g, r0, r1, q, b, y, owned, Q)) return 0u;
    return (1u << F3_ACTIVE) | ((unsigned)(y + 1 < g.H) << F3_UP) | ((unsigned)(y >= 1) << F3_DN) | ((unsigned)owned << F3_OWNED) |
           ((((unsigned)b << yb) | (unsigned)y) << 4);
}

// test hook (cspn_debug_tsw3_dump_plan): one thread per (workgroup, table entry), into global memory
__global__ void cspn2d_plan3_dump_kernel(int4* __restrict__ hdr, unsigned* __restrict__ tab, PlanGeo g) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= g.n_wg * g.stride) return;
    const int wg = gid / g.stride, e = gid - wg * g.stride;
    int bi, r0, r1, Q = 0;
    tsw_wg_share(g, wg, bi, r0, r1);
    const int yb = ybits_of(g.H);
    tab[(size_t)wg * g.stride + e] = tsw3_desc(g, r0, r1, e - TSW3_PADF, yb, &Q);
    if (e == 0) {
        int p0, lo, hi;
        band_of(g, bi, p0, lo, hi);
        hdr[wg] = make_int4(Q, Q > 0 ? 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + g.n_iter : -1, (lo - p0) | ((hi - p0) << 16),
                            (4 * p0) | (yb << 20) | ((p0 == 0) << 28) | ((p0 + BW == g.W) << 29));
    }
}

template <int NORM, int SPARSE, int HIN>
struct Tsw3Asm;
#define TSW3_VARIANT(N, S, H)                                                                                           \
    template <> struct Tsw3Asm<N, S, H> {                                                                               \
        static __device__ __forceinline__ void run(int lane, const float* gd, const float* blur, const float* hin,      \
                                                   const float* sparse, float* out, const void* aux, int W4, int HW4,   \
                                                   int last, int wv, unsigned ldsb, int geom, int lohi, int p04) {      \
            asm volatile(TSW3_ASM_##N##_##S##_##H                                                                       \
                         :                                                                                              \
                         : "{v0}"(lane), "{s[16:17]}"(gd), "{s[18:19]}"(blur), "{s[20:21]}"(hin), "{s[22:23]}"(sparse), \
                           "{s[24:25]}"(out), "{s[26:27]}"(aux), "{s28}"(W4), "{s29}"(HW4), "{s30}"(last), "{s31}"(wv), \
                           "{s15}"(ldsb), "{s14}"(geom), "{s13}"(lohi), "{s12}"(p04)                                    \
                         : TSW3_CLOBBERS);                                                                              \
        }                                                                                                               \
    };

#define TSW3_CLOBBERS                                                                                                  \
    "memory", "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11",           \
        "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45",      \
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
        "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167",       \
        "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180",       \
        "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193",       \
        "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206",       \
        "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219",       \
        "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232",       \
        "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245",       \
        "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

TSW3_VARIANT(0, 0, 0)
#ifdef TSW3_SINGLE_VARIANT  // timing experiments (tools/build_abl3.sh): every variant runs the one generated loop
template <int NORM, int SPARSE, int HIN>
struct Tsw3Asm : Tsw3Asm<0, 0, 0> {};
#else
TSW3_VARIANT(0, 0, 1)
TSW3_VARIANT(0, 1, 0)
TSW3_VARIANT(1, 0, 0)
TSW3_VARIANT(1, 0, 1)
TSW3_VARIANT(1, 1, 0)
TSW3_VARIANT(2, 0, 0)
TSW3_VARIANT(2, 0, 1)
TSW3_VARIANT(2, 1, 0)
#endif

#ifdef TSW3_TRACE
// timing-instrumented single-variant builds only (tools/tsw_trace.py): six s_memtime stamps per wave and step
__device__ char* g_tsw3_trace = nullptr;
constexpr size_t TSW3_TRACE_WG_BYTES = 1024 * 8 * 32;
#endif

template <int NORM, int SPARSE, int HIN>
__global__ __launch_bounds__(NT, 2) void cspn2d_tsw3_kernel(const float* __restrict__ gd, const float* __restrict__ blur,
                                                             const float* __restrict__ hin, const float* __restrict__ sparse,
                                                             float* __restrict__ out, const PlanGeo g, int W4, int HW4) {
    __shared__ __attribute__((aligned(16))) char lds[TSW3_LDS_BYTES];
    int bi, r0, r1;
    if (!tsw_wg_share(g, blockIdx.x, bi, r0, r1)) return;
    // boundary-row buffers, row slots and slot headers start out as zeros (the first steps read them before anyone wrote)
    for (int i = threadIdx.x; i < TSW3_LDS_TAB / 16; i += NT) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
    // the workgroup plans its own stream: 512 threads write the row descriptors straight into the LDS table
    unsigned* tab = reinterpret_cast<unsigned*>(lds + TSW3_LDS_TAB);
    const int yb = ybits_of(g.H);
    int Q = 0, Qe;
    (void)tsw3_desc(g, r0, r1, -1, yb, &Q);   // every thread: the stream length
    for (int e = threadIdx.x; e < g.stride; e += NT) tab[e] = tsw3_desc(g, r0, r1, e - TSW3_PADF, yb, &Qe);
    int p0, lo, hi;
    band_of(g, bi, p0, lo, hi);   // one band per workgroup: its columns are per-workgroup constants
    const int last = Q > 0 ? 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + g.n_iter : -1;
    const int lohi = (lo - p0) | ((hi - p0) << 16);
    const int geom = yb | ((p0 == 0) << G3_FIRST) | ((p0 + BW == g.W) << G3_LAST);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsb = (unsigned)(uintptr_t)lds;  // LDS address of the block (0 unless the compiler adds more shared data)
#ifdef TSW3_TRACE
    const void* aux = (const void*)(g_tsw3_trace + (size_t)blockIdx.x * TSW3_TRACE_WG_BYTES);
#else
    const void* aux = nullptr;
#endif
    Tsw3Asm<NORM, SPARSE, HIN>::run(lane, gd, blur, hin, sparse, out, aux, W4, HW4, __builtin_amdgcn_readfirstlane(last), wv, ldsb,
                                    __builtin_amdgcn_readfirstlane(geom), __builtin_amdgcn_readfirstlane(lohi),
                                    __builtin_amdgcn_readfirstlane(4 * p0));
}

PlanGeo make_geo3(int B, int H, int W) { return tswplan::make_geo(B, H, W, TSW3_PADF, TSW3_PADB, TSW3_TAB_MAX_ROWS); }

template <int NORM>
void launch3(int mode, const PlanGeo& g, hipStream_t st, const float* gd, const float* blur, const float* hin, const float* sparse,
             float* out) {
    const int W4 = 4 * g.W, HW4 = 4 * g.H * g.W;
    if (mode == 1)
        hipLaunchKernelGGL((cspn2d_tsw3_kernel<NORM, 1, 0>), dim3(g.n_wg), dim3(NT), 0, st, gd, blur, hin, sparse, out, g, W4, HW4);
    else if (mode == 2)
        hipLaunchKernelGGL((cspn2d_tsw3_kernel<NORM, 0, 1>), dim3(g.n_wg), dim3(NT), 0, st, gd, blur, hin, sparse, out, g, W4, HW4);
    else
        hipLaunchKernelGGL((cspn2d_tsw3_kernel<NORM, 0, 0>), dim3(g.n_wg), dim3(NT), 0, st, gd, blur, hin, sparse, out, g, W4, HW4);
}

}  // namespace

// A pass of exactly 24 iterations over images at least one band wide; byte offsets inside a tensor must fit 32 bits.
// Not for a continuation pass (hin != blur) with a sparse mask: that combination needs 11 raw planes per row slot and stays on
// the round-2 loop.
bool tsw3_supported(int B, int H, int W, bool sparse, bool hin_differs) {
    if (!tsw2d_supported(B, H, W)) return false;
    if (sparse && hin_differs) return false;
    if (((long long)B << ybits_of(H)) >= (1ll << 28)) return false;   // image | row in 28 descriptor bits
    // Round 4: an experiment outside the default build (make EXPERIMENTS=1), reached only through the hook library's
    // cspn_debug_forward2d_plan(..., plan_mode 3).  Measured on MI355X (profiles/r03_strong_scaling_shapes_1gpu.txt): with long row
    // streams per workgroup it ties with the product loop (KITTI x 64: 0.287 vs 0.288 ms); with short ones its longer prologue
    // (LDS-DMA priming) and deeper pipeline cost 4 .. 8 % (KITTI x 8: 0.0706 vs 0.0660 ms).
    return true;
}

int tsw3_pass(const float* gd, const float* blur, const float* hin, const float* sparse, float* out, int B, int H, int W,
              int norm, hipStream_t st) {
    const PlanGeo g = make_geo3(B, H, W);
    const int mode = sparse ? 1 : (hin != blur ? 2 : 0);
    switch (norm) {
        case 0: launch3<0>(mode, g, st, gd, blur, hin, sparse, out); break;
        case 1: launch3<1>(mode, g, st, gd, blur, hin, sparse, out); break;
        default: launch3<2>(mode, g, st, gd, blur, hin, sparse, out); break;
    }
    return check_launch("cspn2d_tsw3_kernel");
}

// test hooks: the planning arithmetic / the descriptor tables every workgroup would build for itself (tools/tswgen/plan3.py)
extern "C" int cspn_debug_tsw3_plan_geo(int B, int H, int W, int* n_wg, int* stride) {
    const PlanGeo g = make_geo3(B, H, W);
    *n_wg = g.n_wg;
    *stride = g.stride;
    return g.xcd ? (g.gpx | (g.extra << 8) | (g.per_xcd << 16)) : 0;
}

extern "C" int cspn_debug_tsw3_dump_plan(int B, int H, int W, void* hdr, void* tab, void* stream) {
    const PlanGeo g = make_geo3(B, H, W);
    const int n = g.n_wg * g.stride;
    hipLaunchKernelGGL(cspn2d_plan3_dump_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int4*)hdr, (unsigned*)tab, g);
    return check_launch("cspn2d_plan3_dump_kernel");
}

#ifdef TSW3_TRACE
extern "C" int cspn_debug_tsw3_set_trace(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tsw3_trace), &p, sizeof(p));
}
#endif

}  // namespace cspn
