// cspn3d_persistent.hip -- n_iter chained 3x3x3 propagation steps (reference cspn_paddle/demo.py:41-43,50-52: one
// fluid.layers.affinity_propagate call per step, gates normalised by the caller and used as given) with the 26 gates of a
// voxel read from HBM ONCE per forward instead of once per step.
//
// Why a persistent kernel: a step needs all 26 gates of every voxel (104 B/voxel); fusing steps means keeping them on chip.
// One CU holds 4096 voxels' gates in its register file (512 threads x 8 voxels x 26 floats = 416 KB of the 512 KB), the
// whole device about one million: a "chunk" -- an x-slab of the batch's volumes standing side by side along x (four
// never-valid columns between two volumes, so nothing flows across), cut with n_iter halo voxels on each interior side --
// is loaded once, all n_iter steps run on it while the gates stay in registers, then the next chunk is taken.  Between steps
// only the VALUES move: a thread of a boundary row publishes its new values as 16-byte quads = three values + the step tag
// (memory-side sc1 stores), a reader polls the quads it needs until the tag is the step's and drops the values into its LDS
// halo -- no store acknowledgement, no flags, no barrier between tiles or chunks (the first version of this
// file exchanged whole levels through scratch volumes with per-tile flags: profiles/r02_vol3d_kernel_stats.md history).
// Garbage from the cut faces of a chunk travels one voxel per step and stays inside the halo.  HBM traffic = 112 B/voxel x
// (1 + 2*n_iter / owned x-extent) once, independent of n_iter, plus the exchanged quads.
//
// Takes W % 4 == 0, W >= 64, 16-byte aligned tensors, 2 <= n_iter <= 60: the Paddle contract (norm NONE, no mask) directly, the
// normalising / masked modes after fold3d_kernel (HASC), and the transposed operator of the backward (ADJ).  Everything else
// runs cspn3d_stepwise.hip.  Parity unpinned (the Paddle op's source is not in the reference tree), checked against oracle/.
#include <cstddef>
#include <mutex>

#include "cspn_common.h"

// P3_ROWS_PLAIN / P3_ROWS_CF / P3_NO_PRIO / P3_LYP: A/B builds of the round-5 row assignment (correct results, tools/r05/build_p3var.sh)
#if (defined(P3_EXP_NOPOLL) || defined(P3_EXP_NOWAIT) || defined(P3_EXP_LANE_REMAP) || defined(P3_EXP_NT) || defined(P3_PRESLEEP)) && \
    !defined(P3_EXPERIMENT_BUILD)
#error "P3_EXP_* switch timing variants that give WRONG RESULTS: tools/build_p3var.sh defines P3_EXPERIMENT_BUILD for them"
#endif

namespace cspn {
namespace {

// XG = threads (groups of 8 consecutive x) per tile row.  8: one workgroup of 512 threads per CU, tile 8 x 8 x 64.  4 (-DP3_XG=4):
// two workgroups of 256 threads per CU, tile 8 x 8 x 32 -- one's exchange round trip under the other's arithmetic.
#ifndef P3_XG
#define P3_XG 8
#endif
#ifndef P3_BACKOFF
#define P3_BACKOFF 0   // s_sleep units (64 cycles) between two polls of quads that were not there yet
#endif
constexpr int XG = P3_XG, XGS = XG == 8 ? 3 : 2, WG_PER_CU = 8 / XG;
static_assert(XG == 8 || XG == 4, "x-groups per tile row");
constexpr int TZ = 8, TY = 8, TX = 8 * XG;       // 8 consecutive x per thread
constexpr int NTP = 64 * XG;                      // 256 registers per thread, 208 of them gates
constexpr int LZ = TZ + 2, LY = TY + 2, LXU = TX + 2, LX = TX + 4;   // LDS tile with halo, rows padded to a multiple of 4 floats
#ifndef P3_LYP
#define P3_LYP 10
#endif
constexpr int LYP = P3_LYP;                      // rows per z plane of the LDS tile as LAID OUT (>= LY; 12 with -DP3_ROWS_CF: the rows
static_assert(LYP >= LY, "plane pitch");         // (lz, ly) and (lz + 4, ly) are then 64 banks apart)
constexpr int LTILE = LZ * LYP * LX;             // floats per level buffer
constexpr unsigned SPIN_MAX = 1u << 18;
constexpr unsigned CAPTURED_SEQ = 0xffffffffu;   // what a launch captured into a graph stores in the status word (see persistent3d_launch)
constexpr int MAX_WG = 256 * WG_PER_CU;
// layout of the sync words behind the exchange buffers: [0, MAX_WG + 64 * 9) flags of the first version, the error word, [1024, 1024 + MAX_WG) the XCC table,
// [2048, ..) the trace stamps -- an A/B build with more workgroups per CU must not let them overlap
static_assert(MAX_WG + 64 * 9 + 1 <= 1024 && 1024 + MAX_WG <= 2048, "sync-word layout: error word / XCC table / trace stamps overlap");

struct Geo3 {
    int B, D, H, W, n_iter, halo;
    int tz, ty, cx;          // tiles along z, y (whole extent) and along x per chunk
    int S;                   // owned x-extent of a chunk
    int nchunk;              // chunks per launch: the volumes stand side by side along x, `pitch` columns apart (>= W + 4: the
    int pitch, vw;           // columns between them are never inside a volume, so nothing flows across), vw = B pitch - 4
    int n_wg;                // tiles = tz * ty * cx (the exchange buffers and the XCC table are indexed by tile)
    int n_launch;            // workgroups launched: n_wg, or 8 * (tiles per block) with the XCD-aware placement
    int bz, by, bx;          // XCD-aware placement (round 5): the tile grid is cut into <= 8 blocks of bz x by x bx tiles, block k is
    int nby, nbx;            // given to the workgroups with id % 8 == k (ids 8 apart share an XCD: observed, never relied on); bz = 0: off
    int lv0, lvs;            // level output: step it (< n_iter) goes to volume lv0 + it * lvs of `levels`
    int wt;                  // != 0: every published row goes write-through (no L2-resident stores even where all readers share the XCD): A/B and tests
    int mute;                // MUTE instantiation (test-hook library) only: the workgroup that never publishes
    int C;                   // MULTI instantiation only: value channels that share the gates (feat / out are [B][C][V])
    unsigned seq;            // number of this launch on its device (> 0): what a workgroup that gives up writes to *status
    unsigned* status;        // the device's sticky status word (host-mapped, device-visible address)
    long long gps, gbs;      // gate plane / batch stride in floats: [B][26][V] as given (V, 26 V) or folded planes [26][B][V] (B V, V)
};

constexpr int GEO3_KERNARG_OFFSET = 7 * 8;   // cspn3d_persistent_kernel(7 pointers, Geo3): where g starts in the kernel-argument segment
static_assert(alignof(Geo3) == 8 && offsetof(Geo3, status) % 8 == 0, "Geo3 layout (the timeout path reads seq / status by offset)");

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16_sc1(float* p, float4 v) {
    const v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}

__device__ __forceinline__ void st16_l2(float* p, float4 v) {   // no scope bits: the line stays (dirty) in this XCD's L2
    const v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(x) : "memory");
}

// published boundary of a tile, in 16-byte quads = up to three values + the step tag in the fourth word.  Thread xg
// of row (lz, ly) owns the quads q3 = 0, 1, 2 of its values (0,1,2) (3,4,5) (6,7), stored at [q3][row][xg] so that a wave's store (and
// a reader's load) covers whole 128-byte lines; written for boundary rows only.  The x faces (value 0 of xg = 0, value 7 of xg = 7)
// of ALL rows follow at [side][row], one value per quad.
constexpr int QROW = 3 * XG, NROWS = TZ * TY, NQA = NROWS * QROW, NQ = NQA + 2 * NROWS;   // 1536 + 128 quads per tile and level parity
// what a tile fetches per step: 36 halo rows (above / below / beside in y) of 24 quads, and the 200 voxels beside it in x
constexpr int NHROW = 2 * LY + 2 * TZ, NHQ = NHROW * QROW, NSGL = 2 * LZ * LY, NIT = NHQ + NSGL, NSLOT = (NIT + NTP - 1) / NTP;

// Which row of the tile a thread owns.  A wave = 8 rows x 8 threads (8 voxels along x each).  Round 5 tried three assignments, all
// bit-identical, all within 1 % of each other at config 5 (profiles/r05_vol3d_rows_first_ab.md: a step is the cross-XCD trip of the
// publications, not its arithmetic phase):
//   default            the 28 boundary rows of the 8 x 8 rows first -- rows 0..7 plane lz = 0, 8..15 plane lz = TZ - 1, 16..27 the rows
//                      ly = 0 / TY - 1 of the planes between, 28..63 the 36 interior rows: waves 0..2 publish whole rows without
//                      exec-masked halves (-0.6 % against the plain assignment, the best of the three by a hair);
//   -DP3_ROWS_PLAIN    a wave = one z plane (rounds 2-4);
//   -DP3_ROWS_CF -DP3_LYP=12   rows (lz, ly) / (lz + 4, ly) / (lz, ly + 1) / (lz + 4, ly + 1) per group of four: the four quarter-rows
//                      a ds_read_b128 lane group touches ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) then tile the 64 banks
//                      -- 160 -> 96 LDS cycles per neighbour row and workgroup on paper (tools/r05/lds_conflicts.py), -0.4 % measured:
//                      the arithmetic phase is not bound by LDS conflicts either.
// [q3][row][xg] in the exchange buffers is indexed by the LOGICAL row lz * TY + ly: readers see no difference.
__device__ __forceinline__ void row_of(int t, int& lx, int& ly, int& lz) {
    lx = (t & (XG - 1)) * 8;
    const int r = t >> XGS;   // 0 .. 63
#if defined(P3_ROWS_PLAIN)
    ly = r & 7;
    lz = r >> 3;
#elif defined(P3_ROWS_CF)
    const int g = r >> 2, k = r & 3;   // group of four rows, row in the group
    lz = (g & 3) + 4 * (k & 1);
    ly = 2 * (g >> 2) + (k >> 1);
#else
    const int i = r - 16, j = r - 28, j6 = (j * 43) >> 8;   // j / 6 for 0 <= j < 36
    lz = r < 8 ? 0 : r < 16 ? TZ - 1 : r < 28 ? 1 + (i >> 1) : 1 + j6;
    ly = r < 8 ? r : r < 16 ? r - 8 : r < 28 ? (i & 1) * (TY - 1) : 1 + j - 6 * j6;
#endif
}
static_assert(TZ == 8 && TY == 8, "row_of enumerates the boundary rows of an 8 x 8 tile cross-section");

__device__ __forceinline__ unsigned lds_addr(const void* p) {   // byte address inside the workgroup's LDS
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ v4f ldq_sc1(const float4* base, unsigned byte_off) {   // uniform base + per-lane 32-bit offset
    v4f v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
    return v;
}

// ADJ: the transposed operator A_t(q) = sum_k g_k(q - off_k) A_{t+1}(q - off_k) of the backward, i.e. the same propagation with the
// gates w_j(q) = g_{opp(j)}(q + off_j) (off_{opp(j)} = -off_j): only the chunk prologue differs, it reads plane opp(j) shifted by
// off_j (4-byte aligned 16-byte loads; a quad that sticks out of the volume by its first / last element is read aligned and
// shifted in the registers).  levels != nullptr: every step but the last also stores its owned voxels (the level history
// the gate gradient multiplies with).
// HASC: a constant term per voxel, H_{t+1} = c' + sum_k w'_k H_t(p + off_k): the folded form of the normalising / masked modes
// (fold3d_kernel of cspn3d_stepwise.hip writes w' and c'); c' of the thread's eight voxels waits in LDS between the steps.
// Gate quads of the NEXT chunk parked in the LDS the level buffers leave free (round 3): during the first NPRE / 2 steps of a
// chunk each thread requests both quads of one gate plane of its voxels in the next chunk by LDS-DMA (global_load_lds_dwordx4:
// no destination register -- the 208 gate registers leave none; M0 carries the LDS address, all 160 KB are reachable on gfx950,
// profiles/r03_ubench_dma.txt), while HBM is otherwise idle; the next prologue reads them back with ds_read_b128 and requests
// 52 - NPRE quads from memory instead of 52.  A wave reads only what it requested itself (its vmcnt wait makes the data visible
// to it: no barrier), laid out [quad][wave][lane] so that one request fills 1 KiB.
template <bool ADJ, bool HASC>
struct Pre3 { static constexpr int N = ADJ ? 0 : (XG == 8 ? (HASC ? 10 : 12) : 12); };

__device__ __forceinline__ void lds_dma16(unsigned byte_off, const float* base, unsigned lds_dst) {   // lds_dst: wave-uniform
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory");
}

// MUTE (test hook): workgroup g.mute computes but never publishes, so that its neighbours run into the poll timeout
// MULTI (round 4): feat / out hold C value channels per volume that share the gates (reference cspn_paddle/README.md:56: "gate_weight
// would be shared in the channel dimension for input when C>1"): the gates of a chunk are loaded ONCE and stay in the registers
// while the n_iter steps are run for one channel after the other (only level 0 is re-read per channel, 4 B/voxel against 104)
template <bool ADJ, bool HASC, bool MUTE = false, bool MULTI = false>
__global__ __launch_bounds__(NTP) __attribute__((amdgpu_waves_per_eu(2, 2))) void cspn3d_persistent_kernel(const float* __restrict__ gate, const float* __restrict__ feat,
                                                                 const float* __restrict__ cprime, float* __restrict__ out,
                                                                 float* __restrict__ levels, float* __restrict__ scratch,
                                                                 unsigned* __restrict__ sync, Geo3 g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * LTILE];
    __shared__ __attribute__((aligned(16))) float4 s_c[HASC ? 2 * NTP : 1];   // c' of the thread's two quads, [quad][thread]
    constexpr int NPRE = Pre3<ADJ, HASC>::N;
    __shared__ __attribute__((aligned(16))) float4 s_pre[NPRE ? NPRE * NTP : 1];   // parked gate quads, [quad][thread] (addressed by hand)
    __shared__ int s_bail;
    unsigned* err = sync + MAX_WG + 64 * 9;  // [1] (the words in front of it belonged to the flag exchange of the first version)
#ifdef P3_TRACE
    unsigned long long* trc = reinterpret_cast<unsigned long long*>(sync + 2048);   // [n_iter][6] stamps of workgroup 37, chunk 1
#define P3_STAMP(k) if (wg == 37 && tid == 0 && round == 1) trc[(it - 1) * 6 + (k)] = __builtin_readcyclecounter()
#define P3_CHUNK(k) if (wg == 37 && tid == 0 && round == 1) trc[g.n_iter * 6 + (k)] = __builtin_readcyclecounter()
#else
#define P3_STAMP(k)
#define P3_CHUNK(k)
#endif
    __shared__ unsigned long long s_rowl2;   // bit (lz * TY + ly): every reader of that boundary row's quads runs on THIS XCD
    __shared__ unsigned s_xl2;               // bit (side * TZ + lz): the same for the line of x-face quads of plane lz
    const int tid = threadIdx.x;
    const size_t HW = (size_t)g.H * g.W, V = (size_t)g.D * HW, total = (size_t)g.B * V;
    const int nch = MULTI ? g.C : 1;
    const int FBS = MULTI ? g.C * (int)V : (int)V;   // volume stride of the value tensors in floats ([B][C][V])
    float4* X = reinterpret_cast<float4*>(scratch + 2 * total);   // [2][n_wg][NQ] published boundaries
    // ---- which tile this workgroup owns.  XCD-aware placement (round 5): workgroup ids 8 apart have so far always shared an XCD
    // (MI355X_MICROARCH.md: "observed, for speed only"), so block k of the tile grid goes to the ids = k mod 8 -- a tile's
    // neighbours are then mostly on its own XCD and their exchange stays in that XCD's L2.  Nothing below RELIES on the
    // placement: every workgroup publishes the XCC it really runs on, and a row is stored L2-resident only if every workgroup
    // that reads it has published the same XCC (s_rowl2 / s_xl2); everything else goes write-through as before.
    int ix, iy, iz;
    bool have_tile;
    {
        const int b = blockIdx.x;
        if (g.bz > 0) {
            const int k = b & 7, s = b >> 3, bsz = g.bz * g.by * g.bx;
            const int kx = k % g.nbx, ky = (k / g.nbx) % g.nby, kz = k / (g.nbx * g.nby);
            const int sx = s % g.bx, sy = (s / g.bx) % g.by, sz = s / (g.bx * g.by);
            ix = kx * g.bx + sx; iy = ky * g.by + sy; iz = kz * g.bz + sz;
            have_tile = s < bsz && iz < g.tz;   // (blocks beyond the grid: fewer than 8 blocks)
        } else {
            ix = b % g.cx; iy = (b / g.cx) % g.ty; iz = b / (g.cx * g.ty);
            have_tile = b < g.tz * g.ty * g.cx;
        }
    }
    const int wg = have_tile ? (iz * g.ty + iy) * g.cx + ix : 0;   // the TILE's number: exchange buffers, XCC table, mute
    unsigned* xcctab = sync + 1024;   // [n_wg] 1 + XCC id of the workgroup that owns the tile (cleared by the launch's memset)
    if (tid == 0) {
        s_bail = 0;
        s_rowl2 = 0ull;
        s_xl2 = 0u;
        if (have_tile) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            __hip_atomic_store(xcctab + wg, 1u + (xcc & 0xfu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (sc1: write-through)
        }
    }
    // byte offsets of a thread's two gate quads in chunk cc (what the chunk prologue below calls voff0 / voff1)
    auto gate_offs = [&](int cc, int tc, unsigned& v0, unsigned& v1) {
        const int x0n = cc * g.S - g.halo + ix * TX;
        const int bfn = __builtin_amdgcn_readfirstlane(x0n > 0 ? x0n / g.pitch : 0);
        int lxn, lyn, lzn;
        row_of(tc, lxn, lyn, lzn);
        const int zn = iz * TZ + lzn, yn = iy * TY + lyn;
        const int xr0 = x0n + lxn - bfn * g.pitch, xr1 = xr0 + 4;
        const int bq0 = bfn + (xr0 >= g.pitch ? 1 : 0), xq0n = xr0 >= g.pitch ? xr0 - g.pitch : xr0;
        const int bq1 = bfn + (xr1 >= g.pitch ? 1 : 0), xq1n = xr1 >= g.pitch ? xr1 - g.pitch : xr1;
        const bool inzy = zn < g.D && yn < g.H;
        const int rown = (zn * g.H + yn) * g.W;
        v0 = inzy && xq0n >= 0 && xq0n + 3 < g.W && bq0 < g.B ? (unsigned)(bq0 * (int)g.gbs + rown + xq0n) * 4u : 0u;
        v1 = inzy && xq1n >= 0 && xq1n + 3 < g.W && bq1 < g.B ? (unsigned)(bq1 * (int)g.gbs + rown + xq1n) * 4u : 0u;
    };
    auto park_gate = [&](int cc, int k, int tc) {   // both quads of gate plane k of chunk cc -> s_pre[2k], s_pre[2k + 1]
        unsigned v0, v1;
        gate_offs(cc, tc, v0, v1);
        const float* gk = gate + (size_t)k * g.gps;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(s_pre) + (unsigned)((2 * k * NTP + (tc & ~63)) * 16));
        lds_dma16(v0, gk, dst);
        lds_dma16(v1, gk, dst + NTP * 16);
    };
    if (NPRE && have_tile && g.nchunk > 0) {   // the first chunk's share, so that every prologue finds its parked quads
        int tc = tid;
        asm volatile("" : "+v"(tc));
#pragma unroll 1
        for (int k = 0; k < NPRE / 2; ++k) park_gate(0, k, tc);
    }
    __syncthreads();   // (s_rowl2 / s_xl2 / s_bail initialised)
    if (have_tile && tid < 64) {
        // Thread r < 64 = boundary row (lz, ly) = (r / TY, r % TY): are all tiles that read this row's quads on my XCD?
        // Readers: the tiles at (dz, dy) != (0, 0) with dz in {0, -1 if lz == 0, +1 if lz == TZ - 1}, dy likewise, dx = 0.
        // A neighbour's XCC is polled a bounded number of times (all workgroups of a grid start within a microsecond of each
        // other and this runs under the first chunk's LDS-DMA requests); if it does not show, the row simply stays
        // write-through.  Done HERE, before any gate register is live: inside the chunk loop it cost ten spilled registers.
        unsigned mine_x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(mine_x));
        mine_x = 1u + (mine_x & 0xfu);
        auto same_xcd = [&](int dz, int dy, int dx) -> bool {
            const int tz2 = iz + dz, ty2 = iy + dy, tx2 = ix + dx;
            if (tz2 < 0 || tz2 >= g.tz || ty2 < 0 || ty2 >= g.ty || tx2 < 0 || tx2 >= g.cx) return true;   // nobody there
            const unsigned* pe = xcctab + (tz2 * g.ty + ty2) * g.cx + tx2;
            unsigned v = 0;
            for (int t = 0; t < 2048 && v == 0; ++t) {
                v = __hip_atomic_load(pe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == 0) __builtin_amdgcn_s_sleep(8);
            }
            return v == mine_x;
        };
        const int rz = tid >> 3, ry = tid & 7;
        bool rl = true;
        for (int dz = (rz == 0 ? -1 : 0); dz <= (rz == TZ - 1 ? 1 : 0); ++dz)
            for (int dy = (ry == 0 ? -1 : 0); dy <= (ry == TY - 1 ? 1 : 0); ++dy)
                if (dz || dy) rl = rl && same_xcd(dz, dy, 0);
        if (rl && !g.wt) atomicOr(&s_rowl2, 1ull << tid);
        if (tid < 2 * TZ) {
            // the 128-byte line of x-face quads of plane lz = tid % TZ, side = tid / TZ: read by the tiles at dx = -1 (side 0)
            // / +1 (side 1), dy in {-1, 0, 1} (the line holds all eight rows of the plane), dz as above
            const int pz = tid % TZ, dx = tid < TZ ? -1 : 1;
            bool xl = true;
            for (int dz = (pz == 0 ? -1 : 0); dz <= (pz == TZ - 1 ? 1 : 0); ++dz)
                for (int dy = -1; dy <= 1; ++dy) xl = xl && same_xcd(dz, dy, dx);
            if (xl && !g.wt) atomicOr(&s_xl2, 1u << tid);
        }
    }
    __syncthreads();   // s_rowl2 / s_xl2 are complete (and s_bail initialised) before anybody publishes
    {
        for (int c = 0; c < g.nchunk; ++c) {
            const unsigned round = (unsigned)c;
            if (have_tile) {
                // x coordinates below are positions on the row of volumes (volume b starts at b * pitch)
                const int ox0 = c * g.S, ox1 = min(g.vw, ox0 + g.S);    // owned columns of the chunk
                const int wx0 = ox0 - g.halo;                            // window start (may be negative)
                const int z0 = iz * TZ, y0 = iy * TY, x0 = wx0 + ix * TX;
                // position -> (volume, column in it): pitch >= TX + 2, so a tile and its shell lie across one volume boundary
                // at most -- one scalar division per chunk, a compare per lane
                const int bf = __builtin_amdgcn_readfirstlane(x0 > 0 ? x0 / g.pitch : 0);   // (keeps it in a scalar register)
                const int xbf = bf * g.pitch;
                auto loc = [&](int xv, int& bb, int& xx) {
                    const int xr = xv - xbf;
                    const bool wr = xr >= g.pitch;
                    bb = bf + (wr ? 1 : 0);
                    xx = wr ? xr - g.pitch : xr;
                };
                if (c > 0) {
                    // the last step of the chunk before read the level buffers without a barrier behind it, and this prologue writes
                    // level 0 into both: every wave must be through with those reads first (their results were consumed, so they
                    // are complete; nothing else is in flight that the barrier would have to wait for)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (NPRE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's parked gate quads have landed
                P3_CHUNK(0);
                // ---- level 0 first (its loads are issued ahead of the gates' so that they return first): the thread's own eight
                // voxels and its share of the 2504-voxel halo shell (outside the volume: 0, for good; outside the chunk window:
                // the level-0 value, any finite number will do there).
                // All of these loads are issued by hand (uniform base + 32-bit lane offset, lanes outside the volume clamped to offset
                // 0 and zeroed afterwards) so that the count in flight is known: 2 + NSHT level-0 loads, then 52 gate loads
                // (59 <= the 63 vmcnt can count), level 0 goes to LDS while the gates are still arriving, and nothing is spilled
                // in between (a spill reload would queue behind every load in flight).  The s_nop in front of each load covers the
                // "VALU wrote the SGPR base (v_readlane of a spilled SGPR) -> VMEM reads it" hazard, which the compiler cannot see
                // inside inline assembly.
                constexpr int SH_Z = 2 * LY * LXU, SH_Y = 2 * TZ * LXU, SH_X = 2 * TZ * TY, NSH = SH_Z + SH_Y + SH_X;
                constexpr int NSHT = (NSH + NTP - 1) / NTP;
                int tc = tid;   // opaque per chunk: nothing below may be hoisted out of the chunk loop (it would be spilled, and a
                asm volatile("" : "+v"(tc));   // spill reload waits for every load in flight)
                int lx, ly, lz;
                row_of(tc, lx, ly, lz);
                const int z = z0 + lz, y = y0 + ly;
                int b0, xq0, b1, xq1;     // the thread's two quads: volume and first column
                loc(x0 + lx, b0, xq0);
                loc(x0 + lx + 4, b1, xq1);
                const bool in_zy = z < g.D && y < g.H;
                const bool in0 = in_zy && xq0 >= 0 && xq0 + 3 < g.W && b0 < g.B, in1 = in_zy && xq1 >= 0 && xq1 + 3 < g.W && b1 < g.B;
                const int row = (z * g.H + y) * g.W;
                // byte offsets: 1-channel tensors (feat, c') and the gate tensor (volume stride gbs floats)
                const unsigned fo0 = in0 ? (unsigned)(b0 * FBS + row + xq0) * 4u : 0u, fo1 = in1 ? (unsigned)(b1 * FBS + row + xq1) * 4u : 0u;
                const unsigned go0 = (unsigned)(b0 * (int)g.gbs + row + xq0) * 4u, go1 = (unsigned)(b1 * (int)g.gbs + row + xq1) * 4u;
                const unsigned voff0 = in0 ? go0 : 0u, voff1 = in1 ? go1 : 0u;
                auto shell_pos = [&](int i, int& pz, int& py, int& px) {
                    if (i < SH_Z) { pz = i < LY * LXU ? 0 : LZ - 1; const int r = i < LY * LXU ? i : i - LY * LXU; py = r / LXU; px = r - py * LXU; }
                    else if (i < SH_Z + SH_Y) { const int u = i - SH_Z, r = u / LXU; px = u - r * LXU; pz = 1 + (r >> 1); py = (r & 1) ? LY - 1 : 0; }
                    else { const int u = i - SH_Z - SH_Y, r = u >> 1; px = (u & 1) ? LXU - 1 : 0; pz = 1 + r / TY; py = 1 + r % TY; }
                };
                v4f f0, f1;
                float fs[NSHT];
                const float* fb = feat;
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(f0) : "v"(fo0), "s"(fb) : "memory");
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(f1) : "v"(fo1), "s"(fb) : "memory");
#pragma unroll
                for (int j = 0; j < NSHT; ++j) {
                    const int i = tc + j * NTP;
                    int pz, py, px;
                    shell_pos(i, pz, py, px);
                    const int vz = z0 + pz - 1, vy = y0 + py - 1;
                    int vb, vx;
                    loc(x0 + px - 1, vb, vx);
                    const bool ok = i < NSH && vz >= 0 && vz < g.D && vy >= 0 && vy < g.H && vx >= 0 && vx < g.W && vb < g.B;
                    const unsigned so = ok ? (unsigned)(vb * FBS + (vz * g.H + vy) * g.W + vx) * 4u : 0u;
                    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(fs[j]) : "v"(so), "s"(fb) : "memory");
                }
                v4f cq0, cq1;
                if (HASC) {
                    const float* cb = cprime;
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(cq0) : "v"(fo0), "s"(cb) : "memory");
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(cq1) : "v"(fo1), "s"(cb) : "memory");
                }
                // ---- the 26 gates of the thread's eight voxels: read once, kept in registers for all steps
                v4f w[26][2];
                // the 26 plane bases are worked out per chunk from an opaque copy of the plane stride: computed once per kernel they
                // were 52 scalar registers live across everything (all of them spilled to VGPR lanes, which cost the vector
                // registers the gates need: round 3 had 50 .. 78 SGPR spills and a VGPR spill per variant)
                long long gps_c = g.gps;
                asm volatile("" : "+s"(gps_c));
                int Hc = g.H, Wc = g.W;   // (ADJ: the 26 row shifts, the same way)
                if (ADJ) asm volatile("" : "+s"(Hc), "+s"(Wc));
#pragma unroll
                for (int k = 0; k < 26; ++k) {
                    const float* gk = gate + (size_t)k * (size_t)gps_c;
                    if (ADJ) {
                        const int c27 = k < 13 ? k : k + 1, dz = 1 - c27 / 9, dy = 1 - (c27 / 3) % 3, dx = 1 - c27 % 3;
                        const int ko = (26 - c27) < 13 ? (26 - c27) : (26 - c27) - 1;   // the plane of the opposite offset
                        const float* gko = gate + (size_t)ko * (size_t)gps_c;
                        const bool rowok = z + dz >= 0 && z + dz < g.D && y + dy >= 0 && y + dy < g.H;
                        const int sh = ((dz * Hc + dy) * Wc) * 4;
                        // first / last element outside the volume: read the aligned quad, shift afterwards
                        const int e0 = (dx < 0 && xq0 == 0) || (dx > 0 && xq0 + 4 == g.W) ? 0 : dx * 4;
                        const int e1 = (dx < 0 && xq1 == 0) || (dx > 0 && xq1 + 4 == g.W) ? 0 : dx * 4;
                        const unsigned a0 = in0 && rowok ? (unsigned)((int)go0 + sh + e0) : 0u;
                        const unsigned a1 = in1 && rowok ? (unsigned)((int)go1 + sh + e1) : 0u;
                        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][0]) : "v"(a0), "s"(gko) : "memory");
                        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][1]) : "v"(a1), "s"(gko) : "memory");
                    } else {
#if defined(P3_EXP_NT)   // timing experiments (profiles/r02_perf_notes.md): the nontemporal hint costs 10 % of the forward
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=&v"(w[k][0]) : "v"(voff0), "s"(gk) : "memory");
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=&v"(w[k][1]) : "v"(voff1), "s"(gk) : "memory");
#elif defined(P3_EXP_LANE_REMAP)   // WRONG RESULTS: each instruction reads 128 contiguous bytes per row
                    const int xa = x0 + (tc & (XG - 1)) * 4;   // (single volume only)
                    const unsigned ra = (unsigned)((z * g.H + y) * g.W + xa) * 4u;
                    const unsigned r0_ = (in_zy && xa >= 0 && xa + 3 < g.W) ? ra : 0u, r1_ = (in_zy && xa + 32 >= 0 && xa + 35 < g.W) ? ra + 128u : 0u;
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][0]) : "v"(r0_), "s"(gk) : "memory");
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][1]) : "v"(r1_), "s"(gk) : "memory");
#else
                    if (2 * k < NPRE) {   // parked during the chunk before (chunk 0: in front of the loop)
                        const unsigned pa = lds_addr(s_pre) + (unsigned)tc * 16u + (unsigned)(2 * k * NTP * 16), pb = pa + NTP * 16;
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(w[k][0]) : "v"(pa) : "memory");
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(w[k][1]) : "v"(pb) : "memory");
                    } else {
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][0]) : "v"(voff0), "s"(gk) : "memory");
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(w[k][1]) : "v"(voff1), "s"(gk) : "memory");
                    }
#endif
                    }
                }
                {   // level 0 into both LDS buffers once ITS loads are back (52 gate loads may still be in flight)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(52 - NPRE) : "memory");
                    if (HASC) {   // (outside the volume c' = 0: such voxels keep the value 0)
                        asm volatile("" : "+v"(cq0), "+v"(cq1));
                        const v4f zero = {0.f, 0.f, 0.f, 0.f};
                        const v4f c0 = in0 ? cq0 : zero, c1 = in1 ? cq1 : zero;
                        // (explicit 32-bit LDS addresses: with more than 64 KB of LDS the compiler's base + immediate-offset
                        // addressing went wrong, profiles/r02_perf_notes.md)
                        const unsigned ca = lds_addr(s_c) + (unsigned)tc * 16u;
                        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%3" : : "v"(ca), "v"(c0), "v"(c1), "n"(NTP * 16) : "memory");
                    }
#pragma unroll
                    for (int j = 0; j < NSHT; ++j) asm volatile("" : "+v"(fs[j]));
                    int tid_ = tid;
                    asm volatile("" : "+v"(tid_));
                    int lx, ly, lz;
                    row_of(tid_, lx, ly, lz);
                    const int z = z0 + lz, y = y0 + ly;
                    int b0, xq0, b1, xq1;
                    loc(x0 + lx, b0, xq0);
                    loc(x0 + lx + 4, b1, xq1);
                    const bool in_zy = z < g.D && y < g.H;
                    const bool in0 = in_zy && xq0 >= 0 && xq0 + 3 < g.W && b0 < g.B, in1 = in_zy && xq1 >= 0 && xq1 + 3 < g.W && b1 < g.B;
                    const int o = ((lz + 1) * LYP + (ly + 1)) * LX + lx + 1;
                    const float own8[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = (i < 4 ? in0 : in1) ? own8[i] : 0.f;
                        lds[o + i] = v;
                        lds[LTILE + o + i] = v;
                    }
#pragma unroll
                    for (int j = 0; j < NSHT; ++j) {
                        const int i = tid_ + j * NTP;
                        int pz, py, px;
                        shell_pos(i, pz, py, px);
                        const int vz = z0 + pz - 1, vy = y0 + py - 1;
                        int vb, vx;
                        loc(x0 + px - 1, vb, vx);
                        const bool ok = vz >= 0 && vz < g.D && vy >= 0 && vy < g.H && vx >= 0 && vx < g.W && vb < g.B;
                        if (i < NSH) {
                            const float v = ok ? fs[j] : 0.f;
                            lds[(pz * LYP + py) * LX + px] = v;
                            lds[LTILE + (pz * LYP + py) * LX + px] = v;
                        }
                    }
                }
                // the gates are back; outside the volume they are zero (such voxels keep the value 0)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]) : : "memory");
#pragma unroll
                for (int k = 0; k < 26; ++k) {
                    if (k) asm volatile("" : "+v"(w[k][0]), "+v"(w[k][1]));
                    const v4f zero = {0.f, 0.f, 0.f, 0.f};
                    if (ADJ) {
                        const int c27 = k < 13 ? k : k + 1, dz = 1 - c27 / 9, dy = 1 - (c27 / 3) % 3, dx = 1 - c27 % 3;
                        const bool rowok = z + dz >= 0 && z + dz < g.D && y + dy >= 0 && y + dy < g.H;
                        v4f q0 = in0 && rowok ? w[k][0] : zero, q1 = in1 && rowok ? w[k][1] : zero;
                        if (dx < 0) {   // element 0 is the voxel left of the volume
                            if (xq0 == 0) q0 = v4f{0.f, q0.x, q0.y, q0.z};
                            if (xq1 == 0) q1 = v4f{0.f, q1.x, q1.y, q1.z};
                        } else if (dx > 0) {   // element 3 is the voxel right of the volume
                            if (xq0 + 4 == g.W) q0 = v4f{q0.y, q0.z, q0.w, 0.f};
                            if (xq1 + 4 == g.W) q1 = v4f{q1.y, q1.z, q1.w, 0.f};
                        }
                        w[k][0] = q0;
                        w[k][1] = q1;
                    } else {
                        w[k][0] = in0 ? w[k][0] : zero;
                        w[k][1] = in1 ? w[k][1] : zero;
                    }
                }
                P3_CHUNK(1);
                __syncthreads();
                P3_CHUNK(2);
                for (int ch = 0; ch < nch; ++ch) {
                if (MULTI && ch > 0) {
                    // the next value channel on the same (resident) gates: level 0 of channel ch into both level buffers.  The last
                    // step of the channel before read them without a barrier behind it (as at the top of a chunk)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    const float* fb = feat + (size_t)ch * V;
                    int tid_ = tid;
                    asm volatile("" : "+v"(tid_));
                    int lx, ly, lz;
                    row_of(tid_, lx, ly, lz);
                    const int z = z0 + lz, y = y0 + ly;
                    int b0, xq0, b1, xq1;
                    loc(x0 + lx, b0, xq0);
                    loc(x0 + lx + 4, b1, xq1);
                    const bool in_zy = z < g.D && y < g.H;
                    const bool in0 = in_zy && xq0 >= 0 && xq0 + 3 < g.W && b0 < g.B, in1 = in_zy && xq1 >= 0 && xq1 + 3 < g.W && b1 < g.B;
                    const int row = (z * g.H + y) * g.W;
                    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 q0 = in0 ? *reinterpret_cast<const float4*>(fb + (unsigned)(b0 * FBS + row + xq0)) : zero4;
                    const float4 q1 = in1 ? *reinterpret_cast<const float4*>(fb + (unsigned)(b1 * FBS + row + xq1)) : zero4;
                    const int o = ((lz + 1) * LYP + (ly + 1)) * LX + lx + 1;
                    const float own8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        lds[o + i] = own8[i];
                        lds[LTILE + o + i] = own8[i];
                    }
#pragma unroll 1
                    for (int i = tid_; i < NSH; i += NTP) {
                        int pz, py, px;
                        shell_pos(i, pz, py, px);
                        const int vz = z0 + pz - 1, vy = y0 + py - 1;
                        int vb, vx;
                        loc(x0 + px - 1, vb, vx);
                        const bool ok = vz >= 0 && vz < g.D && vy >= 0 && vy < g.H && vx >= 0 && vx < g.W && vb < g.B;
                        const float v = ok ? fb[(unsigned)(vb * FBS + (vz * g.H + vy) * g.W + vx)] : 0.f;
                        lds[(pz * LYP + py) * LX + px] = v;
                        lds[LTILE + (pz * LYP + py) * LX + px] = v;
                    }
                    __syncthreads();
                }
                for (int it = 1; it <= g.n_iter; ++it) {
                    // the 208 gate registers leave no room for loop-invariant addresses: everything below is recomputed from
                    // tid_ each step (a handful of integer instructions) instead of being kept live across the loop
                    int tid_ = tid;
                    asm volatile("" : "+v"(tid_));
                    int lx, ly, lz;
                    row_of(tid_, lx, ly, lz);
                    const float* cur = lds + ((it - 1) & 1) * LTILE;
                    float* nxt = lds + (it & 1) * LTILE;
                    P3_STAMP(0);
                    if (NPRE && c + 1 < g.nchunk && (!MULTI || ch == 0)) {   // one gate plane of the next chunk per step (all of them if there are few steps)
#pragma unroll 1
                        for (int k = it - 1; k < NPRE / 2; k += g.n_iter) park_gate(c + 1, k, tid_);
                    }
#if !defined(P3_ROWS_PLAIN) && !defined(P3_ROWS_CF) && !defined(P3_NO_PRIO)
                    // waves 0..3 hold the boundary rows (row_of): raised issue priority for their arithmetic.  Measured: no effect on the
                    // time (the two builds are equal to 0.1 %, profiles/r05_vol3d_rows_first_ab.md); kept because THIS instruction stream
                    // is the one the register allocator fits without a spill in every variant (without it the HASC variant spills two)
                    if (__builtin_amdgcn_readfirstlane(tid_) < 4 * 64) __builtin_amdgcn_s_setprio(3);
#endif
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (HASC) {
                        const unsigned ca = lds_addr(s_c) + (unsigned)tid_ * 16u;
                        v4f c0, c1;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(c0), "=&v"(c1) : "v"(ca), "n"(NTP * 16) : "memory");
                        acc[0] = c0.x; acc[1] = c0.y; acc[2] = c0.z; acc[3] = c0.w;
                        acc[4] = c1.x; acc[5] = c1.y; acc[6] = c1.z; acc[7] = c1.w;
                    }
#pragma unroll
                    for (int n = 0; n < 9; ++n) {   // the 9 neighbour rows (dz, dy); three x-taps each
                        const int dz = 1 - n / 3, dy = 1 - n % 3;
                        const float* row = cur + ((lz + 1 + dz) * LYP + (ly + 1 + dy)) * LX + lx;
                        const float4 a0 = *reinterpret_cast<const float4*>(row);        // x-1 .. x+2
                        const float4 a1 = *reinterpret_cast<const float4*>(row + 4);    // x+3 .. x+6
                        const float2 e = *reinterpret_cast<const float2*>(row + 8);     // x+7, x+8
                        const float h[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, e.x, e.y};
                        asm volatile("" ::: "memory");   // one row of values in flight at a time: the gate registers need the room
#pragma unroll
                        for (int t = 0; t < 3; ++t) {   // dx = 1 - t
                            const int c27 = n * 3 + t;
                            if (c27 == 13) continue;
                            const int k = c27 < 13 ? c27 : c27 - 1;
                            const int dx = 1 - t;
                            acc[0] = fmaf(w[k][0].x, h[1 + dx], acc[0]);
                            acc[1] = fmaf(w[k][0].y, h[2 + dx], acc[1]);
                            acc[2] = fmaf(w[k][0].z, h[3 + dx], acc[2]);
                            acc[3] = fmaf(w[k][0].w, h[4 + dx], acc[3]);
                            acc[4] = fmaf(w[k][1].x, h[5 + dx], acc[4]);
                            acc[5] = fmaf(w[k][1].y, h[6 + dx], acc[5]);
                            acc[6] = fmaf(w[k][1].z, h[7 + dx], acc[6]);
                            acc[7] = fmaf(w[k][1].w, h[8 + dx], acc[7]);
                        }
                    }
                    // voxels outside the volume have zero gates: their value stays 0
                    const float4 r0 = make_float4(acc[0], acc[1], acc[2], acc[3]), r1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
                    // the owned voxels of a level go to memory only in the last step (and, for the backward, as level history):
                    // where they go is worked out inside those (uniform) branches, the registers are needed elsewhere
                    auto store_owned = [&](float* dst) {
                        const int z = z0 + lz, y = y0 + ly, x = x0 + lx;   // x: position on the row of volumes
                        if (z >= g.D || y >= g.H) return;
                        const int row = (z * g.H + y) * g.W;
                        int bb, xx;
                        loc(x, bb, xx);
                        if (xx >= 0 && xx + 3 < g.W && bb < g.B && x >= ox0 && x < ox1)
                            *reinterpret_cast<float4*>(dst + (unsigned)(bb * FBS + row + xx)) = r0;
                        loc(x + 4, bb, xx);
                        if (xx >= 0 && xx + 3 < g.W && bb < g.B && x + 4 >= ox0 && x + 4 < ox1)
                            *reinterpret_cast<float4*>(dst + (unsigned)(bb * FBS + row + xx)) = r1;
                    };
                    if (it == g.n_iter) {
                        store_owned(MULTI ? out + (size_t)ch * V : out);
#if !defined(P3_ROWS_PLAIN) && !defined(P3_ROWS_CF) && !defined(P3_NO_PRIO)
                        __builtin_amdgcn_s_setprio(0);
#endif
                        break;
                    }
                    float* own = nxt + ((lz + 1) * LYP + (ly + 1)) * LX + lx + 1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) own[i] = acc[i];
                    // the level history of the backward (MULTI: volumes laid out [level][B][C][V], like feat / out)
                    if (levels) store_owned(levels + (size_t)(g.lv0 + it * g.lvs) * (MULTI ? total * (size_t)g.C : total) + (MULTI ? (size_t)ch * V : 0));
                    P3_STAMP(1);
                    // publications are numbered through the whole launch and alternate between the two buffers, so the
                    // one overwritten was consumed by every neighbour (they published the step in between) and chunks need no barrier
                    const unsigned target = (MULTI ? round * (unsigned)nch + (unsigned)ch : round) * (unsigned)(g.n_iter - 1) + (unsigned)it;
                    {
                        // ---- publish the tile's boundary straight from the registers as self-validating 16-byte quads (up to three
                        // values + the step tag): no wait for the stores, no flag, no barrier -- a reader polls the quad it needs
                        // until the tag is the step's.  A thread's eight values are the quads (0,1,2) (3,4,5) (6,7) of its row;
                        // boundary rows publish all 24, the others only the first and the last (the x faces).
                        if (!MUTE || wg != g.mute) {
                            float4* mine = X + ((size_t)(target & 1) * g.n_wg + wg) * NQ;
                            float4* rowq = mine + (lz * TY + ly) * XG + (lx >> 3);
                            const float tagf = __uint_as_float(target);
                            // a row all of whose readers run on this XCD is stored without scope bits (the line stays in the
                            // XCD's L2, where the readers' sc1 loads find it: no memory-side round trip, no memory-side bytes);
                            // every other row write-through (sc1) as before.  Per 128-byte line, never mixed inside one.
                            if (lz == 0 || lz == TZ - 1 || ly == 0 || ly == TY - 1) {
                                if ((s_rowl2 >> (lz * TY + ly)) & 1ull) {
                                    st16_l2(reinterpret_cast<float*>(rowq), make_float4(acc[0], acc[1], acc[2], tagf));
                                    st16_l2(reinterpret_cast<float*>(rowq + NROWS * XG), make_float4(acc[3], acc[4], acc[5], tagf));
                                    st16_l2(reinterpret_cast<float*>(rowq + 2 * NROWS * XG), make_float4(acc[6], acc[7], 0.f, tagf));
                                } else {
                                    st16_sc1(reinterpret_cast<float*>(rowq), make_float4(acc[0], acc[1], acc[2], tagf));
                                    st16_sc1(reinterpret_cast<float*>(rowq + NROWS * XG), make_float4(acc[3], acc[4], acc[5], tagf));
                                    st16_sc1(reinterpret_cast<float*>(rowq + 2 * NROWS * XG), make_float4(acc[6], acc[7], 0.f, tagf));
                                }
                            }
                            if (lx == 0) {
                                float* q = reinterpret_cast<float*>(mine + NQA + lz * TY + ly);
                                if ((s_xl2 >> lz) & 1u) st16_l2(q, make_float4(acc[0], 0.f, 0.f, tagf));
                                else st16_sc1(q, make_float4(acc[0], 0.f, 0.f, tagf));
                            }
                            if (lx == TX - 8) {
                                float* q = reinterpret_cast<float*>(mine + NQA + NROWS + lz * TY + ly);
                                if ((s_xl2 >> (TZ + lz)) & 1u) st16_l2(q, make_float4(acc[7], 0.f, 0.f, tagf));
                                else st16_sc1(q, make_float4(acc[7], 0.f, 0.f, tagf));
                            }
                        }
#if !defined(P3_ROWS_PLAIN) && !defined(P3_ROWS_CF) && !defined(P3_NO_PRIO)
                        __builtin_amdgcn_s_setprio(0);
#endif
                        // ---- the halo shell: 36 rows of the neighbours above / below / beside in y (24 quads each) and the 200 voxels
                        // beside the tile in x (one value of a neighbour's first or last quad), polled until their tag is this step's
                        unsigned src[NSLOT];   // byte offset of the quad in X
                        int dstp[NSLOT];       // LDS float index | count << 14 ; -1: nothing to fetch
#pragma unroll
                        for (int j = 0; j < NSLOT; ++j) {
                            // (recomputed every step: a table of these in LDS was measured no faster, and the registers to keep
                            // them are taken by the gates)
                            const int item = tid_ + j * NTP;
                            dstp[j] = -1;
                            src[j] = 0;
                            if (item >= NIT) continue;
                            int pz, py, px, q3 = -1, xg = 0, cnt = 1;
                            if (item < NHQ) {
                                const int hr = item / QROW, quad = item - hr * QROW;
                                if (hr < 2 * LY) { pz = hr < LY ? 0 : LZ - 1; py = hr < LY ? hr : hr - LY; }
                                else { const int u = hr - 2 * LY; py = u < TZ ? 0 : LY - 1; pz = 1 + (u < TZ ? u : u - TZ); }
                                q3 = quad >> XGS;
                                xg = quad & (XG - 1);
                                px = 1 + 8 * xg + 3 * q3;
                                cnt = q3 == 2 ? 2 : 3;
                            } else {
                                const int u = item - NHQ, f = u / (LZ * LY), r = u - f * (LZ * LY);
                                pz = r / LY; py = r - pz * LY; px = f ? LXU - 1 : 0;
                                xg = f ? 0 : 1;      // right halo: the neighbour's x = 0 (side 0); left halo: its x = 63 (side 1)
                            }
                            const int tz2 = iz + (pz == 0 ? -1 : (pz == LZ - 1 ? 1 : 0)), ty2 = iy + (py == 0 ? -1 : (py == LY - 1 ? 1 : 0)),
                                      tx2 = ix + (px == 0 ? -1 : (px == LXU - 1 ? 1 : 0));
                            if (tz2 < 0 || tz2 >= g.tz || ty2 < 0 || ty2 >= g.ty || tx2 < 0 || tx2 >= g.cx) continue;
                            // the row inside the neighbour tile
                            const int sz = pz == 0 ? TZ - 1 : (pz == LZ - 1 ? 0 : pz - 1), sy = py == 0 ? TY - 1 : (py == LY - 1 ? 0 : py - 1);
                            const int nbw = (tz2 * g.ty + ty2) * g.cx + tx2;
                            const int quad = q3 >= 0 ? (q3 * NROWS + sz * TY + sy) * XG + xg : NQA + xg * NROWS + sz * TY + sy;
                            src[j] = (unsigned)((((int)(target & 1) * g.n_wg + nbw) * NQ + quad) * 16);
                            dstp[j] = ((pz * LYP + py) * LX + px) | (cnt << 14);
                        }
                        // A neighbour that never publishes (it is not resident: the device is shared with work that holds CUs) must not
                        // hang the kernel nor pass unnoticed: after SPIN_MAX polls (~0.5 s) the values that did not come are
                        // taken as NaN -- the remaining steps carry them into this tile's voxels and on --, the device's sticky
                        // status word is raised (cspn3d_check_status), and this workgroup waits only briefly from then on.
                        const unsigned limit = s_bail ? 8u : SPIN_MAX;
                        unsigned tries = 0;
                        bool pend = false, lost = false;
#pragma unroll
                        for (int j = 0; j < NSLOT; ++j) pend = pend || dstp[j] >= 0;
#ifdef P3_PRESLEEP
                        asm volatile("s_sleep %0" : : "n"(P3_PRESLEEP));
#endif
#ifdef P3_EXP_NOPOLL   // WRONG RESULTS, timing only: no halo at all
                        pend = false;
#endif
                        while (pend) {
                            v4f qv[NSLOT];
#pragma unroll
                            for (int j = 0; j < NSLOT; ++j)
                                if (dstp[j] >= 0) qv[j] = ldq_sc1(X, src[j]);
#pragma unroll
                            for (int j = 0; j < NSLOT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(qv[j]) : : "memory");
#ifdef P3_TRACE
                            if (tries == 0) { P3_STAMP(2); }
#endif
                            pend = false;
#pragma unroll
                            for (int j = 0; j < NSLOT; ++j) {
                                if (dstp[j] < 0) continue;
#ifdef P3_EXP_NOWAIT   // WRONG RESULTS, timing only: take whatever is there
                                const bool hit = true;
#else
                                const bool hit = __float_as_uint(qv[j].w) == target;
#endif
                                if (hit || tries >= limit) {
                                    float* lp = nxt + (dstp[j] & 0x1fff);
                                    const int cnt = dstp[j] >> 14;
                                    const float bad = __uint_as_float(0x7fc00000u);
                                    lp[0] = hit ? qv[j].x : bad;
                                    if (cnt >= 2) lp[1] = hit ? qv[j].y : bad;
                                    if (cnt == 3) lp[2] = hit ? qv[j].z : bad;
                                    dstp[j] = -1;
                                    lost = lost || !hit;
                                } else {
                                    pend = true;
                                }
                            }
                            ++tries;
#if P3_BACKOFF > 0
                            // a quad that was not there yet will take a good part of a memory-side round trip to appear: polling at
                            // full speed only multiplies the polled bytes (1.4 GB per forward at config 5, profiles/pmc_traffic.json)
                            if (pend) __builtin_amdgcn_s_sleep(P3_BACKOFF);
#endif
                        }
                        if (lost) {
                            *err = 2;
                            {   // the device's sticky status word <- this launch's number.  Both come straight out of the kernel-argument
                                // segment HERE: as ordinary uses of g.status / g.seq they were live (and spilled) across the whole kernel
                                typedef __attribute__((address_space(4))) const char kchar;   // (constant address space: scalar loads, no flat access)
                                kchar* ka = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
                                const unsigned seq_ = *(__attribute__((address_space(4))) const volatile unsigned*)(ka + GEO3_KERNARG_OFFSET + offsetof(Geo3, seq));
                                unsigned* const st_ = *(unsigned* __attribute__((address_space(4))) const volatile*)(ka + GEO3_KERNARG_OFFSET + offsetof(Geo3, status));
                                if (st_) __hip_atomic_store(st_, seq_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            s_bail = 1;
                        }
                    }
                    __syncthreads();
                    P3_STAMP(4);
                }
                }   // value channel
                P3_CHUNK(3);
            }
        }
    }
}

// per-device host state: what the residency needs (CU count), the chaining event of the plain launches, and the sticky status
// word the kernel raises when a workgroup gives up -- one pinned, device-visible word per device, read by the host without a
// synchronisation at the start of every later 3D call.  (This is the engine's only global mutable state.)
struct Dev3 {
    int wgs = 0;                      // workgroups that can be resident at once: one per CU (the kernel takes a CU's whole
    bool wgs_known = false;           //   register file), at most MAX_WG; 0: the occupancy API says the kernel does not fit
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
    unsigned* status_host = nullptr;  // host-mapped; status_dev is its device address (a kernel argument: Geo3::status)
    unsigned* status_dev = nullptr;
    unsigned seq = 0;                 // launches so far; a workgroup that gives up stores its launch's number in the status word
    unsigned reported = 0;            // the highest launch number persistent3d_take_status has reported
};
std::mutex g_mu3;
Dev3 g_dev3[64];

Dev3& dev3() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return g_dev3[dev];
}

// Pre-flight (round 5): how many workgroups of the persistent kernel the device can hold AT ONCE = CUs x what the occupancy API
// says one CU takes of each product variant (1 with 512 threads at 256 registers and ~150 KB of LDS; 0 if a variant cannot be
// launched at all on this device -- then the persistent path is off and AUTO runs the per-step kernels instead of producing NaN).
// The plan never asks for more workgroups than this (make_geo3), so "n_wg <= occupancy x CUs" holds by construction.  What the
// API cannot see -- CUs held by ANOTHER process or a CU mask applied behind the runtime's back -- is what the poll timeout and
// the status word are for.
int resident_wgs() {
    std::lock_guard<std::mutex> lock(g_mu3);
    Dev3& d = dev3();
    if (!d.wgs_known) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = MAX_WG;
        int occ = WG_PER_CU;
        const void* fns[] = {(const void*)cspn3d_persistent_kernel<false, false>, (const void*)cspn3d_persistent_kernel<false, true>,
                             (const void*)cspn3d_persistent_kernel<true, false>, (const void*)cspn3d_persistent_kernel<false, false, false, true>,
                             (const void*)cspn3d_persistent_kernel<true, false, false, true>};
        for (const void* fn : fns) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NTP, 0) == hipSuccess && nb < occ) occ = nb < 0 ? 0 : nb;
        }
        (void)hipGetLastError();
        v *= occ;
        d.wgs = v < MAX_WG ? v : MAX_WG;
        d.wgs_known = true;
    }
    return d.wgs;
}

Geo3 make_geo3(int B, int D, int H, int W, int n_iter, bool placement = true) {
    Geo3 g{};
    g.B = B; g.D = D; g.H = H; g.W = W; g.n_iter = n_iter;
    g.halo = 4 * ((n_iter + 3) / 4);
    g.tz = (D + TZ - 1) / TZ;
    g.ty = (H + TY - 1) / TY;
    const int per_col = g.tz * g.ty;
    g.cx = per_col > 0 ? resident_wgs() / per_col : 0;
    g.pitch = W + 4;                  // the volumes of the batch side by side along x, four never-valid columns between them
    g.vw = B * g.pitch - 4;
    const int need = (g.vw + 2 * g.halo + TX - 1) / TX;     // x-tiles that cover the whole row plus halos: no need for more
    if (g.cx > need) g.cx = need;
    g.S = g.cx * TX - 2 * g.halo;
    g.nchunk = g.S > 0 ? (g.vw + g.S - 1) / g.S : 0;
    g.n_wg = per_col * g.cx;
    g.n_launch = g.n_wg;
    g.bz = g.by = g.bx = 0;
    g.nby = g.nbx = 1;
    // XCD-aware placement: cut the tz x ty x cx tile grid into at most 8 equal blocks of at most 32 tiles and give block k to the
    // workgroup ids = k (mod 8).  Among the exact cuts with the most blocks (every XCD busy) take the one with the smallest
    // cross-block surface, in quads a tile fetches per step: z face LY x QROW, y face TZ x QROW, x face LZ x LY.  Off (bz = 0, plain
    // order) when the device is not 8 x 32 CUs, when there are fewer tiles than 16, or when no exact cut gives at least 4 blocks.
    if (placement && resident_wgs() == 256 && g.n_wg >= 16) {
        long best = -1;
        int best_blocks = 0;
        for (int bz = 1; bz <= g.tz; ++bz) {
            if (g.tz % bz) continue;
            for (int by = 1; by <= g.ty; ++by) {
                if (g.ty % by) continue;
                for (int bx = 1; bx <= g.cx; ++bx) {
                    if (g.cx % bx) continue;
                    const int nz = g.tz / bz, ny = g.ty / by, nx = g.cx / bx, blocks = nz * ny * nx;
                    if (blocks > 8 || blocks < 4 || bz * by * bx > 32) continue;
                    const long surf = (long)(nz - 1) * g.ty * g.cx * (LY * QROW) + (long)(ny - 1) * g.tz * g.cx * (TZ * QROW) +
                                      (long)(nx - 1) * g.tz * g.ty * (LZ * LY);
                    if (blocks > best_blocks || (blocks == best_blocks && surf < best)) {
                        best_blocks = blocks; best = surf;
                        g.bz = bz; g.by = by; g.bx = bx; g.nby = ny; g.nbx = nx;
                    }
                }
            }
        }
        if (g.bz > 0) g.n_launch = 8 * g.bz * g.by * g.bx;
    }
    return g;
}

}  // namespace

bool persistent3d_supported(int B, int D, int H, int W, int n_iter) {
    if (B <= 0 || n_iter < 2 || n_iter > 60 || (W % 4) != 0) return false;
    const Geo3 g = make_geo3(B, D, H, W, n_iter);
    // worth it only when a chunk owns clearly more than it recomputes, and the device can hold it
    // (lane offsets into the gate tensor are 32-bit byte offsets)
    return g.pitch >= LXU && g.cx >= 1 && g.n_launch <= resident_wgs() && g.S >= 4 * g.halo && g.nchunk < (1 << 24) &&
           (long long)B * 26 * D * H * W * 4 < (1LL << 32);
}

constexpr size_t XBYTES = 2 * (size_t)MAX_WG * NQ * 16;   // published tile boundaries, two level parities

// One size for the whole Paddle-contract call, whichever path ends up taking it (the size query has no algo argument and the
// same workspace must serve `algo` 1): two value volumes -- the ping-pong of the per-step path, which also takes the calls
// the persistent kernel declines (residency, n_iter); the persistent kernel itself does not touch them --, then the published
// tile boundaries and the sync / error words.
size_t persistent3d_workspace(int B, int D, int H, int W) {
    return 2 * (size_t)B * D * H * W * sizeof(float) + XBYTES + 4096 * sizeof(unsigned);
}

// adjoint: the transposed operator (backward); levels: volume lv0 + it * lvs receives the result of step it < n_iter;
// cprime != nullptr: gate holds the 26 folded planes [26][B][V] and cprime the constant term (normalising / masked modes)
static int persistent3d_launch(const float* gate, const float* feat, const float* cprime, float* out, float* levels, int lv0, int lvs,
                               bool adjoint, int B, int D, int H, int W, int n_iter, void* ws, hipStream_t st, const P3Options& opt, int C = 1) {
    Geo3 g = make_geo3(B, D, H, W, n_iter, opt.placement);
    g.C = C;
    g.wt = opt.write_through ? 1 : 0;
    g.lv0 = lv0;
    g.lvs = lvs;
    const size_t total = (size_t)B * D * H * W;
    g.gps = cprime ? (long long)total : (long long)(total / B);
    g.gbs = cprime ? (long long)(total / B) : 26LL * (long long)(total / B);
    float* scratch = (float*)ws;
    unsigned* sync = (unsigned*)((char*)(scratch + 2 * total) + XBYTES);
    // tags of an earlier call in this workspace must not validate: clear the published boundaries this launch indexes
    // ([2][n_wg][NQ] quads) and the sync words
    hipError_t e = hipMemsetAsync(scratch + 2 * total, 0, 2 * (size_t)g.n_wg * NQ * 16, st);
    if (e == hipSuccess) e = hipMemsetAsync(sync, 0, 4096 * sizeof(unsigned), st);
    if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    // Every workgroup waits for its neighbours' publications, so all of them must be resident at once, and two such kernels must
    // never be interleaved on the device (each would hold the CUs the other is waiting for).  Kernels of this process are kept
    // apart with an event: a launch waits for the previous persistent launch (whatever stream it went to) and records itself.
    // Other work that happens to occupy CUs only delays the start; if it never lets go, the waiting workgroups give up after
    // SPIN_MAX polls (~0.5 s), take NaN for what did not come and raise the device's status word (persistent3d_take_status).
    // opt.coop (test-hook library only) uses hipLaunchCooperativeKernel instead (the runtime then checks the residency too; ~23 us
    // per launch, 0.905 -> 0.928 ms at config 5).  A stream that is being captured into a graph takes a plain launch: the
    // replaying graph is the caller's to keep alone on the device.
    std::lock_guard<std::mutex> lock(g_mu3);
    Dev3& d = dev3();
    if (!d.status_host) {   // first launch on this device: one pinned word; its device address travels as a kernel argument, so
        // nothing here touches a stream (no symbol copy, no implicit synchronisation: safe under stream capture)
        // (a pinned allocation is an "unsafe" call while some stream of the thread is being captured in the default global mode: it
        // would invalidate the capture -- the documented way for a library to allocate regardless is the relaxed mode, for the call)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        e = hipHostMalloc((void**)&d.status_host, 64, hipHostMallocMapped);
        if (e == hipSuccess) { *d.status_host = 0; e = hipHostGetDevicePointer((void**)&d.status_dev, d.status_host, 0); }
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) { set_error("status word of the persistent kernel: %s", hipGetErrorString(e)); d.status_host = nullptr; return (int)e; }
    }
    const int mute = opt.mute;
    g.mute = mute;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // the status word receives the NUMBER of the launch that gave up -- except for a launch captured into a graph: its kernel
    // arguments are frozen at capture time, every replay would store the same number and only the first failing replay would be
    // reported.  Captured launches therefore store the fixed marker CAPTURED_SEQ, which persistent3d_take_status clears with an
    // exchange when it reports it: every failing replay is reported (a replay whose other workgroups time out after the report
    // may be reported twice -- the pre-round-4 behaviour, confined to graphs).
    if (capturing) {
        g.seq = CAPTURED_SEQ;
    } else {
        if (++d.seq == 0 || d.seq == CAPTURED_SEQ) d.seq = 1;
        g.seq = d.seq;
    }
    g.status = d.status_dev;
    void* args[] = {(void*)&gate, (void*)&feat, (void*)&cprime, (void*)&out, (void*)&levels, (void*)&scratch, (void*)&sync, (void*)&g};
    const void* fn = cprime ? (const void*)cspn3d_persistent_kernel<false, true>
                   : adjoint ? (C > 1 ? (const void*)cspn3d_persistent_kernel<true, false, false, true> : (const void*)cspn3d_persistent_kernel<true, false>)
                   : C > 1 ? (const void*)cspn3d_persistent_kernel<false, false, false, true>
                   : mute >= 0 ? (const void*)cspn3d_persistent_kernel<false, false, true> : (const void*)cspn3d_persistent_kernel<false, false>;
    const bool coop = opt.coop;
    if (coop && !capturing) {
        e = hipLaunchCooperativeKernel(fn, dim3(g.n_launch), dim3(NTP), args, 0, st);
    } else if (capturing) {
        e = hipLaunchKernel(fn, dim3(g.n_launch), dim3(NTP), args, 0, st);
    } else {
        if (!d.last) {
            e = hipEventCreateWithFlags(&d.last, hipEventDisableTiming);
            if (e != hipSuccess) { set_error("hipEventCreate: %s", hipGetErrorString(e)); return (int)e; }
        } else if (d.last_stream != st) {
            e = hipStreamWaitEvent(st, d.last, 0);
            if (e != hipSuccess) { set_error("hipStreamWaitEvent: %s", hipGetErrorString(e)); return (int)e; }
        }
        e = hipLaunchKernel(fn, dim3(g.n_launch), dim3(NTP), args, 0, st);
        if (e == hipSuccess) {
            e = hipEventRecord(d.last, st);
            d.last_stream = st;
        }
    }
    if (e != hipSuccess) { set_error("launch of cspn3d_persistent_kernel: %s", hipGetErrorString(e)); return (int)e; }
    return check_launch("cspn3d_persistent_kernel");
}

// the sticky status of this device's persistent launches: 0 = no launch gave up that was not reported before.
// Reads a pinned host word the kernel writes with a system-scope store (no synchronisation).  The word holds the NUMBER of the
// launch that gave up: a launch is reported once -- its other workgroups run into their own timeouts later and store the same
// number again, which a later, innocent call must not see as a second failure.  (Launches of a device are chained by an event, so
// an older launch never overwrites a newer one's number.)
int persistent3d_take_status() {
    std::lock_guard<std::mutex> lock(g_mu3);
    Dev3& d = dev3();
    if (!d.status_host) return 0;
    const unsigned v = __atomic_load_n(d.status_host, __ATOMIC_RELAXED);
    if (v == CAPTURED_SEQ) {   // a replayed graph's launch gave up: clear, so that the next failing replay is seen as well
        __atomic_exchange_n(d.status_host, 0u, __ATOMIC_RELAXED);
        return 2;
    }
    if (v == 0 || v == d.reported) return 0;
    d.reported = v;
    return 2;
}

int persistent3d_run(const float* gate, const float* feat, float* out, float* levels, int lv0, int lvs, bool adjoint, int B, int D,
                     int H, int W, int n_iter, void* ws, hipStream_t st, const P3Options& opt, int C) {
    return persistent3d_launch(gate, feat, nullptr, out, levels, lv0, lvs, adjoint, B, D, H, W, n_iter, ws, st, opt, C);
}

// H_{t+1} = c' + sum_k w'_k H_t(p + off_k) with the folded planes wf = [26 w'][c'] of fold3d_kernel
int persistent3d_forward_folded(const float* wf, const float* feat, float* out, int B, int D, int H, int W, int n_iter, void* ws,
                                hipStream_t st) {
    return persistent3d_launch(wf, feat, wf + 26 * (size_t)B * D * H * W, out, nullptr, 0, 0, false, B, D, H, W, n_iter, ws, st, P3Options());
}

// C value channels per volume on shared gates (feat, out: [B][C][V]; the Paddle contract): one gate load per chunk for all of them
bool persistent3d_multi_supported(int B, int C, int D, int H, int W, int n_iter) {
    if (!(C >= 1 && persistent3d_supported(B, D, H, W, n_iter) && (long long)B * C * D * H * W * 4 < (1LL << 32))) return false;
    // publications are numbered (round * C + channel) * (n_iter - 1) + step in 32 bits: the number must never wrap to a tag the
    // cleared exchange buffers would validate
    const Geo3 g = make_geo3(B, D, H, W, n_iter);
    return (long long)g.nchunk * C * (n_iter - 1) < (1LL << 31);
}

int persistent3d_forward_multi(const float* gate, const float* feat, float* out, int B, int C, int D, int H, int W, int n_iter, void* ws,
                               hipStream_t st) {
    return persistent3d_launch(gate, feat, nullptr, out, nullptr, 0, 0, false, B, D, H, W, n_iter, ws, st, P3Options(), C);
}

int persistent3d_forward(const float* gate, const float* feat, float* out, int B, int D, int H, int W, int n_iter, void* ws,
                         hipStream_t st) {
    return persistent3d_run(gate, feat, out, nullptr, 0, 0, false, B, D, H, W, n_iter, ws, st);
}

// (test-hook library) the plan of a persistent launch: info[9] = tz, ty, cx, tiles, workgroups launched, bz, by, bx (0: plain order), chunks
void persistent3d_geo(int B, int D, int H, int W, int n_iter, int* info) {
    const Geo3 g = make_geo3(B, D, H, W, n_iter);
    const int v[9] = {g.tz, g.ty, g.cx, g.n_wg, g.n_launch, g.bz, g.by, g.bx, g.nchunk};
    for (int i = 0; i < 9; ++i) info[i] = v[i];
}

// (test-hook library) the error word of the last run in this workspace (0 ok, 2 neighbour-quad timeout); synchronises
int persistent3d_error_word(const void* ws, int B, int D, int H, int W) {
    unsigned v = 0;
    const unsigned* p = (const unsigned*)((const char*)((const float*)ws + 2 * (size_t)B * D * H * W) + XBYTES) + MAX_WG + 64 * 9;
    if (hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

}  // namespace cspn
