"""tools/tswgen/kernel.py -- generator of the gfx950 main loop of the fused CSPN kernel ("time-skewed wave ring").

Same algorithm as the C++ kernel in cspn_amd/csrc/cspn2d_fused.hip (executable spec: tools/tsw_model.py), but every
register is assigned by hand and the 24 phases of a wave's ring counter are unrolled, so that
  * the event-free step is exactly 64 v_pk_fma_f32 + 24 moves + 4 LDS ops,
  * retire / inject code exists only in the 4 phases that need it, specialised per slot,
  * cooking (normalise + fold, cspn.py:85-144,:76,:81) exists only in the 8 phases with counter % 3 == 2,
  * row descriptors come from a table built by a tiny planning kernel and are fetched with scalar loads.
The C++ side (cspn2d_tsw.hip) provides kernel arguments in fixed SGPRs and the LDS allocation.

Register / LDS maps are module constants; `build(cfg)` returns an isa.Prog.
cfg: norm (0 '8sum', 1 '8sum_abs', 2 'none', 3 'prenorm': the caller hands over what reference affinity_normalization returns
-- gate_wb, cspn.py:85-144: normalised and consumer-sited -- so cooking is only sigma = sum_k w_k, c' = (1 - sigma) H0 and the
mask fold: SURVEY 8f-2, second alternative), sparse (bool), hin (bool: level-0 values come from a previous pass),
n_iter (only 24 for now).
"""
from .isa import Prog, V, S, EXEC, VCC, schedule, check_hazards, expand_pseudos

NW, NSLOT, LV = 8, 4, 24
HIST_EVERY = 4   # the history / adjoint variants of the product keep every fourth level (cfg hist_every; cspn2d_backward.hip recomputes the rest)
PADF, PADB = 36, 48          # inactive descriptor rows before / after a workgroup's stream
DESC_BYTES = 16              # goff_lo, goff_hi, boff, flags | lo << 8 | hi << 20
# ring of cooked rows: 8 slots x 64 lane records; a record = [plane 0..9][column 0..3] floats + 2 pad = 42 dwords, so that the
# cooking lanes (two per record, 8 adjacent bytes each) and the reading lanes (stride 42 dwords) are free of bank conflicts
import os as _os
RING_REC = int(_os.environ.get("TSW_RING_REC", "168"))   # (experiment builds: tools/r05/build_ringrec.sh -- 168 is the product's)
RING_W2 = _os.environ.get("TSW_RING_W2", "0") == "1"      # ring stores as ds_write2_b32 (4-byte aligned): allows an odd dword stride, e.g. 172 bytes
LDS_BND, LDS_RING, RING_SLOT = 0, 32768, 64 * RING_REC
# cfg elastic: no barrier in the loop -- a wave tells its ring neighbours with a tag that the boundary rows of a step are in LDS
# (mailbox of the READER: [8 waves][2 counter parities][top, bottom] dwords), and the cooking waves tag the ring slots
# ([8 slots][2 halves] dwords, at TAG_RT).  Inside the range the C++ preamble zero-fills: tag 0 is never expected.
LDS_TAGS = LDS_RING + 8 * RING_SLOT
TAG_RT = 128
LDS_TAB = LDS_TAGS                   # the workgroup's row-descriptor table (written by the C++ part of the kernel)
TAB_MAX_ROWS = min(2816, (163840 - LDS_TAB) // 16)
LDS_BYTES = LDS_TAB + TAB_MAX_ROWS * DESC_BYTES   # 160 KB
assert LDS_BYTES <= 163840


def configure(tag_area):
    """LDS map with / without the 256-byte tag area of cfg elastic in front of the descriptor table.  The product map has none:
    with the table 256 bytes higher the loop measured 0.5 .. 1 % slower (profiles/r02_ablations_l_elastic.txt)."""
    global LDS_TAB, TAB_MAX_ROWS, LDS_BYTES
    import os
    shift = int(os.environ.get("TSW_TAB_SHIFT", "0"))   # experiment: where the table starts (multiple of 16 bytes)
    LDS_TAB = LDS_TAGS + (256 if tag_area else 0) + shift
    TAB_MAX_ROWS = min((2800 if tag_area else 2816) - shift // 16, (163840 - LDS_TAB) // 16)
    LDS_BYTES = LDS_TAB + TAB_MAX_ROWS * DESC_BYTES
DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]

# flags in descriptor dword 3
F_ACTIVE, F_UP, F_DN, F_FIRST, F_LAST, F_OWNED, F_PLAIN = 0, 1, 2, 3, 4, 5, 6  # PLAIN: active, interior row, interior band

# ---- VGPR map ----
V_LANE, V_COL4, V_L16 = V(0), V(1), V(2)
V_WR, V_RT, V_RB = [V(3), V(4)], [V(5), V(6)], [V(7), V(8)]
V_RINGR = V(9)              # ring record of this lane in slot 0 of the wave's half of the ring
V_RINGE = [V(0), V(1)]      # (after the prologue) the same record in the slot of the current event / of the slot before it
V_RINGW = [V(10), V(11)]
V_OFFK = [V(12 + k) for k in range(8)]
V_OFF1 = V(20)
V_TMP = V(21)
V_DE, V_DO, V_DC = V(21), V(22, 2), V(252, 4)  # descriptor fetches (event: new row's d3, old row's d2:d3; cooking: whole row)
V_DN = V(74, 2)                                 # history mode: new row's d2:d3
TQ, BQ, TA, TB, HN, HA, OUTQ = V(24, 4), V(28, 4), V(32, 4), V(36, 4), V(40, 4), V(44, 4), V(48, 4)
# one pair per shifted row value of a step (D = (c3 of lane-1, c0 of lane+1)), so that the pushes nobody waits for can
# trail behind the chain: received rows, slots 3..1, and slot 0's deferred pushes (cooking, at the top of a step, uses
# TA, TB and OUTQ[0:1] while only D_TAIL is live)
D_BQ, D_TQ, D_SLOT, D_TAIL = OUTQ.sub(0, 2), TB.sub(2, 2), {3: TA.sub(0, 2), 2: TA.sub(2, 2), 1: TB.sub(0, 2)}, OUTQ.sub(2, 2)
CK = V(24, 28)  # cooking temporaries alias the step temporaries
PEND_G = [V(52 + 2 * k, 2) for k in range(8)]
PEND_BLUR, PEND_HIN, PEND_SP = V(68, 2), V(70, 2), V(72, 2)
ACC_BASE, WT_BASE = 76, 108


def ACC(p, j):
    return V(ACC_BASE + (p * 4 + j) * 4, 4)


def WT(j, k):
    return V(WT_BASE + (j * 9 + k) * 4, 4)


# ---- SGPR map ----
S_GD, S_BLUR, S_HIN, S_SP, S_OUT, S_PLAN = S(16, 2), S(18, 2), S(20, 2), S(22, 2), S(24, 2), S(26, 2)
S_W4, S_HW4, S_LAST, S_WV = S(28), S(29), S(30), S(31)
S_LDSB = S(15)  # LDS base address of the kernel's __shared__ block
S_NROWS = S(14)  # cfg tab_in_lds = False only: descriptors in this workgroup's table (<= TAB_MAX_ROWS)
S_GLAST = S(14)  # cfg pf: largest row-base byte offset inside the guidance tensor a prefetch may start from (0: tensor >= 4 GB)
V_PF, V_PFD = V(74), V(75)   # cfg pf (not with hist: V_DN): per-lane byte offset of the cache line a lane touches; dummy target
S_PFB = S(54, 2)             # cfg pf: base address of the row being prefetched
S_LOHI = S(13)   # input: owned columns of this workgroup's band, band relative: lo | hi << 16 (one band per workgroup)
S_OMASK = S(42, 2)  # lanes whose 4 columns lie inside [lo, hi)
S_LOHIC = S(0)      # cfg mband (forward variants; s0 is S_HM[0] in history mode): lo | hi << 12 of the row S_OMASK was derived from
S_TAU, S_ACT, S_QB, S_PQ, S_PFLAGS = S(45), S(33), S(34), S(35), S(36)  # (s32 is reserved by the compiler: stack pointer)
S_AM = [S(56), S(57), S(58), S(59)]   # per ring slot: -1 if it holds a real row, 0 for a separator / padding row (round 3)
S_EL, S_ER = S(38, 2), S(40, 2)
S_ENF = S(44)                   # event: the entering row's descriptor dword 3 (flags)
S_EO = S(48, 2)                 # event: the retiring row's descriptor dwords 2:3 (boff, flags | lo | hi)
S_CD = S(60, 4)                 # cooking: descriptor of the task being requested
S_TABB = S(37)                  # LDS address of descriptor row 0 (table base + PADF rows)
# history mode (cfg hist): every level 1..23 of every owned row is also written out (the backward pass needs all H_t)
S_HIST, S_HSTRIDE = S(92, 2), S(94, 2)       # inputs: base of the level-1 plane, bytes from one level's plane to the next
S_ENB = S(46)                                   # entering row's descriptor dword 2 (byte offset in a 1-channel tensor)
S_HB = [S(84, 2), S(86, 2), S(88, 2), S(90, 2)]     # per slot: where the row's next level goes
S_HM = [S(0, 2), S(4, 2), S(96, 2), S(98, 2)]       # per slot: lanes that own columns of the row (0: halo / inactive row)
S_WF = S(50, 2)     # history mode: the 8 folded coefficient planes w'_k ([8][B*H*W], right behind the 23 level planes)
S_PBOFF = S(47)     # history mode: byte offset (1-channel tensor) of the pending task's row
S_CMASK = S(52, 2)  # history mode: cooking lanes (2 pixels each) whose pixels lie in the band's owned columns
T = [S(68 + i) for i in range(12)]  # scalar temporaries s68..s79
# cfg early (forward variants; s92:95 are the history variants' inputs): a pass of n < 24 iterations -- a row's output is the level n it reaches n
# steps after its injection; it then stays in its slot, harmlessly, until the slot's next event
SAVE2 = [V(70), V(71), V(74), V(75)]   # cfg early: copy of slot 2's completed quad (its accumulator is re-initialised by slot 1's pushes in the same step); PEND_HIN / V_DN are unused there
S_NIT, S_EMASK = S(94), S(95)   # inputs: n (1..23); bit c set <=> some slot of a wave completes level n in the step with ring counter c: c = n .. n + 3 (mod 24)
# cfg elastic (never together with hist / hin / pf / trace, whose registers these are)
S_MBR, S_TWU, S_TWD, S_RTW = [S(84), S(85)], [S(86), S(87)], [S(88), S(89)], [S(90), S(91)]   # LDS addresses, per counter parity
S_RTR, S_EXP, S_EXR, S_SPIN, S_MB0, S_EXW = S(96), S(97), S(98), S(99), S(92), S(93)
V_TGB, V_TGR = V(70, 2), V(74, 2)   # boundary tags (top, bottom) / ring tags (cooking half 0, 1); also the tag writers' temporaries
SPIN_MAX = 1 << 16
S_ELC, S_ERC = S(80, 2), S(82, 2)   # per-wave constant lane masks (half 0: lane 0 / half 1: lane 63)
GB_MID, B_BLUR, B_HIN, B_SP = S(2, 2), S(6, 2), S(8, 2), S(10, 2)  # row bases of the requested task (s0:1, s4:5: S_HM)


class Gen(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.p = Prog()
        self.norm, self.sparse, self.hin = cfg.get("norm", 0), cfg.get("sparse", False), cfg.get("hin", False)
        self.hist = cfg.get("hist", False)
        # history mode keeps the levels that are multiples of hist_every (1: all of 1..23; 4: the checkpoints 4, 8 .. 20 the
        # recomputing final pass of the backward starts from); plane i of the history = level (i + 1) * hist_every
        self.hist_every = cfg.get("hist_every", 1)
        assert 24 % self.hist_every == 0
        # adj: the adjoint sweep of the backward pass as a propagation.  Coefficients come from the folded planes w'
        # ([8][B*H*W], what fold2d_kernel writes) read neighbour-sited with the channel order reversed --
        # G_k(p) = w'_{7-k}(p + off_k), since off_{7-k} = -off_k -- and are used as they are (no normalisation, c' = 0)
        self.adj = cfg.get("adj", False)
        if self.adj:
            assert self.norm == 2 and not self.sparse and not self.hin
        # s8: the guidance comes "pre-sited, pair-interleaved" -- [B][H][W/2][8][2] floats, record of pixel pair (x, x+1) =
        # (G_0(x), G_0(x+1), G_1(x), ...) with G_k(p) = g_k(p + off_k) already gathered (zero outside the image) by the producer
        # (cspn2d_guidance_to_sited8_f32): a task's guidance is four aligned 16-byte loads instead of eight unaligned 8-byte ones
        # and needs no edge patching (SURVEY 8f-2 experiment)
        self.s8 = cfg.get("s8", False)
        if self.s8:
            assert not self.adj and not self.hist
        # early (round 5): the pass delivers level n < 24 (n at run time, S_NIT): the ring still runs its 24 levels (rows live 24 steps, the band halo and
        # the warm-up rows stay 24 deep), but a row is STORED at the end of the step in which it completed level n, by an out-of-line stub that one
        # scalar test per step (S_EMASK bit c) guards; nothing is stored at retirement.  n_iter = 24 k + r runs r iterations here first, then k full passes.
        self.early = cfg.get("early", False)
        if self.early:
            assert not (self.hist or self.adj or self.s8 or self.hin or cfg.get("elastic") or cfg.get("stagger") or cfg.get("pf"))
        self.xstubs, self.xstub_of = [], {}
        self.given = self.norm in (2, 3)   # coefficients are used as given, centre-sited (3: with the centre term, 2: without)
        if self.norm == 3:
            assert not self.adj and not self.s8
        self.sited = (not self.given or self.adj) and not self.s8   # guidance plane k is read at (y + dy_k, x + dx_k)
        assert cfg.get("n_iter", 24) == 24
        self.stubs = []
        self.estubs = []
        # mband (round 4): a workgroup's stream may continue in the next band (linear plan, cspn2d_tsw_plan.h kind 1).  Everything
        # about a row's band travels in its descriptor already (offsets, first / last band, lo | hi); only the mask of the lanes
        # that own columns was a per-workgroup constant: a retirement now compares the row's lo | hi with the pair the mask was
        # derived from (2 scalar instructions + a never-taken branch) and re-derives it out of line when the band changed.
        # The history / adjoint / sited8 variants keep one band per workgroup (their per-slot masks are taken at injection).
        self.mband = cfg.get("mband", not (self.hist or self.adj or self.s8))
        assert not (self.mband and self.hist)
        self.mstubs = []
        self.elastic = cfg.get("elastic", False)
        if self.elastic:
            assert not (self.hist or self.hin or cfg.get("pf") or cfg.get("trace") or cfg.get("cook_early") or cfg.get("spread3"))
            assert cfg.get("tab_in_lds", True) and cfg.get("partial_wait", True) and cfg.get("slim_events", True)
        self.ab = set(cfg.get("ablate", ()))  # timing experiments only (results are wrong): nocook noevents noact nobar nolds

    # ---------------------------------------------------------------------------------- small helpers
    def e(self, op, dst=(), src=(), **m):
        return self.p.emit(op, dst, src, **m)

    def fma(self, d, a, b, c, **m):
        keep = m.pop("keep", False)
        if "nostep" in self.ab and not keep:
            return
        if m.pop("late", False) and self.cfg.get("late_at") is not None:
            m["at"] = self.cfg["late_at"]   # nobody in this step waits for the result: keep it out of the way of the chain
        self.e("v_pk_fma_f32", d, [a, b, c], **m)

    def mov(self, d, s):
        self.e("v_mov_b32", d, s)

    # Register layout of a row (4 columns c0..c3 per lane): the quad (c0, c3, c1, c2), i.e. the pairs X = (c0,c3) and
    # Y = (c1,c2).  With D = (c3 of lane-1, c0 of lane+1) -- two DPP moves -- every tap of every column is one half of a
    # v_pk_fma_f32 whose two h operands sit in ONE aligned register pair, so no value is ever copied:
    #   dst X = (a0,a3):  x   : X = (h0,h3)          x+1|x-1 : Y = (h1,h2)           x-1|x+1 : D = (h-1,h4)
    #   dst Y = (a1,a2):  x   : Y = (h1,h2)          x+1|x-1 : Y swapped (op_sel)    x-1|x+1 : X = (h0,h3)
    # The coefficient registers are laid out to match (WT(j, q), q = 0..8): for a row of taps with planes (kr, km, kl) =
    # (dx = +1, 0, -1) the quads  R = (kr0, kl3, kr1, kl2),  M = (km0, km3, km1, km2),  L = (kl0, kr3, kl1, kr2);
    # q = 0,1,2: below taps (planes 0,1,2)   q = 3,4: self taps R, L (planes 3,4)   q = 5,6,7: above taps (planes 5,6,7)
    # q = 8: c'.  inject() builds them straight out of the LDS ring with ds_read2st64_b32 (ring layout [plane][column][lane]).
    def shift(self, q, t):
        """q = (c0,c3,c1,c2); t[0:1] <- D = (c3 of lane-1, c0 of lane+1)"""
        if "nostep" in self.ab:
            return
        self.e("v_mov_b32", t[0], q[1], dpp="wave_shr:1")
        self.e("v_mov_b32", t[1], q[0], dpp="wave_shl:1")

    SWAP = dict(op_sel=[0, 1, 0], op_sel_hi=[1, 0, 1])   # src1 halves exchanged

    def push3(self, qr, qm, ql, j, q, t, acc, init=None, late=False):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, qm).sub(0, 2), X, c0, late=late)
        self.fma(ay, WT(j, qm).sub(2, 2), Y, c1, late=late)
        self.fma(ax, WT(j, qr).sub(0, 2), Y, ax, late=late)
        self.fma(ay, WT(j, qr).sub(2, 2), Y, ay, late=late, **self.SWAP)
        self.fma(ay, WT(j, ql).sub(2, 2), X, ay, late=late)
        self.fma(ax, WT(j, ql).sub(0, 2), D, ax, late=late)

    def push_below(self, j, q, t, acc, init=None):
        self.push3(0, 1, 2, j, q, t, acc, init)

    def push_above(self, j, q, t, acc, init=None, late=False):
        self.push3(5, 6, 7, j, q, t, acc, init, late=late)

    def push_self(self, j, q, t, acc, init=None, late=False):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, 3).sub(0, 2), Y, c0, late=late)
        self.fma(ay, WT(j, 3).sub(2, 2), Y, c1, late=late, **self.SWAP)
        self.fma(ay, WT(j, 4).sub(2, 2), X, ay, late=late)
        self.fma(ax, WT(j, 4).sub(0, 2), D, ax, late=late)

    # ring: [slot][lane 0..63][plane 0..9][column 0..3] floats (+ 2 pad per lane record); plane 8 = c', plane 9 = H0
    QUADS = {0: (0, 2), 2: (2, 0), 3: (3, 4), 4: (4, 3), 5: (5, 7), 7: (7, 5)}   # q -> (plane of elements 0,2 ; of elements 1,3)

    def ring_read(self, dst, addr, q, **m):
        """dst quad <- coefficient quad q (or plane 8 / 9 in (c0,c3,c1,c2) order) of the lane's ring record at addr"""
        if "noevlds" in self.ab:
            return
        pa, pb = self.QUADS.get(q, (q, q))
        self.e("ds_read2_b32", dst.sub(0, 2), [addr], offset0=pa * 4 + 0, offset1=pb * 4 + 3, **m)
        self.e("ds_read2_b32", dst.sub(2, 2), [addr], offset0=pa * 4 + 1, offset1=pb * 4 + 2, **m)

    def zero_quad(self, q):
        for k in (0, 3, 2, 1):  # the DPP sources first (VALU -> DPP distance)
            self.mov(q[k], 0)

    def desc_addr(self, dst, qreg, add):
        """dst <- LDS address of the descriptor of stream row (qreg + add)"""
        self.e("s_add_i32", dst, [qreg, add])
        self.e("s_lshl_b32", dst, [dst, 4])
        self.e("s_add_i32", dst, [dst, S_TABB])

    # ---------------------------------------------------------------------------------- events
    def fetch_event(self, ev):
        """LDS reads of the descriptors the event of slot ev needs (issued with the boundary-row reads)"""
        self.desc_addr(T[0], S_QB, ev)
        if self.hist:
            self.mov(V_DN[0], T[0])
            self.e("ds_read_b64", V_DN, [V_DN[0]], offset=8, at=0.0)
        else:
            self.mov(V_DE, T[0])
            self.e("ds_read_b32", V_DE, [V_DE], offset=12, at=0.0)
        self.e("s_add_i32", T[1], [T[0], -32 * DESC_BYTES])
        self.mov(V_DO[0], T[1])
        self.e("ds_read_b64", V_DO, [V_DO[0]], offset=8, at=0.0)

    def fetch_cook(self):
        self.desc_addr(T[2], S_PQ, 0)
        self.mov(V_DC[0], T[2])
        self.e("ds_read_b128", V_DC, [V_DC[0]], at=0.0)

    def take_event(self):
        if self.hist:
            self.e("v_readfirstlane_b32", S_ENB, [V_DN[0]])
            self.e("v_readfirstlane_b32", S_ENF, [V_DN[1]])
        else:
            self.e("v_readfirstlane_b32", S_ENF, [V_DE])
        self.e("v_readfirstlane_b32", S_EO[0], [V_DO[0]])
        self.e("v_readfirstlane_b32", S_EO[1], [V_DO[1]])

    def take_cook(self):
        for k in range(4):
            self.e("v_readfirstlane_b32", S_CD[k], [V_DC[k]])

    def retire(self, j, vq, at_event=True):
        """at_event = False (cfg early): j is a scalar register holding the slot number, OUTQ is loaded already"""
        if self.early and at_event:
            return   # (cfg early: the row's output left at level n, see the end of step() / emit_early_stub)
        eo = S_EO
        lab = self.p.newlabel("noret")
        self.e("s_bitcmp1_b32", (), [S_ACT, j])
        self.e("s_cbranch_scc0", (), [lab])
        self.e("s_bitcmp1_b32", (), [eo[1], F_OWNED])
        self.e("s_cbranch_scc0", (), [lab])
        if self.mband:
            stub, back = self.p.newlabel("mband"), self.p.newlabel("mbback")
            self.e("s_lshr_b32", T[10], [eo[1], 8])           # lo | hi << 12 of the retiring row's band
            self.e("s_cmp_lg_u32", (), [T[10], S_LOHIC])
            self.e("s_cbranch_scc1", (), [stub])
            self.p.label(back)
            self.mstubs.append((stub, back))
        if vq is not None:
            self.mov(OUTQ[0], vq[0])   # registers hold (c0,c3,c1,c2)
            self.mov(OUTQ[1], vq[2])
            self.mov(OUTQ[2], vq[3])
            self.mov(OUTQ[3], vq[1])
        self.e("s_add_u32", T[8], [S_OUT[0], eo[0]])
        self.e("s_addc_u32", T[9], [S_OUT[1], 0])
        self.e("s_mov_b64", EXEC, [S_OMASK])   # the owned columns of the row's band (mband: kept current by the check above)
        if "nostore" not in self.ab:
            self.e("global_store_dwordx4", (), [V_L16, OUTQ, S(T[8].i, 2)], cache=self.cfg.get("st_cache"))
        self.e("s_mov_b64", EXEC, [-1])
        self.p.label(lab)

    def inject(self, j, vq, hn=HN, copy=True):
        for k in self.late_planes(j):
            self.ring_read(WT(j, k), V_RINGE[0], k, at=0.0)
        if copy:   # slots 1..3 use the freshly read quad itself as the row's level-0 value (nothing reads vq before it is
            for i in (1, 0, 2, 3):   # re-initialised); slot 0's deferred tail needs it in the accumulator register
                self.mov(vq[i], hn[i])
        self.e("s_andn2_b32", S_ACT, [S_ACT, 1 << j])
        self.e("s_bitcmp1_b32", (), [S_ENF, F_ACTIVE])
        self.e("s_cselect_b32", T[2], [1 << j, 0])
        self.e("s_cselect_b32", S_AM[j], [-1, 0])
        self.e("s_or_b32", S_ACT, [S_ACT, T[2]])
        self.e("s_cmp_lg_u32", (), [S_ACT, 15])           # vcc != 0 <=> some slot holds a separator / padding row:
        self.e("s_cselect_b64", VCC, [1, 0])              # the per-slot checks of the event-free steps are one branch each
        if j == 3:
            self.e("s_add_i32", S_QB, [S_QB, 32])
        if self.hist:  # where the row's levels go and which lanes own its columns
            self.e("s_bitcmp1_b32", (), [S_ENF, F_OWNED])
            self.e("s_cselect_b64", S_HM[j], [S_OMASK, 0])
            self.e("s_add_u32", S_HB[j][0], [S_HIST[0], S_ENB])
            self.e("s_addc_u32", S_HB[j][1], [S_HIST[1], 0])

    def hist_store(self, j, vq):
        """history mode: the level slot j just completed (1..23), in register order (c0,c3,c1,c2) per 4-column group"""
        self.e("s_mov_b64", EXEC, [S_HM[j]])
        self.e("global_store_dwordx4", (), [V_L16, vq, S_HB[j]])
        self.e("s_mov_b64", EXEC, [-1])
        self.e("s_add_u32", S_HB[j][0], [S_HB[j][0], S_HSTRIDE[0]])
        self.e("s_addc_u32", S_HB[j][1], [S_HB[j][1], S_HSTRIDE[1]])

    # event planes: the coefficient planes of slot j that are dead when the event step starts (the row in the slot only
    # needs its below taps -- and, for slot 0, its above taps -- to finish its last level) can be replaced at the top of
    # the step, together with the boundary-row reads; the others right after the row completed.  Nothing waits mid-step.
    @staticmethod
    def early_planes(j):
        return (3, 4, 8) if j == 0 else (3, 4, 5, 6, 7, 8)

    @staticmethod
    def late_planes(j):
        return (0, 1, 2, 5, 6, 7) if j == 0 else (0, 1, 2)

    def act_check(self, j, vq):
        """a slot holding a separator / padding row is pinned to zero (0 x NaN from a neighbour must not leak into it).
        Round 3: the completed value is ANDed with the slot's 0 / -1 mask, in the steps that have such a row resident only (the
        step body exists twice: see step()).  The round-2 form -- one never-taken branch per slot in every step, a stub that
        zeroes -- cost 0.03 ms of a 0.29 ms forward: ~8 taken branches per step in the ~26 % of steps with such a row (ring
        fill / drain, image boundaries), profiles/r03_perf_notes.md."""
        if self.cfg.get("act_and", True):
            for k in (0, 3, 2, 1):  # the DPP sources first (VALU -> DPP distance)
                self.e("v_and_b32", vq[k], [S_AM[j], vq[k]])
            return
        stub, back = self.p.newlabel("actz"), self.p.newlabel("actb")
        from .isa import I
        exp = [I("s_cbranch_vccnz", (), [stub]), I("label", (), [back])]
        self.e("pseudo", (), (), expand=exp, reads=[("s", S_ACT.i), ("vcc", 0), ("vcc", 1)] + vq.regs(), writes=[("scc", 0)] + vq.regs())
        self.stubs.append((stub, back, vq, j))

    # ---- cfg trace (timing instrumentation, tools/tsw_trace.py): six s_memtime stamps per step at points that are fences of
    # the schedule anyway, written out per wave and step after the barrier.  Not for hist / hin variants (their registers).
    TRACE_REGS = [S(84, 2), S(86, 2), S(88, 2), S(90, 2), S(96, 2), S(98, 2)]
    TRACE_BYTES = 32   # per wave and step: the six stamps (low dwords), the variant number, spare

    def probe(self, k):
        if self.cfg.get("trace", False):
            r = self.TRACE_REGS[k]
            self.e("raw", (), ["s_memtime s[%d:%d]" % (r.i, r.i + 1)])

    def trace_flush(self, c, cook):
        if not self.cfg.get("trace", False):
            return
        e = self.e
        e("raw", (), ["s_waitcnt lgkmcnt(0)"])
        for k, r in enumerate(self.TRACE_REGS):
            if k == 1 and not cook:
                e("raw", (), ["v_writelane_b32 v74, s%d, 1" % self.TRACE_REGS[0].i])   # no cooking in this step: stamp 1 = stamp 0
            else:
                e("raw", (), ["v_writelane_b32 v74, s%d, %d" % (r.i, k)])
        e("raw", (), ["s_movk_i32 s0, %d" % c])
        e("raw", (), ["v_writelane_b32 v74, s0, 6"])
        e("raw", (), ["s_mov_b64 exec, 0xff"])
        e("raw", (), ["global_store_dword v75, v74, s[26:27]"])
        e("raw", (), ["s_mov_b64 exec, -1"])
        e("raw", (), ["s_add_u32 s26, s26, %d" % (NW * self.TRACE_BYTES)])
        e("raw", (), ["s_addc_u32 s27, s27, 0"])

    def tail(self, c, skip_above1=False):
        """the part of step c nobody else waits for (slot 0's pushes after its value was published); emitted at the top of
        the following step, between the boundary-row reads and their wait.  skip_above1: slot 1 is replaced in the
        following step, the accumulator this push would start is re-initialised there"""
        p = c & 1
        v0 = ACC(p, 0)
        self.shift(v0, D_TAIL)
        self.push_self(0, v0, D_TAIL, ACC(p ^ 1, 0), init=WT(0, 8))
        if not skip_above1:
            self.push_above(1, v0, D_TAIL, ACC(p, 1), init=WT(1, 8))

    def step(self, c, slow=False, hi=False):
        """Round 3: two bodies per counter -- the fast one assumes that all four slots hold real rows (vcc == 0) and carries no
        per-slot checks; `slow` (out of line, entered by one branch at the top of the fast body) pins slots that hold a
        separator / padding row to zero.
        cfg stagger (round 4): the loop exists in two flavours; waves 0..3 run the `lo` one (they cook rows 0, 1 of a group, at
        counters = 2 mod 3), waves 4..7 the `hi` one (rows 2, 3, which enter a step later: cooked at counters = 0 mod 3), so
        that the two waves of a SIMD (w and w + 4) never cook -- scalar / LDS / vector-memory instructions, which only issue for
        free beside the partner's VALU work -- in the same step."""
        LS = ".LH" if hi else ".LS"
        act_fast = self.cfg.get("act_and", True) and not self.elastic and "noact" not in self.ab
        p = c & 1
        N1 = [ACC(p, j) for j in range(4)]
        N2 = [ACC(p ^ 1, j) for j in range(4)]
        ev = c if c < 4 else None
        pev = (c - 1) % LV if (c - 1) % LV < 4 else None   # the previous step's event slot: its late planes arrive now
        if "noevents" in self.ab:
            ev = pev = None
        # cooking: the group of four rows that enters from step 3*gamma on is cooked at step 3*gamma - 1 (counter % 3 == 2).
        # stagger: its last two rows (tasks of waves 4..7) only enter at 3*gamma + 1 / + 2, so those waves cook one step
        # later: of the two waves that share a SIMD never both cook (and queue their loads) in the same step
        stag = self.cfg.get("stagger", False)
        assert stag or not hi
        cook = (c % 3 == (0 if hi else 2)) and "nocook" not in self.ab
        g = (((c + 1) // 3) if c % 3 == 2 else (c // 3)) & 1
        # cook_early: the pending task is normalised and written to the ring at the END of the step before (counter % 3 == 1),
        # behind the chain, where every wave but the one with an event has slack before the barrier; the step with
        # counter % 3 == 2 then only requests the next task
        early = self.cfg.get("cook_early", False) and "nocook" not in self.ab
        cook_math_here = cook and not early
        cook_at_end = early and c % 3 == 1

        if slow:
            self.p.label(LS + "s%d_%%=" % c)
        else:
            self.p.label(LS + "%d_%%=" % c)
            if act_fast:
                self.e("s_cbranch_vccnz", (), [LS + "s%d_%%=" % c])
        # an event makes this wave the slowest of the step while the wave it shares its SIMD with has slack: let it issue first
        prio = self.cfg.get("prio", 1) if ev is not None else 0
        if prio:
            self.e("raw", (), ["s_setprio %d" % prio])
        self.probe(0)
        partial = self.cfg.get("partial_wait", True) and not self.cfg.get("trace", False)
        # slim events: the H0 quad read for the row entering slot ev is, one step later, the "row above" of the row entering
        # slot ev + 1 (a wave's four events are consecutive steps): two quads alternate, nothing is read twice; and slots
        # 1..3 use the quad as the row's level-0 value directly
        slim = self.cfg.get("slim_events", True)
        hn, ha = (HN, HA) if (not slim or ev is None or ev % 2 == 0) else (HA, HN)
        # ---- top: everything that travels through LDS is requested first; what the chain needs at once comes first, because
        # the LDS operations of a wave complete in order and the waits below count the requests that may stay outstanding
        if self.elastic:   # the tags first: LDS serves a wave's requests in order, a valid tag proves the data read behind it
            self.tag_reads(c, p, ev)
        if "nolds" not in self.ab:
            self.e("ds_read_b128", BQ, [V_RB[p]], at=0.0)
            self.e("ds_read_b128", TQ, [V_RT[p]], at=0.0)
        if cook:
            self.fetch_cook()
        n_after = 0   # LDS requests behind those the mid-step wait needs
        if ev is not None:
            self.fetch_event(ev)
            n_after += 2
            n_after += self.top_ring_reads(ev, hn, ha, slim)
            if "noevlds" in self.ab:
                n_after = 2
        loads, deferred = [], []
        if cook_math_here:
            # normalise + fold the pending task while the boundary rows arrive (c' and H0 go to the ring at once, the eight
            # coefficient planes from the main region below)
            deferred = self.cook_pending(V_RINGW[g])
            if not ({"nocookwrite", "nocookmath"} & self.ab):
                n_after += 2
            self.probe(1)
        self.tail((c - 1) % LV, skip_above1=(ev == 1))
        if ev == 0:  # slot 0's self taps were still needed by the deferred tail
            for k in self.early_planes(0):
                self.ring_read(WT(0, k), V_RINGE[0], k, at=0.0)
            if "noevlds" not in self.ab:
                n_after += 2 * len(self.early_planes(0))
        self.p.waitcnt(lgkm=min(n_after, 15) if partial else 0)
        self.probe(2)
        if self.elastic:
            self.tag_check(c, p, ev, hn, ha, slim)
        if ev is not None and not partial:
            self.take_event()
        if cook:
            self.take_cook()
            self.issue_prepare(S_CD)
            self.e("s_add_i32", S_PQ, [S_PQ, 4])
            self.ring_writes(deferred, self.cfg.get("rw_at", 0.02), self.cfg.get("rw_at", 0.02) + self.cfg.get("rw_span", 0.5))
            if self.elastic:   # behind the ring writes (LDS keeps a wave's order): this half of the slot is cooked
                self.mov(V_TGR[0], S_TAU)
                self.mov(V_TGR[1], S_RTW[g])
                self.e("ds_write_b32", (), [V_TGR[1], V_TGR[0]], at=0.6)
            loads = self.load_list()
            if self.cfg.get("spread3", False):   # a third of the requests now, the rest in the two following steps
                loads = loads[0::3]
            for i, item in enumerate(loads):
                self.emit_load(item, at=self.cfg.get("load_at", 0.1) + self.cfg.get("load_span", 0.7) * i / len(loads))
            if self.cfg.get("pf", False):
                self.issue_prefetch(at=0.95)
        elif self.cfg.get("spread3", False) and "nocook" not in self.ab:
            # the pending task's remaining requests: the scalar row bases (s2:3, s6:11) and the lane offsets are still in
            # place; the data is consumed at the next step with counter % 3 == 2
            part = self.load_list()[(c % 3) + 1::3]
            for i, item in enumerate(part):
                self.emit_load(item, at=self.cfg.get("load_at", 0.1) + self.cfg.get("load_span", 0.7) * i / max(1, len(part)))
        # received boundary rows
        self.shift(BQ, D_BQ)
        self.push_below(3, BQ, D_BQ, N1[3])
        self.shift(TQ, D_TQ)
        self.push_above(0, TQ, D_TQ, N1[0])
        late = self.cfg.get("late_at") is not None
        for j in (3, 2, 1, 0):
            vq = N1[j]
            tq = D_SLOT.get(j)
            if ev == j:
                if partial:   # the descriptors and the ring planes of the event were requested at the top of the step
                    self.p.waitcnt(lgkm=0)
                    self.take_event()
                self.retire(j, vq)
                self.inject(j, vq, hn, copy=not (slim and j > 0))
                if slim and j > 0:
                    vq = hn
            elif "noact" not in self.ab and (slow or not act_fast):
                self.act_check(j, vq)
            if self.hist and ev != j and ((c - j) % LV) % self.hist_every == 0:   # slot j has just completed level (c - j) mod 24
                self.hist_store(j, vq)
            if self.early and j == 2 and ev != 2:   # (slots 0, 1 keep their values to the end of the step; slot 3's is in the LDS boundary row)
                for i in range(4):
                    self.mov(SAVE2[i], vq[i])
            if j == 3 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR[p], vq], offset=1024, at=0.0)
                if self.elastic:   # the wave below reads this row as its "top": its mailbox, entry 0
                    self.mov(V_TGB[0], S_TAU)
                    self.mov(V_TGB[1], S_TWU[p])
                    self.e("ds_write_b32", (), [V_TGB[1], V_TGB[0]], at=0.0)
            if j == 0 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR[p], vq], offset=0, at=0.0)
                if self.elastic:   # the wave above reads this row as its "bottom": its mailbox, entry 1
                    self.mov(V_TGB[1], S_TWD[p])
                    self.e("ds_write_b32", (), [V_TGB[1], V_TGB[0]], at=0.0)
            if j == 0:
                break  # slot 0's own pushes: tail(), at the top of the next step
            self.shift(vq, tq)
            self.push_below(j - 1, vq, tq, N1[j - 1])
            if ev == j:
                self.push_self(j, vq, tq, N2[j], init=WT(j, 8), late=late)
                self.shift(ha, D_BQ)
                self.push_above(j, ha, D_BQ, N2[j], late=late)
            else:
                self.push_self(j, vq, tq, N2[j], late=late)
            if j < 3:
                self.push_above(j + 1, vq, tq, N1[j + 1], init=WT(j + 1, 8), late=late)
        if cook_at_end:
            self.ring_writes(self.cook_pending(V_RINGW[((c + 2) // 3) & 1]))
        self.probe(3)
        if prio:
            self.e("raw", (), ["s_setprio 0"])
        if self.early:   # did a slot of this wave complete level n in this step?  (4 of the 24 counters; the stub stores the row)
            # The slow body shares the fast body's stub AND its return point: from here on the two bodies do the same (wait, barrier, count,
            # go to the next counter's fast label, which dispatches on vcc again)
            assert not self.cfg.get("trace", False)
            if not slow:
                self.xstub_of[c] = (self.p.newlabel("early"), self.p.newlabel("eback"))
                self.xstubs.append(self.xstub_of[c] + (c, [N1[0], N1[1], SAVE2, None], V_WR[p]))
            stub, back = self.xstub_of[c]
            self.e("s_bitcmp1_b32", (), [S_EMASK, c])
            self.e("s_cbranch_scc1", (), [stub])
            if not slow:
                self.p.label(back)
        if not self.elastic:
            self.p.waitcnt(lgkm=0)
        self.probe(4)
        if "nobar" not in self.ab and not self.elastic:
            self.e("s_barrier")
        self.probe(5)
        self.trace_flush(c, cook)
        self.e("s_sub_u32", S_TAU, [S_TAU, 1])           # S_TAU counts the remaining steps down; the borrow ends the loop
        self.e("s_cbranch_scc1", (), [".Lexit_%="])
        if slow or c == LV - 1:
            self.e("s_branch", (), [LS + "%d_%%=" % ((c + 1) % LV)])

    # ---------------------------------------------------------------------------------- cfg elastic: tags instead of the barrier
    def tag_reads(self, c, p, ev, stub=False):
        m = {} if stub else {"at": 0.0}
        self.mov(V_TGB[0], S_MBR[p])
        self.e("ds_read_b64", V_TGB, [V_TGB[0]], **m)
        if ev is not None:
            self.mov(V_TGR[0], S_RTR)
            self.e("ds_read_b64", V_TGR, [V_TGR[0]], offset=ev * 8, **m)
        if c % 3 == 1:
            # The step after next cooks group g + 2 into the ring slots of group g.  Their last reader is wave g at its counter 2
            # (global step 3g + 2; this is 3g + 4): it must be through with that step.  Its progress is the tag it leaves in the
            # mailbox of the wave above it (entry "bottom", counter parity 0) -- written after every ring read of the step.
            k = (c - 4) // 3   # wave g = wv + k
            self.e("s_add_i32", T[8], [S_WV, (k - 1) % 8])
            self.e("s_and_b32", T[8], [T[8], 7])
            self.e("s_lshl_b32", T[8], [T[8], 4])
            self.e("s_add_i32", T[8], [T[8], S_MB0])
            self.mov(V_DC[0], T[8])
            self.e("ds_read_b32", V_DC[0], [V_DC[0]], offset=4, **m)

    def tag_compare(self, c, ev):
        """scc = 1 <=> a tag is not the expected one.  Boundary rows: written in the step before (S_TAU + 1); ring slot of
        the event: rows entering at counter 0, 1, 2 were cooked in the step before counter 0, the one entering at 3 in
        the step before (with the next wave's group)."""
        a, b = S(T[4].i, 2), S(T[6].i, 2)
        self.e("s_add_i32", S_EXP, [S_TAU, 1])
        self.e("v_cmp_ne_u32", a, [V_TGB[0], S_EXP])
        self.e("v_cmp_ne_u32", b, [V_TGB[1], S_EXP])
        self.e("s_or_b64", a, [a, b])
        if ev is not None:
            self.e("s_add_i32", S_EXR, [S_TAU, 1 if ev == 3 else ev + 1])
            for i in (0, 1):
                self.e("v_cmp_ne_u32", b, [V_TGR[i], S_EXR])
                self.e("s_or_b64", a, [a, b])
        if c % 3 == 1:   # S_TAU counts down: "through with global step 3g + 2" = a tag <= S_TAU + 2
            self.e("s_add_i32", S_EXW, [S_TAU, 2])
            self.e("v_cmp_gt_u32", b, [V_DC[0], S_EXW])
            self.e("s_or_b64", a, [a, b])
        # (the last s_or_b64 left scc = (a != 0))

    def top_ring_reads(self, ev, hn, ha, slim, stub=False):
        """the ring requests an event step makes before its mid-step wait (shared by the step and its retry stub)"""
        m = {} if stub else {"at": 0.0}
        n = 0
        self.e("v_add_u32", V_RINGE[0], [ev * RING_SLOT, V_RINGR])
        self.ring_read(hn, V_RINGE[0], 9, **m)
        n += 2
        if ev > 0:
            if not slim:
                self.e("v_add_u32", V_RINGE[1], [(ev - 1) * RING_SLOT, V_RINGR])
                self.ring_read(ha, V_RINGE[1], 9, **m)
                n += 2
            for k in self.early_planes(ev):
                self.ring_read(WT(ev, k), V_RINGE[0], k, **m)
            n += 2 * len(self.early_planes(ev))
        return n

    def tag_check(self, c, p, ev, hn, ha, slim):
        stub, back = self.p.newlabel("tagw"), self.p.newlabel("tagb")
        self.tag_compare(c, ev)
        self.e("s_cbranch_scc1", (), [stub])
        self.p.label(back)
        self.estubs.append((stub, back, c, p, ev, hn, ha, slim))

    def emit_tag_stub(self, stub, back, c, p, ev, hn, ha, slim):
        """out of line: poll until the neighbours' (and the cooks') tags of this step are there, then repeat the requests the
        step made at its top, whose answers may be stale"""
        loop = self.p.newlabel("tagl")
        self.p.label(stub)
        self.e("s_mov_b32", S_SPIN, [0])
        self.p.label(loop)
        self.e("s_sleep", (), [1])
        self.e("s_add_u32", S_SPIN, [S_SPIN, 1])
        self.e("s_cmp_gt_u32", (), [S_SPIN, SPIN_MAX])
        self.e("s_cbranch_scc1", (), [".Labort_%="])
        self.tag_reads(c, p, ev, stub=True)
        self.p.waitcnt(lgkm=0)
        self.tag_compare(c, ev)
        self.e("s_cbranch_scc1", (), [loop])
        self.e("ds_read_b128", BQ, [V_RB[p]])
        self.e("ds_read_b128", TQ, [V_RT[p]])
        if ev is not None:
            self.top_ring_reads(ev, hn, ha, slim, stub=True)
            if ev == 0:
                for k in self.early_planes(0):
                    self.ring_read(WT(0, k), V_RINGE[0], k)
        self.p.waitcnt(lgkm=0)
        self.e("s_branch", (), [back])

    # ---------------------------------------------------------------------------------- cooking
    def cook_pending(self, ringw):
        """normalise + fold the pending task (inputs in PEND_*, flags in S_PFLAGS) into the ring slot addressed by ringw"""
        g = PEND_G
        norm = self.norm
        l_inact, l_done = self.p.newlabel("cinact"), self.p.newlabel("cdone")
        if "nocookwait" not in self.ab:
            # cfg pf: the prefetch touch issued behind the task's loads is the youngest request and may stay outstanding
            # (loads return in order; a younger store of a retirement can only make this wait longer, never shorter)
            self.p.waitcnt(vm=1 if self.cfg.get("pf", False) and not getattr(self, "_in_prologue", False) else 0)
        if "nocookmath" in self.ab:
            return []
        l_math = self.p.newlabel("cmath")
        self.e("s_bitcmp1_b32", (), [S_PFLAGS, F_PLAIN])   # most rows: nothing to patch
        self.e("s_cbranch_scc1", (), [l_math])
        self.e("s_bitcmp1_b32", (), [S_PFLAGS, F_ACTIVE])
        self.e("s_cbranch_scc0", (), [l_inact])
        if self.sited:
            # rows above / below the image were read from whatever lies there in the tensor (always inside it: the planes
            # read at dy = -1 are channels 5..7, those at dy = +1 channels 0..2): they count as zero
            for flag, chans in ((F_UP, (0, 1, 2)), (F_DN, (5, 6, 7))):
                lab = self.p.newlabel("edge")
                self.e("s_bitcmp1_b32", (), [S_PFLAGS, flag])
                self.e("s_cbranch_scc1", (), [lab])
                for k in chans:
                    self.mov(g[k][0], 0)
                    self.mov(g[k][1], 0)
                self.p.label(lab)
            lab = self.p.newlabel("noedge")
            self.e("s_and_b32", T[0], [S_PFLAGS, (1 << F_FIRST) | (1 << F_LAST)])
            self.e("s_cbranch_scc0", (), [lab])
            self.e("s_bitcmp1_b32", (), [S_PFLAGS, F_FIRST])
            self.e("s_cselect_b64", S_EL, [S_ELC, 0])
            self.e("s_bitcmp1_b32", (), [S_PFLAGS, F_LAST])
            self.e("s_cselect_b64", S_ER, [S_ERC, 0])
            for k in range(8):
                if DX[k] < 0:
                    self.e("v_cndmask_b32", g[k][0], [g[k][0], 0, S_EL])
                if DX[k] > 0:
                    self.e("v_cndmask_b32", g[k][1], [g[k][1], 0, S_ER])
            self.p.label(lab)
        self.p.label(l_math)
        # temporaries
        # (the boundary rows are arriving in TQ / BQ / HN / HA: only TA, TB and OUTQ are free here)
        sx, sy = TA[0], TA[1]
        tt, scale, cc, t2 = TA.sub(2, 2), TB.sub(0, 2), TB.sub(2, 2), OUTQ.sub(0, 2)
        ex, ey = t2[0], t2[1]          # dead before t2 is formed
        mm, om = TA.sub(0, 2), TA.sub(2, 2)  # sparse only: reuse sx,sy / tt once they are dead
        h0 = PEND_BLUR
        if norm == 1:
            for k in range(8):
                self.e("v_and_b32", g[k][0], [0x7fffffff, g[k][0]])
                self.e("v_and_b32", g[k][1], [0x7fffffff, g[k][1]])
        if norm == 3:
            # prenorm: the planes ARE the w_k(p) of reference cspn.py:138; what is left of the fold is the centre term (cspn.py:76)
            self.e("v_pk_add_f32", tt, [g[0], g[1]])
            for k in range(2, 8):
                self.e("v_pk_add_f32", tt, [tt, g[k]])
            self.fma(cc, tt, h0, h0, neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)  # (1 - sigma) * H0
        elif norm != 2:
            self.e("v_add_f32", sx, [g[0][0].abs(), g[1][0].abs()])
            self.e("v_add_f32", sy, [g[0][1].abs(), g[1][1].abs()])
            for k in range(2, 8):
                self.e("v_add_f32", sx, [sx, g[k][0].abs()])
                self.e("v_add_f32", sy, [sy, g[k][1].abs()])
            if norm == 0:
                self.e("v_pk_add_f32", tt, [g[0], g[1]])
                for k in range(2, 8):
                    self.e("v_pk_add_f32", tt, [tt, g[k]])
            self.e("v_rcp_f32", scale[0], [sx])
            self.e("v_rcp_f32", scale[1], [sy])
            if self.cfg.get("newton", False):   # v_rcp_f32 is 1 ulp: one Newton step buys nothing the 1e-4 gate can see
                self.e("v_fma_f32", ex, [-sx, scale[0], 1.0])
                self.e("v_fma_f32", ey, [-sy, scale[1], 1.0])
                self.e("v_fma_f32", scale[0], [ex, scale[0], scale[0]])
                self.e("v_fma_f32", scale[1], [ey, scale[1], scale[1]])
            if norm == 0:
                self.e("v_pk_mul_f32", t2, [tt, scale])
            else:
                self.mov(tt[0], sx)
                self.mov(tt[1], sy)
                self.e("v_pk_mul_f32", t2, [tt, scale])
            self.fma(cc, t2, h0, h0, neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)  # (1 - sigma) * H0
        else:
            self.mov(cc[0], 0)
            self.mov(cc[1], 0)
        if self.sparse:
            # m = sign(sparse) (NaN / 0 pass through), cspn.py:64,81
            for i in (0, 1):
                self.mov(mm[i], PEND_SP[i])
                self.e("v_cmp_gt_f32", S(T[4].i, 2), [PEND_SP[i], 0])
                self.e("v_cndmask_b32", mm[i], [mm[i], 1.0, S(T[4].i, 2)])
                self.e("v_cmp_lt_f32", S(T[6].i, 2), [PEND_SP[i], 0])
                self.e("v_cndmask_b32", mm[i], [mm[i], -1.0, S(T[6].i, 2)])
                self.e("v_sub_f32", om[i], [1.0, mm[i]])
            if not self.given:
                self.e("v_pk_mul_f32", scale, [scale, om])
            else:
                self.mov(scale[0], om[0])
                self.mov(scale[1], om[1])
            self.e("v_pk_mul_f32", t2, [mm, h0])
            self.fma(cc, om, cc, t2, keep=True)
        nw = "nocookwrite" in self.ab
        deferred = []   # the eight coefficient planes are written from the caller's main region, between its FMAs
        for k in range(8):
            if not self.given or self.sparse:
                self.e("v_pk_mul_f32", g[k], [g[k], scale])
            if not nw:
                deferred.append((ringw, g[k], k * 16))
        if self.hist and not self.adj:
            # the folded coefficients are also what the backward's adjoint sweep propagates with: keep a copy of the rows
            # and columns this workgroup owns (planar, [8][B*H*W])
            l_nown = self.p.newlabel("nown")
            self.e("s_bitcmp1_b32", (), [S_PFLAGS, F_OWNED])
            self.e("s_cbranch_scc0", (), [l_nown])
            self.e("s_add_u32", T[8], [S_WF[0], S_PBOFF])
            self.e("s_addc_u32", T[9], [S_WF[1], 0])
            self.e("s_mov_b64", EXEC, [S_CMASK])
            for k in range(8):
                self.e("global_store_dwordx2", (), [V_OFF1, g[k], S(T[8].i, 2)])
                if k < 7:
                    self.e("s_add_u32", T[8], [T[8], S_HSTRIDE[0]])
                    self.e("s_addc_u32", T[9], [T[9], S_HSTRIDE[1]])
            self.e("s_mov_b64", EXEC, [-1])
            self.p.label(l_nown)
        hv = PEND_HIN if self.hin else PEND_BLUR
        if not nw:
            self.ring_w8(ringw, cc, 8 * 16)
            self.ring_w8(ringw, hv, 9 * 16)
        self.e("s_branch", (), [l_done])
        self.p.label(l_inact)
        self.mov(OUTQ[0], 0)
        self.mov(OUTQ[1], 0)
        for k in range(8):
            self.mov(g[k][0], 0)
            self.mov(g[k][1], 0)
        if not nw:
            for k in (8, 9):
                self.ring_w8(ringw, OUTQ.sub(0, 2), k * 16)
        self.p.label(l_done)
        return deferred

    def ring_w8(self, addr, reg, off, **m):
        """8 bytes (a pixel pair) into the ring record at addr + off.  RING_W2 (round 5): as ds_write2_b32 with adjacent offsets instead of ds_write_b64 --
        the same one instruction, but it needs 4-byte alignment only, which lets the records stand an ODD number of dwords apart: the events' one-record-
        per-lane ds_read2_b32 then hit 32 different banks (with the even 42-dword stride they hit 16, twice each), and what 2-way conflicts the stores
        keep is hidden under their operand transfer (MI355X_MICROARCH.md, LDS; tools/r05/tsw_lds_conflicts.py)."""
        if RING_W2:
            assert off % 4 == 0 and off // 4 + 1 < 256
            self.e("ds_write2_b32", (), [addr, reg[0], reg[1]], offset0=off // 4, offset1=off // 4 + 1, **m)
        else:
            self.e("ds_write_b64", (), [addr, reg], offset=off, **m)

    def ring_writes(self, deferred, lo=None, hi=None):
        for i, (addr, reg, off) in enumerate(deferred):
            m = {} if lo is None else {"at": lo + (hi - lo) * i / max(1, len(deferred))}
            self.ring_w8(addr, reg, off, **m)

    def issue_prepare(self, cd):
        """scalar side of a task request: flags and the base addresses of its rows (kept in s0..s11 while the loads are
        issued between the FMAs of the step).  No clamping: a row above / below the image, or the all-zero descriptor of
        an inactive row, still addresses memory inside the tensors (see cook_pending)."""
        self.e("s_mov_b32", S_PFLAGS, [cd[3]])
        if self.hist and not self.adj:
            self.e("s_mov_b32", S_PBOFF, [cd[2]])
        if self.adj:   # planar coefficient planes: a row starts where it starts in a 1-channel tensor.  Plane 0 is read
            # one row up and one pixel left: the base is biased by W4 + 16 bytes (and the lane offsets by the same) so that
            # every lane offset stays non-negative; the caller keeps that much addressable memory in front of the planes
            self.e("s_add_u32", GB_MID[0], [S_GD[0], cd[2]])
            self.e("s_addc_u32", GB_MID[1], [S_GD[1], 0])
            self.e("s_add_i32", T[6], [S_W4, 16])
            self.e("s_sub_u32", GB_MID[0], [GB_MID[0], T[6]])
            self.e("s_subb_u32", GB_MID[1], [GB_MID[1], 0])
        elif self.s8:   # 32 bytes per pixel: the row starts at 8 x its 1-channel byte offset
            self.e("s_lshr_b32", T[7], [cd[2], 29])
            self.e("s_lshl_b32", T[6], [cd[2], 3])
            self.e("s_add_u32", GB_MID[0], [S_GD[0], T[6]])
            self.e("s_addc_u32", GB_MID[1], [S_GD[1], T[7]])
        else:
            self.e("s_add_u32", GB_MID[0], [S_GD[0], cd[0]])
            self.e("s_addc_u32", GB_MID[1], [S_GD[1], cd[1]])
        self.e("s_add_u32", B_BLUR[0], [S_BLUR[0], cd[2]])
        self.e("s_addc_u32", B_BLUR[1], [S_BLUR[1], 0])
        if self.hin:
            self.e("s_add_u32", B_HIN[0], [S_HIN[0], cd[2]])
            self.e("s_addc_u32", B_HIN[1], [S_HIN[1], 0])
        if self.sparse:
            self.e("s_add_u32", B_SP[0], [S_SP[0], cd[2]])
            self.e("s_addc_u32", B_SP[1], [S_SP[1], 0])

    def load_list(self):
        """-> [(op, dst, lane-offset register, scalar base, immediate offset)] of the pending task"""
        if "nocookload" in self.ab:
            return []
        X2, X4 = "global_load_dwordx2", "global_load_dwordx4"
        if self.s8:
            items = [(X4, V(PEND_G[0].i + 4 * i, 4), V_OFFK[0], GB_MID, 16 * i) for i in range(4)]
        else:
            items = [(X2, PEND_G[k], V_OFFK[k], GB_MID, 0) for k in range(8)]
            if "alignedloads" in self.ab:
                items = [(X2, PEND_G[k], V_OFF1, GB_MID, 0) for k in range(8)]
            if "halfloads" in self.ab:
                items = items[:4]
            if "sameload" in self.ab:
                items = [(X2, PEND_G[k], V_OFFK[k], S_GD, 0) for k in range(8)]
        items.append((X2, PEND_BLUR, V_OFF1, B_BLUR, 0))
        if self.hin:
            items.append((X2, PEND_HIN, V_OFF1, B_HIN, 0))
        if self.sparse:
            items.append((X2, PEND_SP, V_OFF1, B_SP, 0))
        return items

    def emit_load(self, item, **m):
        op, dst, voff, base, off = item
        if self.cfg.get("ld32", False) and op == "global_load_dwordx2":   # experiment: two 4-byte loads instead of one 8-byte load
            for i in (0, 1):
                self.e("global_load_dword", dst[i], [voff, base], cache=self.cfg.get("ld_cache"), offset=off + 4 * i, **m)
            return
        if off:
            m["offset"] = off
        self.e(op, dst, [voff, base], cache=self.cfg.get("ld_cache"), **m)

    def issue_loads(self, items):
        for it in items:
            self.emit_load(it)

    def issue_prefetch(self, **m):
        """cfg pf: the task after the one just requested (descriptor in S_CD) reads rows 4 further down the stream: touch its
        cache lines now (rows of another image / beyond the share: a wasted touch, clamped into the tensor)"""
        lead = self.cfg.get("pf_rows", 4)
        self.e("s_mul_i32", T[6], [S_W4, lead])
        self.e("s_add_u32", T[6], [T[6], S_CD[0]])
        self.e("s_min_u32", T[6], [T[6], S_GLAST])
        self.e("s_add_u32", S_PFB[0], [S_GD[0], T[6]])
        self.e("s_addc_u32", S_PFB[1], [S_GD[1], 0])
        self.e("global_load_dword", V_PFD, [V_PF, S_PFB], dummy=True, **m)

    def issue_task(self, cd, first_third=False):
        self.issue_prepare(cd)
        self.issue_loads(self.load_list()[0::3] if first_third else self.load_list())

    # ---------------------------------------------------------------------------------- prologue
    def prologue(self):
        e = self.e
        self._in_prologue = True
        # LDS below the descriptor table (boundary rows, ring) was zeroed by the C++ part of the kernel (cspn2d_tsw.hip)
        e("v_lshlrev_b32", V_L16, [4, V_LANE])
        e("v_lshlrev_b32", V_COL4, [2, V_LANE])
        # state
        for r in range(ACC_BASE, WT_BASE + 144):
            self.mov(V(r), 0)
        for i in range(4):   # wave 7 enters the loop at its slot-3 event: the "row above" quad of that first event
            self.mov(HN[i], 0)
            self.mov(HA[i], 0)
        for k in range(8):
            self.mov(PEND_G[k][0], 0)
            self.mov(PEND_G[k][1], 0)
        for q in (PEND_BLUR, PEND_HIN, PEND_SP):
            self.mov(q[0], 0)
            self.mov(q[1], 0)
        e("s_mov_b32", S_TAU, [S_LAST])
        e("s_mov_b32", S_ACT, [0])
        for j in range(4):
            e("s_mov_b32", S_AM[j], [0])
        e("s_mov_b64", VCC, [1])
        if self.mband:   # the first retirement derives the mask (no descriptor has lo | hi << 12 == -1)
            e("s_mov_b32", S_LOHIC, [-1])
            e("s_mov_b64", S_OMASK, [0])
        else:
            e("s_and_b32", T[2], [S_LOHI, 0xffff])
            e("s_lshr_b32", T[3], [S_LOHI, 16])
            e("v_cmp_ge_u32", S(T[4].i, 2), [V_COL4, T[2]])
            e("v_cmp_lt_u32", S(T[6].i, 2), [V_COL4, T[3]])
            e("s_and_b64", S_OMASK, [S(T[4].i, 2), S(T[6].i, 2)])
        if self.hist:
            for j in range(4):
                e("s_mov_b64", S_HM[j], [0])
                e("s_mov_b64", S_HB[j], [S_HIST])
            # cooking lanes inside the owned columns: 4*xb (V_OFF1, set below) against 4*lo, 4*hi -- computed after V_OFF1
            npl = LV // self.hist_every - 1   # level planes in front of the coefficient planes
            e("s_mul_i32", T[2], [S_HSTRIDE[0], npl])
            e("s_mul_hi_u32", T[3], [S_HSTRIDE[0], npl])
            e("s_mul_i32", T[4], [S_HSTRIDE[1], npl])
            e("s_add_i32", T[3], [T[3], T[4]])
            e("s_add_u32", S_WF[0], [S_HIST[0], T[2]])
            e("s_addc_u32", S_WF[1], [S_HIST[1], T[3]])
        e("s_and_b32", T[1], [S_WV, 1])            # T1 = wave parity (= cooking half)
        e("s_lshr_b32", T[2], [S_WV, 1])           # T2 = wv >> 1
        # boundary exchange addresses
        for p in (0, 1):
            # buffer written in phase parity p: p ^ (wv & 1)
            e("s_xor_b32", T[3], [T[1], p])
            e("s_lshl_b32", T[4], [T[3], 14])          # written buffer * 16384
            e("s_xor_b32", T[5], [T[4], 16384])        # read buffer
            e("s_add_i32", T[4], [T[4], S_LDSB])
            e("s_add_i32", T[5], [T[5], S_LDSB])
            e("s_lshl_b32", T[6], [S_WV, 11])
            e("s_add_i32", T[7], [T[4], T[6]])
            e("v_add_u32", V_WR[p], [T[7], V_L16])
            e("s_add_i32", T[6], [S_WV, 7])
            e("s_and_b32", T[6], [T[6], 7])
            e("s_lshl_b32", T[6], [T[6], 11])
            e("s_add_i32", T[7], [T[5], T[6]])
            e("s_add_i32", T[7], [T[7], 1024])
            e("v_add_u32", V_RT[p], [T[7], V_L16])
            e("s_add_i32", T[6], [S_WV, 1])
            e("s_and_b32", T[6], [T[6], 7])
            e("s_lshl_b32", T[6], [T[6], 11])
            e("s_add_i32", T[7], [T[5], T[6]])
            e("v_add_u32", V_RB[p], [T[7], V_L16])
        if self.elastic:
            e("s_add_i32", T[8], [S_LDSB, LDS_TAGS])
            e("s_mov_b32", S_MB0, [T[8]])
            for p in (0, 1):
                e("s_lshl_b32", T[9], [S_WV, 4])                 # own mailbox, counter parity p: [top, bottom]
                e("s_add_i32", T[9], [T[9], T[8]])
                e("s_add_i32", S_MBR[p], [T[9], p * 8])
                for reg, dw, ent in ((S_TWU, 1, 0), (S_TWD, 7, 4)):   # the wave below reads my row as "top", the wave above as "bottom"
                    e("s_add_i32", T[9], [S_WV, dw])
                    e("s_and_b32", T[9], [T[9], 7])
                    e("s_lshl_b32", T[9], [T[9], 4])
                    e("s_add_i32", T[9], [T[9], T[8]])
                    e("s_add_i32", reg[p], [T[9], p * 8 + ent])
            e("s_add_i32", T[8], [T[8], TAG_RT])
            e("s_lshl_b32", T[9], [T[1], 5])                     # the event side reads the four slots of the wave's half
            e("s_add_i32", S_RTR, [T[8], T[9]])
            for x in (0, 1):                                     # the cooking side: same slot arithmetic as V_RINGW[x]
                e("s_xor_b32", T[9], [T[1], x])
                e("s_lshl_b32", T[9], [T[9], 2])
                e("s_add_i32", T[9], [T[9], T[2]])
                e("s_add_i32", T[9], [T[9], 7])
                e("s_and_b32", T[9], [T[9], 7])
                e("s_lshl_b32", T[9], [T[9], 3])
                e("s_add_i32", T[9], [T[9], T[8]])
                e("s_lshl_b32", T[10], [T[1], 2])
                e("s_add_i32", S_RTW[x], [T[9], T[10]])
            # what the first step expects to find: "written in the step before" = S_LAST + 1, in the mailboxes of the two
            # neighbours for the counter parity they start with ((wv + 1) & 1, the same for both)
            # (the other parity's entries are written during step 0; until then they say "not yet" to everybody)
            e("s_add_i32", T[10], [S_LAST, 1])
            self.mov(V_TGB[0], T[10])
            self.mov(V_TGR[0], 0x7fffffff)
            e("s_cmp_eq_u32", (), [T[1], 0])
            for reg in (S_TWU, S_TWD):
                e("s_cselect_b32", T[9], [reg[1], reg[0]])
                self.mov(V_TGB[1], T[9])
                e("ds_write_b32", (), [V_TGB[1], V_TGB[0]])
                e("s_cselect_b32", T[9], [reg[0], reg[1]])
                self.mov(V_TGR[1], T[9])
                e("ds_write_b32", (), [V_TGR[1], V_TGR[0]])
        # ring read base: LDS_RING + (wv&1)*4*RING_SLOT + lane*RING_REC
        e("s_mul_i32", T[3], [T[1], 4 * RING_SLOT])
        e("s_add_i32", T[3], [T[3], LDS_RING])
        e("s_add_i32", T[3], [T[3], S_LDSB])
        e("v_mul_u32_u24", V_RINGR, [RING_REC, V_LANE])
        e("v_add_u32", V_RINGR, [T[3], V_RINGR])
        # ring write addresses: slot(g) = (4g + (wv>>1) - 1) & 7; the cooking lane holds pixels 2*lane, 2*lane + 1 of half
        # wv&1 of the row = columns 2*(lane&1), +1 of row lane 32*(wv&1) + (lane>>1): byte that lane * RING_REC + (lane&1)*8
        # inside a slot (+ plane * 16); V_RINGW[x] serves gamma parity x ^ (wv & 1)
        e("v_and_b32", V_TMP, [1, V_LANE])
        e("v_lshlrev_b32", V_TMP, [3, V_TMP])
        e("v_lshrrev_b32", CK[5], [1, V_LANE])
        e("v_mul_u32_u24", CK[5], [RING_REC, CK[5]])
        e("v_add_u32", V_TMP, [V_TMP, CK[5]])
        e("s_mul_i32", T[3], [T[1], 32 * RING_REC])
        e("v_add_u32", V_TMP, [T[3], V_TMP])
        for x in (0, 1):
            e("s_xor_b32", T[3], [T[1], x])            # gamma parity
            e("s_lshl_b32", T[3], [T[3], 2])
            e("s_add_i32", T[3], [T[3], T[2]])
            e("s_add_i32", T[3], [T[3], 7])            # -1 mod 8
            e("s_and_b32", T[3], [T[3], 7])
            e("s_mul_i32", T[3], [T[3], RING_SLOT])
            e("s_add_i32", T[3], [T[3], LDS_RING])
            e("s_add_i32", T[3], [T[3], S_LDSB])
            e("v_add_u32", V_RINGW[x], [T[3], V_TMP])
        # global offsets of the lane's pixel pair: 4*xb = 512*(wv&1) + 8*lane
        e("v_lshlrev_b32", V_OFF1, [3, V_LANE])
        e("s_lshl_b32", T[3], [T[1], 9])
        e("v_add_u32", V_OFF1, [T[3], V_OFF1])
        if self.s8:   # the lane's pixel pair inside a row of 64-byte pair records: 64 * (64 * (wv&1) + lane)
            e("v_lshlrev_b32", V_OFFK[0], [6, V_LANE])
            e("s_lshl_b32", T[3], [T[1], 12])
            e("v_add_u32", V_OFFK[0], [T[3], V_OFFK[0]])
        for k in ([] if self.s8 else range(8)):
            e("s_mul_i32", T[3], [S_HW4, (7 - k) if self.adj else k])   # S_HW4: bytes from one guidance plane to the next
            if self.sited:
                if DY[k] > 0:
                    e("s_add_i32", T[3], [T[3], S_W4])
                if DY[k] < 0:
                    e("s_sub_i32", T[3], [T[3], S_W4])
                if DX[k] != 0 and "aligned2" not in self.ab:
                    e("s_add_i32", T[3], [T[3], 4 * DX[k]])
            if self.adj:
                e("s_add_i32", T[3], [T[3], S_W4])
                e("s_add_i32", T[3], [T[3], 16])
            e("v_add_u32", V_OFFK[k], [T[3], V_OFF1])
        if self.cfg.get("pf", False):
            # L2 prefetch by touching: lane l = 8*k + j touches cache line j (0..4) of the half-row plane k of a future task
            # reads (512 bytes starting up to 4 bytes before / after the row base: five 128-byte lines cover it)
            assert not self.hist
            e("v_lshrrev_b32", CK[5], [3, V_LANE])                 # k
            e("v_and_b32", CK[6], [7, V_LANE])                     # j
            e("v_cmp_lt_u32", S(T[4].i, 2), [CK[6], 4])
            e("v_cndmask_b32", CK[6], [4, CK[6], S(T[4].i, 2)])     # min(j, 4)
            e("v_lshlrev_b32", CK[6], [7, CK[6]])                   # * 128
            e("v_mov_b32", V_PF, [0])
            for k in range(8):
                e("s_mul_i32", T[3], [S_HW4, (7 - k) if self.adj else k])
                if self.sited:
                    if DY[k] > 0:
                        e("s_add_i32", T[3], [T[3], S_W4])
                    if DY[k] < 0:
                        e("s_sub_i32", T[3], [T[3], S_W4])
                    if k > 0:
                        e("s_sub_i32", T[3], [T[3], 4])
                e("v_cmp_eq_u32", S(T[4].i, 2), [CK[5], k])
                e("v_mov_b32", CK[8], [T[3]])
                e("v_cndmask_b32", V_PF, [V_PF, CK[8], S(T[4].i, 2)])
            e("v_add_u32", V_PF, [V_PF, CK[6]])
            e("s_lshl_b32", T[3], [T[1], 9])                        # half * 512
            e("v_add_u32", V_PF, [T[3], V_PF])
            e("v_mov_b32", V_PFD, [0])
        if self.hist:
            e("s_and_b32", T[2], [S_LOHI, 0xffff])
            e("s_lshl_b32", T[2], [T[2], 2])
            e("s_lshr_b32", T[3], [S_LOHI, 16])
            e("s_lshl_b32", T[3], [T[3], 2])
            e("v_cmp_ge_u32", S(T[4].i, 2), [V_OFF1, T[2]])
            e("v_cmp_lt_u32", S(T[6].i, 2), [V_OFF1, T[3]])
            e("s_and_b64", S_CMASK, [S(T[4].i, 2), S(T[6].i, 2)])
        # constant edge-lane masks
        e("s_cmp_eq_u32", (), [T[1], 0])
        e("s_cselect_b32", S_ELC[0], [1, 0])
        e("s_mov_b32", S_ELC[1], [0])
        e("s_mov_b32", S_ERC[0], [0])
        e("s_cselect_b32", S_ERC[1], [0, 0x80000000])
        # ring counters: wave 7's slot 3 fires at step 0 for the (inactive) row -1
        e("s_lshl_b32", S_QB, [S_WV, 2])
        e("s_cmp_eq_u32", (), [S_WV, 7])
        e("s_cselect_b32", S_QB, [-4, S_QB])
        # the workgroup's descriptor table: already in LDS (written by the C++ part of the kernel before this block,
        # cspn2d_tsw.hip: tsw_fill_table) -- or, cfg tab_in_lds = False, copied from global memory here (512 threads x 16 bytes
        # per sweep)
        e("s_add_i32", T[3], [S_LDSB, LDS_TAB])
        e("s_add_i32", S_TABB, [T[3], PADF * DESC_BYTES])
        if not self.cfg.get("tab_in_lds", True):
            e("s_lshl_b32", T[2], [S_WV, 10])
            e("v_add_u32", V_TMP, [T[2], V_L16])              # tid * 16
            e("v_add_u32", CK[4], [T[3], V_TMP])              # LDS destination
            e("s_mov_b64", S(T[4].i, 2), [S_PLAN])
            l_copied = self.p.newlabel("copied")
            for i in range(TAB_MAX_ROWS // 512):
                e("s_cmp_gt_u32", (), [S_NROWS, i * 512])
                e("s_cbranch_scc0", (), [l_copied])
                e("global_load_dwordx4", CK.sub(0, 4), [V_TMP, S(T[4].i, 2)])
                e("s_add_u32", T[4], [T[4], 8192])
                e("s_addc_u32", T[5], [T[5], 0])
                self.p.waitcnt(vm=0)
                e("ds_write_b128", (), [CK[4], CK.sub(0, 4)], offset=i * 8192)
            self.p.label(l_copied)
        self.p.waitcnt(lgkm=0)
        e("s_barrier")                               # LDS zero-fill and the table are complete
        # cooking pipeline: task n = row 4n - 1 + (wv>>1), half wv&1.  Task 0 is cooked synchronously, task 1 requested.
        e("s_lshr_b32", T[2], [S_WV, 1])
        e("s_add_i32", S_PQ, [T[2], -1])
        self.fetch_cook()
        self.p.waitcnt(lgkm=0)
        self.take_cook()
        self.issue_task(S_CD)
        e("s_add_i32", S_PQ, [S_PQ, 4])
        self.fetch_cook()
        # gamma = 0 has parity 0: ring address V_RINGW[0 ^ (wv & 1)]
        e("s_and_b32", T[1], [S_WV, 1])
        e("s_cmp_eq_u32", (), [T[1], 0])
        e("s_cselect_b64", S(T[4].i, 2), [-1, 0])
        e("v_cndmask_b32", V_TMP, [V_RINGW[1], V_RINGW[0], S(T[4].i, 2)])
        l_late = None
        if self.cfg.get("stagger", False):
            # waves 4..7 cook one step after the others: their task 0 is cooked by the first step of the loop
            l_late = self.p.newlabel("late")
            e("s_bitcmp1_b32", (), [S_WV, 2])
            e("s_cbranch_scc1", (), [l_late])
        self.ring_writes(self.cook_pending(V_TMP))
        if self.elastic:   # task 0 counts as cooked in the step before the first one
            e("s_add_i32", T[10], [S_LAST, 1])
            self.mov(V_TGR[0], T[10])
            e("s_and_b32", T[9], [S_WV, 1])
            e("s_cmp_eq_u32", (), [T[9], 0])
            e("s_cselect_b32", T[9], [S_RTW[0], S_RTW[1]])
            self.mov(V_TGR[1], T[9])
            e("ds_write_b32", (), [V_TGR[1], V_TGR[0]])
        self.p.waitcnt(lgkm=0)
        self.take_cook()
        self.issue_task(S_CD, first_third=self.cfg.get("spread3", False))   # the loop's first two steps request the rest
        if self.cfg.get("pf", False):
            self.issue_prefetch()
        e("s_add_i32", S_PQ, [S_PQ, 4])
        if l_late:
            self.p.label(l_late)
        self.p.waitcnt(lgkm=0)
        e("s_barrier")
        if self.cfg.get("trace", False):
            assert not self.hist and not self.hin and not self.cfg.get("pf", False)
            e("s_mul_i32", T[3], [S_WV, self.TRACE_BYTES])
            e("v_lshlrev_b32", V(75), [2, V_LANE])
            e("v_add_u32", V(75), [T[3], V(75)])
            e("v_mov_b32", V(74), [0])
        self._in_prologue = False
        for w in range(NW):
            c0 = (LV - 3 * w) % LV
            e("s_cmp_eq_u32", (), [S_WV, w])
            e("s_cbranch_scc1", (), [(".LH" if self.cfg.get("stagger", False) and w >= NW // 2 else ".LS") + "%d_%%=" % c0])

    def emit_early_stub(self, stub, back, c, vqs, v_wr):
        """cfg early, end of the step with ring counter c: slot j has just completed level (c - j) mod 24; the one whose level is n (S_NIT) is stored
        exactly as a retirement would store it -- its descriptor is fetched again from the LDS table (stream row S_QB + j, or that - 32 once
        slot 3's injection of this cycle has moved S_QB on), so the linear plan's band changes (mband) work unchanged"""
        self.p.label(stub)
        cand = [j for j in range(4) if (c - j) % LV != 0]   # (level 0 = the slot had its event in this step: never an early output)
        sel = {j: self.p.newlabel("esel") for j in cand}
        common = self.p.newlabel("ecommon")
        for j in cand:
            self.e("s_cmp_eq_u32", (), [S_NIT, (c - j) % LV])
            self.e("s_cbranch_scc1", (), [sel[j]])
        self.e("s_branch", (), [back])
        for j in cand:
            self.p.label(sel[j])
            vq = vqs[j]
            if j == 3:   # published as the wave's last row in this step (LDS operations of a wave complete in order)
                self.e("ds_read_b128", TQ, [v_wr], offset=1024)
                self.p.waitcnt(lgkm=0)
                vq = TQ
            self.mov(OUTQ[0], vq[0])   # registers hold (c0,c3,c1,c2)
            self.mov(OUTQ[1], vq[2])
            self.mov(OUTQ[2], vq[3])
            self.mov(OUTQ[3], vq[1])
            self.e("s_mov_b32", T[0], [j])
            self.e("s_add_i32", T[1], [S_QB, j if j < c < 3 else j - 32])
            self.e("s_branch", (), [common])
        self.p.label(common)
        self.e("s_lshl_b32", T[1], [T[1], 4])
        self.e("s_add_i32", T[1], [T[1], S_TABB])
        self.mov(V_DO[0], T[1])
        self.e("ds_read_b64", V_DO, [V_DO[0]], offset=8)
        self.p.waitcnt(lgkm=0)
        self.e("v_readfirstlane_b32", S_EO[0], [V_DO[0]])
        self.e("v_readfirstlane_b32", S_EO[1], [V_DO[1]])
        self.retire(T[0], None, at_event=False)
        self.e("s_branch", (), [back])

    def build(self):
        self.prologue()
        stag = self.cfg.get("stagger", False)
        for c in range(LV):
            self.step(c)
        if stag:
            for c in range(LV):
                self.step(c, hi=True)
        self.p.label(".Lexit_%=")
        self.e("s_branch", (), [".Lend_%="])
        if self.cfg.get("act_and", True) and not self.elastic and "noact" not in self.ab:
            for c in range(LV):
                self.step(c, slow=True)
            if stag:
                for c in range(LV):
                    self.step(c, slow=True, hi=True)
        for st in self.estubs:
            self.emit_tag_stub(*st)
        if self.elastic:
            self.p.label(".Labort_%=")   # a tag never came (cannot happen unless a wave died): leave instead of hanging
            self.e("s_endpgm")
        for stub, back, vq, j in self.stubs:
            self.p.label(stub)
            self.e("s_bitcmp1_b32", (), [S_ACT, j])
            self.e("s_cbranch_scc1", (), [back])
            self.zero_quad(vq)
            self.e("s_branch", (), [back])
        for st in self.xstubs:
            self.emit_early_stub(*st)
        for stub, back in self.mstubs:
            # the band changed: S_OMASK <- lanes [lo / 4, hi / 4) of the new band (T[10] = lo | hi << 12, from retire())
            self.p.label(stub)
            self.e("s_mov_b32", S_LOHIC, [T[10]])
            self.e("s_and_b32", T[11], [T[10], 0xfff])
            self.e("s_lshr_b32", T[11], [T[11], 2])
            self.e("s_lshr_b32", T[10], [T[10], 14])
            self.e("s_sub_u32", T[10], [T[10], T[11]])
            self.e("s_bfm_b64", S_OMASK, [T[10], T[11]])      # ((1 << width) - 1) << first; width 64 wraps to 0:
            self.e("s_cmp_eq_u32", (), [T[10], 64])
            self.e("s_cselect_b64", S_OMASK, [-1, S_OMASK])
            self.e("s_branch", (), [back])
        self.p.label(".Lend_%=")
        return self.p


def build(cfg, sched=True):
    from . import isa
    assert (LDS_TAB - LDS_TAGS) % 512 == (256 if cfg.get("elastic", False) else 0), "configure(tag_area) must match cfg elastic"
    isa.SOFT_VALU_LATENCY = cfg.get("soft_lat", 1)
    g = Gen(cfg)
    p = g.build()
    if sched:
        schedule(p)
    expand_pseudos(p)
    errs = check_hazards(p)
    if errs:
        raise RuntimeError("hazards:\n" + "\n".join(errs[:20]))
    return p
