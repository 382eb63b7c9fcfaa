"""tools/tswgen/kernel4.py -- generator of the round-6 main loop of the fused CSPN kernel: a ring of 12 waves x 3 resident rows
(3 waves per SIMD at <= 168 VGPRs) instead of kernel.py's 8 waves x 4 rows (2 per SIMD at 256).

What stays (kernel.py / DESIGN.md 3.1b): 4 columns per lane in the pairing X = (c0,c3), Y = (c1,c2); every resident row advances one
CSPN iteration (reference cspn_pytorch/models/cspn.py:66-81) per step; rows of a wave sit on a one-level staircase, the last row of wave w
and the first of wave w + 1 at the same level, so that 12 x (3 - 1) = 24 levels -- the whole forward -- are in flight; boundary rows
through LDS, one s_barrier per step; 24 unrolled phases of a wave's ring counter c = (tau - 2 wave) mod 24.

What is new:
  * 36 resident rows, 1.5 rows enter per step (stream row q lives in wave (q div 3) mod 12, slot q mod 3, enters at step
    2 (q div 3) + q mod 3): 8/9 of the steps of the 8 x 4 ring for the same rows, and a third wave per SIMD to issue from while
    the other two wait for their boundary rows.
  * 36 working registers instead of 80.  No value of the feed ever sits in a VGPR longer than one task:
      - raw rows (8 guidance planes neighbour-sited, blur, sparse) arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per piece = one
        plane of one band row: lane L's 16 bytes are the 4 columns of consumer lane L), FIVE steps before the row enters, into a pool
        of 9 row slots of 11 KiB (slot = stream row mod 9; the slot is free exactly then: its previous row entered six steps before);
      - ONE wave per row computes, in the step before the row enters, what normalise + fold (cspn.py:85-144, :76, :81) needs per PIXEL
        -- scale = (1 - m) / sum_k |G_k| and c' -- from the plane-pure quads (8 + 2 ds_read_b128, 2 ds_write_b128, ~55 VALU for 256
        pixels), leaving the raw planes where they are;
      - the event reads the 8 raw planes, c', level 0 and the scale quad with eleven ds_read_b128 straight into the slot's registers,
        turns the plane-pure quads into the mixed (dx = +1 | dx = -1) quads the packed FMAs want with two v_swap_b32 per pair of planes,
        and multiplies by the scale there (16 v_pk_mul_f32).  First version of this loop (profiles/r06_perf_notes.md): four 1-pixel-
        per-lane tasks per row cooked the rows completely in LDS -- 2.1x the instructions, and as unbalanced as the barrier is
        unforgiving.
  * Values and accumulators are quads (c0, c3, c2, c1): a natural quad with elements 1 and 3 exchanged, pairs X = (c0,c3), Y = (c2,c1).
  * Every wave has a ROLE per step, static per ring counter (at any step the 12 waves hold the 12 counters of the step's parity), and
    the roles are placed so that each of the four SIMDs (waves w, w + 4, w + 8: counters c, c + 8, c + 16) carries exactly one heavy
    job per step: an event (c = 0, 1, 2), a row to cook (c = 11, 12, 13) or a row to request (10 DMA pieces: c = 7, 14, 15).
  * Slot addressing is static: 36 = 0 (mod 9), so a wave's own rows always use the same three slots, and the rows a role works on
    are a static number of groups ahead: a per-wave table of three slot-group addresses (S_SL3) covers every case without a
    run-time modulo.
  * The resident rows' output offset and flags live in SGPRs from injection to retirement (no descriptor re-fetch).
Variants: norm (0 '8sum', 1 '8sum_abs', 2 'none', 3 'prenorm') x sparse; passes of exactly 24 iterations, first pass only
(continuation passes, n_iter != 24, history / adjoint sweeps stay on kernel.py's loop).
"""
from .isa import Prog, V, S, R, EXEC, VCC, M0, I, schedule, check_hazards, expand_pseudos

NW, NSLOT, LV = 12, 3, 24
LEAD = 6                     # the loop starts at step -LEAD: the feed pipeline primes itself on inactive rows
PADF, PADB = 16, 52          # inactive descriptor rows before / after a workgroup's stream
DESC_BYTES = 16              # goff_lo, goff_hi, boff, flags | lo << 8 | hi << 20   (cspn2d_tsw_desc.h)
F_ACTIVE, F_UP, F_DN, F_FIRST, F_LAST, F_OWNED, F_PLAIN = 0, 1, 2, 3, 4, 5, 6

SLOT, NR = 11264, 9          # a row slot: planes 0..7 raw guidance, 8 blur (level 0), 9 sparse -> scale, 10 c'
P_H0, P_SC, P_CC = 8, 9, 10
BND_BUF = NW * 2048          # one boundary buffer: per wave its top row (slot 0's value) and its bottom row (slot 2's), 1 KiB each
LDS_BND, LDS_ROWS = 0, 2 * BND_BUF
LDS_TAB = LDS_ROWS + NR * SLOT
TAB_MAX_ROWS = (163840 - LDS_TAB) // DESC_BYTES
LDS_BYTES = LDS_TAB + TAB_MAX_ROWS * DESC_BYTES
assert LDS_BYTES == 163840 and 36 % NR == 0

DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]
PAIRS = [(0, 2), (3, 4), (5, 7)]     # (dx = +1, dx = -1) planes of a row of taps; planes 1, 6: dx = 0

# ---- VGPR map: 36 working registers + 24 accumulators + 108 coefficients = 168 ----
V_LANE = V(0)
V_L16, V_WR, V_EV, V_RB = V(0), V(1), V(6), V(7)
SQX, SQY = V(2, 4), V(32, 4)   # scale quads of the entering rows (slots 0, 2 / slot 1): alive from the top of the event step to the top of the next
BQ, TQ, HN, HA = V(8, 4), V(12, 4), V(16, 4), V(20, 4)
D_A, D_B, D_C, D_EV = V(24, 2), V(26, 2), V(28, 2), V(30, 2)
V_D = V(30, 2)               # descriptor fetch of an event (dead before D_EV is formed)
# shifted pairs D = (c3 of lane-1, c0 of lane+1): the deferred tail and slot 2 share D_A, the row below and slot 1 D_B, the row above D_C;
# the entering row's "row above" (event steps) D_EV
D_TAIL, D_BQ, D_TQ, D_SLOT = D_A, D_B, D_C, {2: D_A, 1: D_B}
# a row's cooking task (c = 11, 12, 13: never an event step): sums in SQY / SQX, raw quads cycle through HN, HA, BQ, TQ (the latter two
# once the chain has consumed the boundary rows), flags and the lane address in v30 / v31
CK_S, CK_T, CK_BUF, CK_F, CK_A = SQY, SQX, [HN, HA, BQ, TQ], V(30), V(31)
ACC_BASE, WT_BASE = 36, 60


def ACC(p, j):
    return V(ACC_BASE + (p * NSLOT + j) * 4, 4)


def WT(j, k):
    return V(WT_BASE + (j * 9 + k) * 4, 4)


# ---- SGPR map (inputs: cspn2d_tsw4.hip) ----
S_GD, S_BLUR, S_HIN, S_SP, S_OUT, S_AUX = S(16, 2), S(18, 2), S(20, 2), S(22, 2), S(24, 2), S(26, 2)
S_W4, S_HW4, S_LAST, S_WV = S(28), S(29), S(30), S(31)
S_LDSB = S(15)
S_LOHIC, S_OMASK = S(0), S(42, 2)
S_P = [S(2, 2), S(4, 2), S(6, 2), S(8, 2)]      # source base of a DMA piece (four in rotation)
S_DD = [S(10), S(11), S(12)]                    # descriptor dwords 0..2 of the row a DMA role requests (goff_lo, goff_hi, boff)
S_CF = S(14)                                    # flags of the row a cooking task works on
S_QB, S_QTA, S_TABB = S(34), S(35), S(37)       # stream row of the wave's slot 0 (current cycle); LDS address of the descriptor of row 3 floor(tau / 2); of row 0
S_M0L, S_M63 = S(38, 2), S(40, 2)               # lane masks: lane 0 / lane 63
S_NB, S_NF = S(46), S(44)                       # entering row: descriptor dwords 2, 3
S_TAU = S(45)
S_SB = [S(52), S(53), S(54)]                    # per slot: the resident row's output byte offset
S_AM = [S(56), S(57), S(58)]                    # per slot: -1 real row, 0 separator / padding row
S_SF = [S(59), S(60), S(61)]                    # per slot: the resident row's descriptor dword 3
S_SL3 = [S(62), S(63), S(64)]                   # LDS address of the slot group (3 slots) of row group (this wave's + i) mod 3
S_RT = S(65)                                    # LDS address (buffer 0, lane 0) of the row above this wave's rows
T = [S(68 + i) for i in range(12)]
S_GDK = [S(80 + 2 * k, 2) for k in range(8)]    # guidance base + plane k's (sited) offset
TRACE_REGS = [S(48, 2), S(50, 2), S(96, 2), S(98, 2)]   # cfg trace: s_memtime stamps of a step
TRACE_BYTES = 32


class Gen(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.p = Prog()
        self.norm, self.sparse = cfg.get("norm", 0), cfg.get("sparse", False)
        self.given = self.norm in (2, 3)
        self.sited = not self.given          # guidance plane k is read at (y + dy_k, x + dx_k)
        self.scaled = not self.given or self.sparse   # the event multiplies the raw planes by the row's scale quad
        self.ab = set(cfg.get("ablate", ()))   # timing experiments only (results are wrong)
        self.mstubs, self.cstubs = [], []
        self.npieces = 10 if self.sparse else 9
        self.roles = self.make_roles()

    # ---------------------------------------------------------------------------------- roles
    def make_roles(self):
        """ring counter -> dict(cook=d | None, dma=d | None): the row is stream row 3 floor(tau / 2) + d.
        Even steps: the row entering 5 steps on is d = 7, the row entering next step d = 1; odd steps: d = 8, 9 and d = 2, 3.
        SIMDs hold the counters {c, c + 8, c + 16}: even steps have events in {0,8,16} and {2,10,18}, so the cook goes to {4,12,20}
        and the requests to {6,14,22}; odd steps have their event in {1,9,17}, the cooks in {3,11,19} and {5,13,21}, the requests
        in {7,15,23}.  No requests at c = 21 .. 23 (the wait for them would fall into the event steps, behind the retirement
        stores on the same counter), no cooking at c = 3 (the late planes of slot 2 are finished there with SQX)."""
        rc = self.cfg.get("role_counters", dict(cook={12: 1, 11: 2, 13: 3}, dma={14: 7, 7: 8, 15: 9}))
        roles = {c: dict(cook=None, dma=None) for c in range(LV)}
        for c, d in rc["cook"].items():
            assert (c & 1) == (0 if d == 1 else 1) and c > 3
            roles[c]["cook"] = d
        for c, d in rc["dma"].items():
            assert (c & 1) == (0 if d == 7 else 1) and 3 <= c < 21
            roles[c]["dma"] = d
        if "nocook" in self.ab:
            for r in roles.values():
                r["cook"] = None
        if "nodma" in self.ab or "nocook" in self.ab:
            for r in roles.values():
                r["dma"] = None
        return roles

    # ---------------------------------------------------------------------------------- small helpers
    def e(self, op, dst=(), src=(), **m):
        return self.p.emit(op, dst, src, **m)

    def fma(self, d, a, b, c, **m):
        keep = m.pop("keep", False)
        if "nostep" in self.ab and not keep:
            return
        self.e("v_pk_fma_f32", d, [a, b, c], **m)

    def mov(self, d, s):
        self.e("v_mov_b32", d, s)

    def swap(self, a, b):
        self.e("v_swap_b32", [a, b], [b, a])

    def shift(self, q, t):
        """q = (c0,c3,c2,c1); t[0:1] <- D = (c3 of lane-1, c0 of lane+1)"""
        if "nostep" in self.ab:
            return
        self.e("v_mov_b32", t[0], q[1], dpp="wave_shr:1")
        self.e("v_mov_b32", t[1], q[0], dpp="wave_shl:1")

    FLIP1 = dict(op_sel=[0, 1, 0], op_sel_hi=[1, 0, 1])   # src1 halves exchanged

    # Register layout of a row (4 columns c0..c3 per lane): the quad (c0, c3, c2, c1) -- a natural quad with elements 1 and 3 exchanged
    # --, pairs X = (c0,c3), Y = (c2,c1); D = (c3 of lane-1, c0 of lane+1) by two DPP moves.  Coefficient quads WT(j, q) (q = plane:
    # 0,1,2 below taps with dx = +1, 0, -1; 3,4 self taps dx = +1, -1; 5,6,7 above taps; 8: c') after the event's fix-up:
    #   dx = 0:   M = (km0, km3, km2, km1)
    #   dx = +1:  A = (kr0, kl3, kr2, kl1)        dx = -1:  B = (kl0, kr3, kl2, kr1)      (kr / kl: the dx = +1 / -1 plane)
    # so that every tap is one half of a v_pk_fma_f32 whose two depth operands share ONE aligned pair (possibly with its halves
    # exchanged, which op_sel does for free):
    #   dst X = (a0,a3):  M.lo * X            A.lo * (h1,h2) = Y flipped        B.lo * D = (h-1, h4)
    #   dst Y = (a2,a1):  M.hi * Y            A.hi * (h3,h0) = X flipped        B.hi * (h1,h2) = Y flipped
    def push3(self, qr, qm, ql, j, q, t, acc, init=None):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, qm).sub(0, 2), X, c0)
        self.fma(ay, WT(j, qm).sub(2, 2), Y, c1)
        self.fma(ax, WT(j, qr).sub(0, 2), Y, ax, **self.FLIP1)
        self.fma(ay, WT(j, qr).sub(2, 2), X, ay, **self.FLIP1)
        self.fma(ay, WT(j, ql).sub(2, 2), Y, ay, **self.FLIP1)
        self.fma(ax, WT(j, ql).sub(0, 2), D, ax)

    def push_below(self, j, q, t, acc, init=None):
        self.push3(0, 1, 2, j, q, t, acc, init)

    def push_above(self, j, q, t, acc, init=None):
        self.push3(5, 6, 7, j, q, t, acc, init)

    def push_self(self, j, q, t, acc, init=None):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, 3).sub(0, 2), Y, c0, **self.FLIP1)
        self.fma(ay, WT(j, 3).sub(2, 2), X, c1, **self.FLIP1)
        self.fma(ay, WT(j, 4).sub(2, 2), Y, ay, **self.FLIP1)
        self.fma(ax, WT(j, 4).sub(0, 2), D, ax)

    def ring_read(self, dst, ev, plane, **m):
        """dst quad <- plane (0..7 raw guidance, 8 level-0 value, 9 scale, 10 c') of the row entering slot ev, this lane's 4 columns"""
        if "noevlds" in self.ab:
            return
        self.e("ds_read_b128", dst, [V_EV], offset=ev * SLOT + plane * 1024, **m)

    def slot_of(self, c, d):
        """row 3 floor(tau / 2) + d at a step in which this wave's counter is c -> (S_SL3 register, byte offset inside the group)"""
        return S_SL3[(c // 2 + d // 3) % 3], (d % 3) * SLOT

    def fixup(self, j, planes, sq):
        """the planes of slot j just read (plane-pure natural quads) -> the mixed quads of push3, times the row's scale quad
        sq = (s0, s3, s2, s1) (the cooking task wrote it in that order)"""
        if "nofix" in self.ab:
            return
        for kr, kl in PAIRS:
            if kr in planes:
                assert kl in planes
                self.swap(WT(j, kr)[1], WT(j, kl)[3])
                self.swap(WT(j, kr)[3], WT(j, kl)[1])
        for km in (1, 6):
            if km in planes:
                self.swap(WT(j, km)[1], WT(j, km)[3])
        if self.scaled:
            for k in planes:
                if k < 8 and self.norm == 1:   # '8sum_abs': w = |G| / S (cspn.py:88-89); the packed multiply has no |x| modifier
                    for i in range(4):
                        self.e("v_mul_f32", WT(j, k)[i], [WT(j, k)[i].abs(), sq[i]])
                elif k < 8:
                    self.e("v_pk_mul_f32", WT(j, k).sub(0, 2), [WT(j, k).sub(0, 2), sq.sub(0, 2)])
                    self.e("v_pk_mul_f32", WT(j, k).sub(2, 2), [WT(j, k).sub(2, 2), sq.sub(2, 2)])

    # ---------------------------------------------------------------------------------- events
    def fetch_event(self, ev):
        """descriptor dwords 2:3 of the entering row (stream row S_QB + ev)"""
        self.e("s_add_i32", T[0], [S_QB, ev])
        self.e("s_lshl_b32", T[0], [T[0], 4])
        self.e("s_add_i32", T[0], [T[0], S_TABB])
        self.mov(V_D[0], T[0])
        self.e("ds_read_b64", V_D, [V_D[0]], offset=8, at=0.0)

    def take_event(self):
        self.e("v_readfirstlane_b32", S_NB, [V_D[0]])
        self.e("v_readfirstlane_b32", S_NF, [V_D[1]])

    def retire(self, j, vq):
        lab = self.p.newlabel("noret")
        self.e("s_bitcmp1_b32", (), [S_SF[j], F_OWNED])
        self.e("s_cbranch_scc0", (), [lab])
        stub, back = self.p.newlabel("mband"), self.p.newlabel("mbback")
        self.e("s_lshr_b32", T[10], [S_SF[j], 8])           # lo | hi << 12 of the retiring row's band
        self.e("s_cmp_lg_u32", (), [T[10], S_LOHIC])
        self.e("s_cbranch_scc1", (), [stub])
        self.p.label(back)
        self.mstubs.append((stub, back))
        self.swap(vq[1], vq[3])                             # (c0,c3,c2,c1) -> (c0,c1,c2,c3): the quad is dead after the store
        self.e("s_add_u32", T[8], [S_OUT[0], S_SB[j]])
        self.e("s_addc_u32", T[9], [S_OUT[1], 0])
        self.e("s_mov_b64", EXEC, [S_OMASK])
        if "nostore" not in self.ab:
            self.e("global_store_dwordx4", (), [V_L16, vq, S(T[8].i, 2)], cache=self.cfg.get("st_cache"))
        self.e("s_mov_b64", EXEC, [-1])
        self.p.label(lab)

    def inject(self, j, vq, hn, copy):
        for k in self.late_planes(j):
            self.ring_read(WT(j, k), j, k, at=0.0)
        if copy:
            for i in (1, 0, 2, 3):
                self.mov(vq[i], hn[i])
        self.e("s_mov_b32", S_SB[j], [S_NB])
        self.e("s_mov_b32", S_SF[j], [S_NF])
        self.e("s_bfe_i32", S_AM[j], [S_NF, (1 << 16) | F_ACTIVE])
        self.e("s_and_b32", T[2], [S_AM[0], S_AM[1]])
        self.e("s_and_b32", T[2], [T[2], S_AM[2]])
        self.e("s_cmp_lg_u32", (), [T[2], -1])            # vcc != 0 <=> some slot holds a separator / padding row: those
        self.e("s_cselect_b64", VCC, [1, 0])              # steps run the body that pins such slots to zero
        if j == NSLOT - 1:
            self.e("s_add_i32", S_QB, [S_QB, NW * NSLOT])

    # event planes: the coefficient planes of slot j that are dead when the event step starts (the row in the slot only needs its below
    # taps -- and, for slot 0, its above taps -- to finish its last level) are replaced at the top of the step, the others right after
    # the row completed; those are finished (fixup) at the top of the FOLLOWING step, in front of its wait for the boundary rows
    @staticmethod
    def early_planes(j):
        return (3, 4) if j == 0 else (3, 4, 5, 6, 7)

    @staticmethod
    def late_planes(j):
        return (0, 1, 2, 5, 6, 7) if j == 0 else (0, 1, 2)

    @staticmethod
    def sq_of(j):
        return SQY if j == 1 else SQX

    def act_check(self, j, vq):
        for k in (0, 1, 2, 3):
            self.e("v_and_b32", vq[k], [S_AM[j], vq[k]])

    def tail(self, c, skip_above1=False):
        """the part of step c nobody else waits for (slot 0's pushes after its value was published), emitted at the top of the
        following step, in front of the wait for the boundary rows"""
        p = c & 1
        v0 = ACC(p, 0)
        self.shift(v0, D_TAIL)
        self.push_self(0, v0, D_TAIL, ACC(p ^ 1, 0), init=WT(0, 8))
        if not skip_above1:
            self.push_above(1, v0, D_TAIL, ACC(p, 1), init=WT(1, 8))

    # ---------------------------------------------------------------------------------- the step
    def probe(self, k):
        if self.cfg.get("trace", False):
            r = TRACE_REGS[k]
            self.e("raw", (), ["s_memtime s[%d:%d]" % (r.i, r.i + 1)])

    def step(self, c, slow=False):
        p = c & 1
        N1 = [ACC(p, j) for j in range(NSLOT)]
        N2 = [ACC(p ^ 1, j) for j in range(NSLOT)]
        ev = c if c < NSLOT else None
        if "noevents" in self.ab:
            ev = None
        pev = c - 1 if 1 <= c <= NSLOT and "noevents" not in self.ab else None   # the previous step's event slot: its late planes are finished now
        role = self.roles[c]
        cook, dma = role["cook"], role["dma"]
        act_fast = "noact" not in self.ab
        if slow:
            self.p.label(".LSs%d_%%=" % c)
        else:
            self.p.label(".LS%d_%%=" % c)
            if act_fast:
                self.e("s_cbranch_vccnz", (), [".LSs%d_%%=" % c])
        prio = self.cfg.get("prio", 1) if ev is not None else (self.cfg.get("role_prio", 1) if (cook or dma) else 0)
        if prio:
            self.e("raw", (), ["s_setprio %d" % prio])
        self.probe(0)
        trace = self.cfg.get("trace", False)
        hn, ha = (HN, HA) if (ev is None or ev % 2 == 0) else (HA, HN)
        # ---- top: everything that travels through LDS is requested first
        if p == 0:
            self.e("s_add_u32", S_QTA, [S_QTA, 3 * DESC_BYTES])   # a new group of three rows: 3 floor(tau / 2)
        if "nolds" not in self.ab:
            self.e("ds_read_b128", BQ, [V_RB], offset=(p ^ 1) * BND_BUF, at=0.0)
            self.e("v_add_u32", TQ[0], [S_RT, V_L16])
            self.e("ds_read_b128", TQ, [TQ[0]], offset=(p ^ 1) * BND_BUF, at=0.0)
        n_after = 0
        if ev is not None:
            self.fetch_event(ev)
            n_after += 1
            self.ring_read(hn, ev, P_H0, at=0.0)
            n_after += 1
            if ev > 0:
                if self.scaled:
                    self.ring_read(self.sq_of(ev), ev, P_SC, at=0.0)
                    n_after += 1
                for k in self.early_planes(ev) + (8,):
                    self.ring_read(WT(ev, k), ev, P_CC if k == 8 else k, at=0.0)
                n_after += len(self.early_planes(ev)) + 1
            if "noevlds" in self.ab:
                n_after = 1
        if cook:
            self.mov(CK_F, S_QTA)
            self.e("ds_read_b32", CK_F, [CK_F], offset=cook * DESC_BYTES + 12, at=0.0)
        if dma:
            self.mov(HN[0], S_QTA)
            self.e("ds_read_b128", HN, [HN[0]], offset=dma * DESC_BYTES, at=0.0)
        if pev is not None:
            self.fixup(pev, self.late_planes(pev), self.sq_of(pev))
        self.tail((c - 1) % LV, skip_above1=(ev == 1))
        if ev == 0:  # slot 0's self taps and its c' were still needed by the deferred tail
            if self.scaled:
                self.ring_read(self.sq_of(0), 0, P_SC, at=0.0)
            for k in self.early_planes(0) + (8,):
                self.ring_read(WT(0, k), 0, P_CC if k == 8 else k, at=0.0)
            if "noevlds" not in self.ab:
                n_after += len(self.early_planes(0)) + 1 + (1 if self.scaled else 0)
        self.p.waitcnt(lgkm=min(n_after, 15) if (ev is not None and not trace) else 0)
        self.probe(1)
        if dma:
            self.dma_issue(c, dma)
        ck = self.cook_row(c, cook) if cook else None
        if ck:
            ck[0]()
        # received boundary rows
        self.shift(BQ, D_BQ)
        self.push_below(NSLOT - 1, BQ, D_BQ, N1[NSLOT - 1])
        self.shift(TQ, D_TQ)
        self.push_above(0, TQ, D_TQ, N1[0])
        if ck:
            ck[1]()
        for j in range(NSLOT - 1, -1, -1):
            vq = N1[j]
            tq = D_SLOT.get(j)
            if ck:   # the task's next phase shares a scheduling region with this slot's part of the chain
                self.p.waitcnt(lgkm=0)
                ck[4 - j]()
            if ev == j:
                self.p.waitcnt(lgkm=0)
                self.take_event()
                self.retire(j, vq)
                self.swap(hn[1], hn[3])                       # level 0 arrives as a natural quad
                self.fixup(j, self.early_planes(j), self.sq_of(j))
                self.inject(j, vq, hn, copy=(j == 0))
                if j > 0:
                    vq = hn
            elif "noact" not in self.ab and (slow or not act_fast):
                self.act_check(j, vq)
            if j == NSLOT - 1 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR, vq], offset=p * BND_BUF + 1024, at=0.0)
            if j == 0 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR, vq], offset=p * BND_BUF, at=0.0)
            if j == 0:
                break  # slot 0's own pushes: tail(), at the top of the next step
            self.shift(vq, tq)
            self.push_below(j - 1, vq, tq, N1[j - 1])
            if ev == j:
                self.push_self(j, vq, tq, N2[j], init=WT(j, 8))
                self.shift(ha, D_EV)
                self.push_above(j, ha, D_EV, N2[j])
            else:
                self.push_self(j, vq, tq, N2[j])
            if j < NSLOT - 1:
                self.push_above(j + 1, vq, tq, N1[j + 1], init=WT(j + 1, 8))
        self.probe(2)
        if prio:
            self.e("raw", (), ["s_setprio 0"])
        # the pieces this wave requested three steps ago are used in the next step: they must have landed before this step's barrier
        nd = lambda x: self.npieces if self.roles[x % LV]["dma"] else 0
        if nd(c - 3) and "nodmawait" not in self.ab:
            self.p.waitcnt(vm=min(nd(c - 2) + nd(c - 1) + nd(c), 63), lgkm=0)
        else:
            self.p.waitcnt(lgkm=0)
        self.probe(3)
        if "nobar" not in self.ab:
            self.e("s_barrier")
        self.trace_flush(c)
        self.e("s_sub_u32", S_TAU, [S_TAU, 1])
        self.e("s_cbranch_scc1", (), [".Lexit_%="])
        if slow or c == LV - 1:
            self.e("s_branch", (), [".LS%d_%%=" % ((c + 1) % LV)])

    def trace_flush(self, c):
        """cfg trace: the step's four stamps (low dwords) + the ring counter -> 32 bytes per wave and step (tools/r06/tsw4_trace.py)"""
        if not self.cfg.get("trace", False):
            return
        e = self.e
        e("raw", (), ["s_waitcnt lgkmcnt(0)"])
        for k, r in enumerate(TRACE_REGS):
            e("raw", (), ["v_writelane_b32 v9, s%d, %d" % (r.i, k)])
        e("raw", (), ["s_movk_i32 s%d, %d" % (T[0].i, c)])
        e("raw", (), ["v_writelane_b32 v9, s%d, 4" % T[0].i])
        e("raw", (), ["s_lshl_b32 s%d, s%d, 5" % (T[0].i, S_WV.i)])
        e("raw", (), ["v_lshrrev_b32_e32 v8, 2, v0"])
        e("raw", (), ["v_add_u32_e32 v8, s%d, v8" % T[0].i])
        e("raw", (), ["s_mov_b64 exec, 0x1f"])
        e("raw", (), ["global_store_dword v8, v9, s[26:27]"])
        e("raw", (), ["s_mov_b64 exec, -1"])
        e("raw", (), ["s_add_u32 s26, s26, %d" % (NW * TRACE_BYTES)])
        e("raw", (), ["s_addc_u32 s27, s27, 0"])

    # ---------------------------------------------------------------------------------- DMA of a raw row (one wave, all pieces)
    def dma_issue(self, c, d):
        e = self.e
        for i in range(3):
            e("v_readfirstlane_b32", S_DD[i], [HN[i]])
        grp, off = self.slot_of(c, d)
        n = self.npieces
        if "sameload" in self.ab:      # timing experiment: every request reads the same (cache-resident) rows
            e("s_mov_b32", S_DD[0], [0])
            e("s_mov_b32", S_DD[1], [0])
            e("s_mov_b32", S_DD[2], [0])
        for pc in range(n):
            P = S_P[pc % 4]
            if pc < 8:
                e("s_add_u32", P[0], [S_GDK[pc][0], S_DD[0]])
                e("s_addc_u32", P[1], [S_GDK[pc][1], S_DD[1]])
            else:
                base = S_BLUR if pc == 8 else S_SP
                e("s_add_u32", P[0], [base[0], S_DD[2]])
                e("s_addc_u32", P[1], [base[1], 0])
            at = self.cfg.get("dma_at", 0.05) + self.cfg.get("dma_span", 0.8) * pc / n
            e("s_add_u32", M0, [grp, off + pc * 1024], at=at)
            if "nodmaload" not in self.ab:
                e("global_load_lds_dwordx4", (), [V_L16, P, M0], cache=self.cfg.get("ld_cache"), at=at)

    # ---------------------------------------------------------------------------------- cooking: one wave per row, 4 pixels per lane
    def cook_row(self, c, d):
        """what normalise + fold (cspn.py:85-144, :76, :81) needs per pixel of the row that enters in the next step, from its plane-pure
        raw quads: scale = (1 - m) / sum_k |G_k| -> plane 9, c' = (1 - m)(1 - sigma) H0 + m H0 -> plane 10, both as (x0, x3, x2, x1).
        -> five phases the step emits between the parts of its chain (each but the first two behind an s_waitcnt lgkmcnt(0)): the raw
        quads cycle through four buffers, two of which are the boundary-row registers the head of the chain consumes first"""
        e, norm = self.e, self.norm
        grp, off = self.slot_of(c, d)
        Sq, Tq, buf = CK_S, CK_T, CK_BUF
        pair = lambda q: (q.sub(0, 2), q.sub(2, 2))
        need_s, need_t = not self.given, norm in (0, 3)
        sums = need_s or need_t
        nomath = "nocookmath" in self.ab

        def read(k):
            if sums and "nocookread" not in self.ab:
                e("ds_read_b128", buf[k % 4], [CK_A], offset=k * 1024, at=0.0)

        def consume(k):
            g = buf[k % 4]
            if nomath:
                return
            if need_s:
                for i in range(4):
                    if k == 0:
                        e("v_and_b32", Sq[i], [0x7fffffff, g[i]])
                    else:
                        e("v_add_f32", Sq[i], [Sq[i], g[i].abs()])
            if need_t:
                for h in (0, 1):
                    if k == 0:
                        e("v_pk_mul_f32", pair(Tq)[h], [pair(g)[h], 1.0])
                    else:
                        e("v_pk_add_f32", pair(Tq)[h], [pair(Tq)[h], pair(g)[h]])

        def ph0():
            e("v_readfirstlane_b32", S_CF, [CK_F])
            e("s_add_u32", T[3], [grp, off])
            e("v_add_u32", CK_A, [T[3], V_L16])
            if not nomath:
                stub, back = self.p.newlabel("ckfix"), self.p.newlabel("ckfixb")
                e("s_bitcmp1_b32", (), [S_CF, F_PLAIN])
                e("s_cbranch_scc0", (), [stub])
                self.p.label(back)
                self.cstubs.append((stub, back))
            read(0)
            read(1)

        def ph1():
            read(2)
            read(3)

        def ph2():
            for k in range(4):
                consume(k)
                read(k + 4)

        h0, sp, mm, om = buf[0], buf[1], buf[2], buf[3]

        def ph3():
            for k in range(4, 8):
                consume(k)
            e("ds_read_b128", h0, [CK_A], offset=P_H0 * 1024, at=0.0)
            if self.sparse:
                e("ds_read_b128", sp, [CK_A], offset=P_SC * 1024, at=0.0)

        def ph4():
            if nomath:
                return
            # RQ: the scale quad, CQ: the c' quad
            if norm == 0:
                RQ, CQ = Sq, Tq
                for i in range(4):
                    e("v_rcp_f32", RQ[i], [Sq[i]])
                for h in (0, 1):
                    e("v_pk_mul_f32", pair(CQ)[h], [pair(Tq)[h], pair(RQ)[h]])                        # sigma
                    self.fma(pair(CQ)[h], pair(CQ)[h], pair(h0)[h], pair(h0)[h], neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)   # (1 - sigma) H0
            elif norm == 1:
                RQ, CQ = Tq, Sq
                for i in range(4):
                    e("v_rcp_f32", RQ[i], [Sq[i]])
                for h in (0, 1):
                    e("v_pk_mul_f32", pair(CQ)[h], [pair(Sq)[h], pair(RQ)[h]])                        # sigma = S / S (NaN where S = 0)
                    self.fma(pair(CQ)[h], pair(CQ)[h], pair(h0)[h], pair(h0)[h], neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)
            elif norm == 3:
                RQ, CQ = Sq, Tq
                for h in (0, 1):
                    self.fma(pair(CQ)[h], pair(Tq)[h], pair(h0)[h], pair(h0)[h], neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)
            else:
                RQ, CQ = Sq, Tq
                for i in range(4):
                    self.mov(CQ[i], 0)
            if self.sparse:
                # m = sign(sparse) (NaN / 0 pass through), cspn.py:64,81
                for i in range(4):
                    self.mov(mm[i], sp[i])
                    e("v_cmp_gt_f32", S(T[4].i, 2), [sp[i], 0])
                    e("v_cndmask_b32", mm[i], [mm[i], 1.0, S(T[4].i, 2)])
                    e("v_cmp_lt_f32", S(T[6].i, 2), [sp[i], 0])
                    e("v_cndmask_b32", mm[i], [mm[i], -1.0, S(T[6].i, 2)])
                    e("v_sub_f32", om[i], [1.0, mm[i]])
                for h in (0, 1):
                    if not self.given:
                        e("v_pk_mul_f32", pair(RQ)[h], [pair(RQ)[h], pair(om)[h]])
                    else:
                        e("v_pk_mul_f32", pair(RQ)[h], [pair(om)[h], 1.0])
                    e("v_pk_mul_f32", pair(mm)[h], [pair(mm)[h], pair(h0)[h]])
                    self.fma(pair(CQ)[h], pair(om)[h], pair(CQ)[h], pair(mm)[h], keep=True)
            if "nocookwrite" not in self.ab:
                if self.scaled:
                    self.swap(RQ[1], RQ[3])
                    e("ds_write_b128", (), [CK_A, RQ], offset=P_SC * 1024, at=0.5)
                self.swap(CQ[1], CQ[3])
                e("ds_write_b128", (), [CK_A, CQ], offset=P_CC * 1024, at=0.6)

        return [ph0, ph1, ph2, ph3, ph4]

    def emit_cook_stub(self, stub, back):
        """rows that need patching, IN LDS, before the task (and, a step later, the event) reads them: an inactive (separator / padding)
        row becomes one unit coefficient and zero depths; the planes that look at a row above / below the image and the
        columns left / right of it are zeroed (the DMA fetched whatever lies there in the tensor: always inside it)"""
        e = self.e
        z = CK_BUF[0]
        self.p.label(stub)
        for i in range(4):
            self.mov(z[i], 0)
        l_act = self.p.newlabel("ckact")
        e("s_bitcmp1_b32", (), [S_CF, F_ACTIVE])
        e("s_cbranch_scc1", (), [l_act])
        # an inactive row: raw planes that cook to c' = 0 and a finite scale without a 0 / 0 (one unit coefficient, no depths); the slot is
        # pinned to zero anyway, but it pushes its level-0 value in the step it enters
        for pl in range(1, 10 if self.sparse else 9):
            e("ds_write_b128", (), [CK_A, z], offset=pl * 1024)
        for i in range(4):
            self.mov(z[i], 1.0)
        e("ds_write_b128", (), [CK_A, z], offset=0)
        e("s_branch", (), [back])
        self.p.label(l_act)
        if self.sited:
            for flag, planes in ((F_UP, (0, 1, 2)), (F_DN, (5, 6, 7))):
                lab = self.p.newlabel("edge")
                e("s_bitcmp1_b32", (), [S_CF, flag])
                e("s_cbranch_scc1", (), [lab])
                for k in planes:
                    e("ds_write_b128", (), [CK_A, z], offset=k * 1024)
                self.p.label(lab)
            # band column 0 = image column 0 of a first band: the dx = -1 planes' element 0 of lane 0; band column 255 = the image's last
            # column in a last band: the dx = +1 planes' element 3 of lane 63
            for flag, mask, planes, byte in ((F_FIRST, S_M0L, (2, 4, 7), 0), (F_LAST, S_M63, (0, 3, 5), 12)):
                lab = self.p.newlabel("noedgecol")
                e("s_bitcmp1_b32", (), [S_CF, flag])
                e("s_cbranch_scc0", (), [lab])
                e("s_mov_b64", EXEC, [mask])
                for k in planes:
                    e("ds_write_b32", (), [CK_A, z[0]], offset=k * 1024 + byte)
                e("s_mov_b64", EXEC, [-1])
                self.p.label(lab)
        e("s_branch", (), [back])

    # ---------------------------------------------------------------------------------- prologue
    def mod12(self, dst, src, add):
        e = self.e
        e("s_add_i32", dst, [src, add])
        e("s_cmp_ge_u32", (), [dst, NW])
        e("s_cselect_b32", T[11], [NW, 0])
        e("s_sub_u32", dst, [dst, T[11]])

    def prologue(self):
        e = self.e
        # LDS below the descriptor table (boundary rows, row slots) was zeroed by the C++ part of the kernel (cspn2d_tsw4.hip)
        e("v_lshlrev_b32", V_L16, [4, V_LANE])                 # (V_L16 is v0: the lane number is gone from here on)
        for r in range(ACC_BASE, WT_BASE + NSLOT * 36):
            self.mov(V(r), 0)
        for q in (HN, HA, SQX, SQY):
            for i in range(4):
                self.mov(q[i], 0)
        e("s_add_i32", S_TAU, [S_LAST, LEAD])
        for j in range(NSLOT):
            e("s_mov_b32", S_AM[j], [0])
            e("s_mov_b32", S_SB[j], [0])
            e("s_mov_b32", S_SF[j], [0])
        e("s_mov_b64", VCC, [1])
        e("s_mov_b32", S_LOHIC, [-1])    # the first retirement derives the owned-lane mask (no descriptor has lo | hi << 12 == -1)
        e("s_mov_b64", S_OMASK, [0])
        e("s_mov_b64", S_M0L, [1])
        e("s_mov_b32", S_M63[0], [0])
        e("s_mov_b32", S_M63[1], [0x80000000])
        # boundary rows: buffer 0 / 1 = the step's parity; wave w writes its top row at w * 2 KiB, its bottom row 1 KiB further
        e("s_lshl_b32", T[0], [S_WV, 11])
        e("s_add_i32", T[0], [T[0], S_LDSB])
        e("v_add_u32", V_WR, [T[0], V_L16])
        self.mod12(T[1], S_WV, 1)
        e("s_lshl_b32", T[1], [T[1], 11])
        e("s_add_i32", T[1], [T[1], S_LDSB])
        e("v_add_u32", V_RB, [T[1], V_L16])                    # the row below my rows: the next wave's top row
        self.mod12(T[1], S_WV, NW - 1)
        e("s_lshl_b32", T[1], [T[1], 11])
        e("s_add_i32", T[1], [T[1], 1024])
        e("s_add_i32", S_RT, [T[1], S_LDSB])                   # the row above: the previous wave's bottom row
        # slot groups: group (w + i) mod 3
        e("s_mul_i32", T[2], [S_WV, 11])
        e("s_lshr_b32", T[2], [T[2], 5])                       # w div 3 (w < 12)
        e("s_mul_i32", T[2], [T[2], 3])
        e("s_sub_i32", T[2], [S_WV, T[2]])                     # w mod 3
        for i in range(3):
            e("s_add_i32", T[3], [T[2], i])
            e("s_cmp_ge_u32", (), [T[3], 3])
            e("s_cselect_b32", T[4], [3, 0])
            e("s_sub_u32", T[3], [T[3], T[4]])
            e("s_mul_i32", T[3], [T[3], 3 * SLOT])
            e("s_add_i32", T[3], [T[3], LDS_ROWS])
            e("s_add_i32", S_SL3[i], [T[3], S_LDSB])
        e("v_add_u32", V_EV, [S_SL3[0], V_L16])
        # guidance plane bases (neighbour-sited: plane k is read one row / one pixel towards its neighbour)
        for k in range(8):
            e("s_mul_i32", T[3], [S_HW4, k])
            if self.sited:
                if DY[k] > 0:
                    e("s_add_i32", T[3], [T[3], S_W4])
                if DY[k] < 0:
                    e("s_sub_i32", T[3], [T[3], S_W4])
                if DX[k] != 0:
                    e("s_add_i32", T[3], [T[3], 4 * DX[k]])
            e("s_add_u32", S_GDK[k][0], [S_GD[0], T[3]])
            e("s_addc_u32", S_GDK[k][1], [S_GD[1], 0])
        # the workgroup's descriptor table is in LDS already (the C++ part of the kernel wrote it)
        e("s_add_i32", T[3], [S_LDSB, LDS_TAB])
        e("s_add_i32", S_TABB, [T[3], PADF * DESC_BYTES])
        # ring counters: the loop starts at step -LEAD; waves 8 .. 11 have their first events before step 0, on the inactive rows in
        # front of the stream
        e("s_mul_i32", S_QB, [S_WV, NSLOT])
        e("s_cmp_ge_u32", (), [S_WV, 8])
        e("s_cselect_b32", T[3], [NW * NSLOT, 0])
        e("s_sub_i32", S_QB, [S_QB, T[3]])
        assert LEAD % 2 == 0
        e("s_add_i32", S_QTA, [S_TABB, (3 * (-LEAD // 2) - 3) * DESC_BYTES])   # counted up by the first (even) step
        self.p.waitcnt(lgkm=0)
        e("s_barrier")
        for w in range(NW):
            c0 = (-LEAD - 2 * w) % LV
            e("s_cmp_eq_u32", (), [S_WV, w])
            e("s_cbranch_scc1", (), [".LS%d_%%=" % c0])

    def build(self):
        self.prologue()
        for c in range(LV):
            self.step(c)
        self.p.label(".Lexit_%=")
        self.e("s_waitcnt", vmcnt=0)                  # no LDS-DMA may be in flight when the workgroup's LDS is released
        self.e("s_branch", (), [".Lend_%="])
        if "noact" not in self.ab:
            for c in range(LV):
                self.step(c, slow=True)
        for st in self.cstubs:
            self.emit_cook_stub(*st)
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


def last_step(Q):
    """the step in which stream row Q - 1 retires"""
    return 2 * ((Q - 1) // NSLOT) + (Q - 1) % NSLOT + LV if Q > 0 else -1


def vgprs_used(prog):
    hi = -1
    for ins in prog.ins:
        for o in list(ins.dst) + list(ins.src):
            if isinstance(o, R) and o.kind == "v":
                hi = max(hi, o.i + o.n - 1)
    return hi + 1


def build(cfg, sched=True):
    from . import isa
    isa.SOFT_VALU_LATENCY = cfg.get("soft_lat", 1)
    g = Gen(cfg)
    p = g.build()
    if sched:
        isa.MIX_POLICY = cfg.get("mix", True)
        try:
            schedule(p)
        finally:
            isa.MIX_POLICY = False
    expand_pseudos(p)
    errs = check_hazards(p)
    if errs:
        raise RuntimeError("hazards:\n" + "\n".join(errs[:20]))
    assert vgprs_used(p) <= 168, vgprs_used(p)
    return p
