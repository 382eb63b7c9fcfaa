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
  * 36 working registers instead of 80.  No value of the feed ever sits in a VGPR longer than one cooking task:
      - raw rows (8 guidance planes neighbour-sited, blur, sparse) arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per piece = one
        plane of one band row), FIVE steps before the row enters, into a pool of 9 row slots of 10 KiB (slot = stream row mod 9;
        the slot is free exactly then: its previous row entered six steps before the new one);
      - a row is cooked IN PLACE in the step before it enters (normalise + fold, cspn.py:85-144, :76, :81), by four tasks of one
        pixel per lane (lane i of task t = band column 64 t + i): 10 ds_read_b32 / 10 ds_write_b32 over 64 consecutive dwords each
        (no bank conflicts), 14 temporaries that alias the registers only event steps use;
      - an event reads ten ds_read_b128 ([quad][lane][4 floats], the consumer's register order) straight into the slot's registers.
  * Every wave has a ROLE per step, static per ring counter (at any step the 12 waves hold the 12 counters of the step's parity):
    which cooking task it does and which DMA pieces it requests.  No wave requests more than 4 pieces or cooks more than one task per
    step; waves with an event (c = 0, 1, 2) do neither.
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

SLOT, NR = 10240, 9
BND_BUF = NW * 2048          # one boundary buffer: per wave its top row (slot 0's value) and its bottom row (slot 2's), 1 KiB each
LDS_BND, LDS_ROWS = 0, 2 * BND_BUF
LDS_TAB = LDS_ROWS + NR * SLOT
TAB_MAX_ROWS = (163840 - LDS_TAB) // DESC_BYTES
LDS_BYTES = LDS_TAB + TAB_MAX_ROWS * DESC_BYTES
assert LDS_BYTES == 163840 and 36 % NR == 0

DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]
# position (KiB inside a slot) of cooked quad q: the L quad of a row of taps sits 3 KiB behind its R quad, so that a cooking lane
# addresses "the quad my column's dx = +1 value belongs to" with ONE lane-dependent register for all three rows of taps
PQ = {0: 0, 3: 1, 5: 2, 2: 3, 4: 4, 7: 5, 1: 6, 6: 7, 8: 8, 9: 9}
RAW_BLUR, RAW_SP = 8, 9

# ---- VGPR map: 36 working registers + 24 accumulators + 108 coefficients = 168 ----
V_LANE = V(0)
V_L16, V_L4, V_CW, V_HI3K, V_WR, V_EV = V(0), V(1), V(2), V(3), V(4), V(5)
V_D = V(6, 2)                # descriptor fetches (address, then data)
BQ, TQ, HN, HA = V(8, 4), V(12, 4), V(16, 4), V(20, 4)
OUTQ = BQ                    # a retiring row is staged where the "row below" was (consumed at the head of the chain)
D_A, D_B, D_C, D_EV = V(24, 2), V(26, 2), V(28, 2), V(30, 2)
# shifted pairs D = (c3 of lane-1, c0 of lane+1): the deferred tail and slot 2 share D_A, the row below and slot 1 D_B, the row above D_C;
# the entering row's "row above" (event steps, which never cook) D_EV
D_TAIL, D_BQ, D_TQ, D_SLOT = D_A, D_B, D_C, {2: D_A, 1: D_B}
# cooking task (never in an event step): raw planes g0..g7 where HN / HA are, the rest in v30..v35
CK_G = [V(16 + k) for k in range(8)]
CK_RS, CK_T, CK_H0, CK_SP = V(30, 2), V(32, 2), V(34), V(35)    # (r, S) ; (T.lo, T.hi) -> sigma, c'
CK_TM, CK_TR, CK_TL = V(30), V(31), V(32)                         # write addresses, once r / S / T are dead
FETCH = [V(32, 2), V(28, 2)]                                      # descriptor fetches of DMA roles (dead between the top of a step and its use)
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
S_P = [S(2, 2), S(4, 2), S(6, 2), S(8, 2)]      # source base of a DMA piece (up to four per step and wave)
S_DD = [S(10), S(11), S(12), S(13)]             # descriptor dwords a DMA role fetched: (goff_lo, goff_hi) or boff, per row
S_CF = S(14)                                    # flags of the row a cooking task works on
S_QB, S_QTA, S_TABB = S(34), S(35), S(37)       # stream row of the wave's slot 0 (current cycle); LDS address of the descriptor of row 3 floor(tau / 2); of row 0
S_M0L, S_M63 = S(38, 2), S(40, 2)               # lane masks: lane 0 / lane 63
S_NB, S_NF = S(46), S(44)                       # entering row: descriptor dwords 2, 3
S_TAU = S(45)
S_SB = [S(52), S(53), S(54)]                    # per slot: the resident row's output byte offset
S_AM = [S(56), S(57), S(58)]                    # per slot: -1 real row, 0 separator / padding row
S_SF = [S(59), S(60), S(61)]                    # per slot: the resident row's descriptor dword 3
S_SL3 = [S(62), S(63), S(64)]                   # LDS address of the slot group (3 slots) of row group (this wave's + i) mod 3
S_RT, S_RB = S(65), S(66)                       # LDS address (buffer 0, lane 0) of the row above / below this wave's rows
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
        self.ab = set(cfg.get("ablate", ()))   # timing experiments only (results are wrong)
        self.mstubs, self.cstubs = [], []
        self.npieces = 10 if self.sparse else 9
        self.roles = self.make_roles()

    # ---------------------------------------------------------------------------------- roles
    def make_roles(self):
        """ring counter -> dict(cook=(d, t) | None, dma=[(d, piece)]): the row is stream row 3 floor(tau / 2) + d.
        Even steps: the row entering 5 steps on is d = 7, the row entering next step d = 1; odd steps: d = 8, 9 and d = 2, 3."""
        npc = self.npieces
        roles = {c: dict(cook=None, dma=[]) for c in range(LV)}
        for t in range(4):
            roles[4 + 2 * t]["cook"] = (1, t)
            roles[3 + 2 * t]["cook"] = (2, t)
            roles[11 + 2 * t]["cook"] = (3, t)
        if self.cfg.get("roles", 2) == 1:               # first version (profiles/r06_*: c = 2 waited for its own store)
            for pc in range(npc):                       # even steps: one piece per wave (c = 4 .. 22)
                roles[4 + 2 * pc]["dma"].append((7, pc))
            for pc in range(4):                         # odd steps: the cooking waves one piece each, three others the rest
                roles[3 + 2 * pc]["dma"].append((8, pc))
                roles[11 + 2 * pc]["dma"].append((9, pc))
            for pc in range(4, 8):
                roles[19]["dma"].append((8, pc))
                roles[21]["dma"].append((9, pc))
            for pc in range(8, npc):
                roles[23]["dma"].append((8, pc))
                roles[23]["dma"].append((9, pc))
        else:
            # no requests at c = 21 .. 23: the wait for them would fall into the event steps c = 0 .. 2, whose retirement stores share
            # the vmcnt counter.  Even steps: c = 4 .. 20 (nine waves) one piece each, the tenth (mask) with c = 20's.  Odd steps: the
            # cooking waves c = 3 .. 17 two pieces each, c = 19 the rest
            for pc in range(npc):
                roles[min(4 + 2 * pc, 20)]["dma"].append((7, pc))
            for i in range(4):
                roles[3 + 2 * i]["dma"] += [(8, 2 * i), (8, 2 * i + 1)]
                roles[11 + 2 * i]["dma"] += [(9, 2 * i), (9, 2 * i + 1)]
            for pc in range(8, npc):
                roles[19]["dma"] += [(8, pc), (9, pc)]
        if "nocook" in self.ab:
            for r in roles.values():
                r["cook"] = None
        if "nodma" in self.ab or "nocook" in self.ab:
            for r in roles.values():
                r["dma"] = []
        for c in (0, 1, 2):
            assert roles[c]["cook"] is None and not roles[c]["dma"]
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

    def shift(self, q, t):
        """q = (c0,c3,c1,c2); t[0:1] <- D = (c3 of lane-1, c0 of lane+1)"""
        if "nostep" in self.ab:
            return
        self.e("v_mov_b32", t[0], q[1], dpp="wave_shr:1")
        self.e("v_mov_b32", t[1], q[0], dpp="wave_shl:1")

    SWAP = dict(op_sel=[0, 1, 0], op_sel_hi=[1, 0, 1])   # src1 halves exchanged

    # coefficient quads WT(j, q) as in kernel.py: q = 0,1,2 below taps (R, M, L), 3,4 self taps (R, L), 5,6,7 above taps, 8 c'
    def push3(self, qr, qm, ql, j, q, t, acc, init=None):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, qm).sub(0, 2), X, c0)
        self.fma(ay, WT(j, qm).sub(2, 2), Y, c1)
        self.fma(ax, WT(j, qr).sub(0, 2), Y, ax)
        self.fma(ay, WT(j, qr).sub(2, 2), Y, ay, **self.SWAP)
        self.fma(ay, WT(j, ql).sub(2, 2), X, ay)
        self.fma(ax, WT(j, ql).sub(0, 2), D, ax)

    def push_below(self, j, q, t, acc, init=None):
        self.push3(0, 1, 2, j, q, t, acc, init)

    def push_above(self, j, q, t, acc, init=None):
        self.push3(5, 6, 7, j, q, t, acc, init)

    def push_self(self, j, q, t, acc, init=None):
        X, Y, D = q.sub(0, 2), q.sub(2, 2), t.sub(0, 2)
        ax, ay = acc.sub(0, 2), acc.sub(2, 2)
        c0, c1 = (init.sub(0, 2), init.sub(2, 2)) if init is not None else (ax, ay)
        self.fma(ax, WT(j, 3).sub(0, 2), Y, c0)
        self.fma(ay, WT(j, 3).sub(2, 2), Y, c1, **self.SWAP)
        self.fma(ay, WT(j, 4).sub(2, 2), X, ay)
        self.fma(ax, WT(j, 4).sub(0, 2), D, ax)

    def ring_read(self, dst, ev, q, **m):
        """dst quad <- cooked quad q (0..7 coefficients, 8 = c', 9 = level-0 value) of the row entering slot ev"""
        if "noevlds" in self.ab:
            return
        self.e("ds_read_b128", dst, [V_EV], offset=ev * SLOT + PQ[q] * 1024, **m)

    def slot_of(self, c, d):
        """row 3 floor(tau / 2) + d at a step in which this wave's counter is c -> (S_SL3 register, byte offset inside the group)"""
        return S_SL3[(c // 2 + d // 3) % 3], (d % 3) * SLOT

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
        self.mov(OUTQ[0], vq[0])   # registers hold (c0,c3,c1,c2)
        self.mov(OUTQ[1], vq[2])
        self.mov(OUTQ[2], vq[3])
        self.mov(OUTQ[3], vq[1])
        self.e("s_add_u32", T[8], [S_OUT[0], S_SB[j]])
        self.e("s_addc_u32", T[9], [S_OUT[1], 0])
        self.e("s_mov_b64", EXEC, [S_OMASK])
        if "nostore" not in self.ab:
            self.e("global_store_dwordx4", (), [V_L16, OUTQ, S(T[8].i, 2)], cache=self.cfg.get("st_cache"))
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

    @staticmethod
    def early_planes(j):
        return (3, 4, 8) if j == 0 else (3, 4, 5, 6, 7, 8)

    @staticmethod
    def late_planes(j):
        return (0, 1, 2, 5, 6, 7) if j == 0 else (0, 1, 2)

    def act_check(self, j, vq):
        for k in (0, 3, 2, 1):  # the DPP sources first (VALU -> DPP distance)
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
        role = self.roles[c]
        cook, dma = role["cook"], role["dma"]
        act_fast = "noact" not in self.ab
        if slow:
            self.p.label(".LSs%d_%%=" % c)
        else:
            self.p.label(".LS%d_%%=" % c)
            if act_fast:
                self.e("s_cbranch_vccnz", (), [".LSs%d_%%=" % c])
        prio = self.cfg.get("prio", 1) if ev is not None else (self.cfg.get("cook_prio", 2) if cook else 0)
        if prio:
            self.e("raw", (), ["s_setprio %d" % prio])
        self.probe(0)
        trace = self.cfg.get("trace", False)
        hn, ha = (HN, HA) if (ev is None or ev % 2 == 0) else (HA, HN)
        # ---- top: everything that travels through LDS is requested first
        if p == 0:
            self.e("s_add_u32", S_QTA, [S_QTA, 3 * DESC_BYTES])   # a new group of three rows: 3 floor(tau / 2)
        if "nolds" not in self.ab:
            self.e("v_add_u32", BQ[0], [S_RB, V_L16])
            self.e("ds_read_b128", BQ, [BQ[0]], offset=(p ^ 1) * BND_BUF, at=0.0)
            self.e("v_add_u32", TQ[0], [S_RT, V_L16])
            self.e("ds_read_b128", TQ, [TQ[0]], offset=(p ^ 1) * BND_BUF, at=0.0)
        n_after = 0
        if ev is not None:
            self.fetch_event(ev)
            n_after += 1
            self.ring_read(hn, ev, 9, at=0.0)
            n_after += 1
            if ev > 0:
                for k in self.early_planes(ev):
                    self.ring_read(WT(ev, k), ev, k, at=0.0)
                n_after += len(self.early_planes(ev))
            if "noevlds" in self.ab:
                n_after = 1
        fetched = self.dma_fetch(c, dma)
        if cook:
            self.cook_reads(c, *cook)
        self.tail((c - 1) % LV, skip_above1=(ev == 1))
        if ev == 0:  # slot 0's self taps were still needed by the deferred tail
            for k in self.early_planes(0):
                self.ring_read(WT(0, k), 0, k, at=0.0)
            if "noevlds" not in self.ab:
                n_after += len(self.early_planes(0))
        n_cook = 0
        if cook and self.cfg.get("cook_partial", True) and not trace and "nocookread" not in self.ab:
            n_cook = 9 + (1 if self.sparse else 0)     # the raw reads were requested last (behind the flags and the DMA roles' descriptors)
        self.p.waitcnt(lgkm=min(n_after, 15) if (ev is not None and not trace) else n_cook)
        self.probe(1)
        if dma:
            self.dma_issue(c, dma, fetched)
        if cook:
            self._cook_pending = (c,) + tuple(cook) if n_cook else None
            if not n_cook:
                self.cook_math(c, *cook)
                self.cook_writes(c, *cook)
        # received boundary rows
        self.shift(BQ, D_BQ)
        self.push_below(NSLOT - 1, BQ, D_BQ, N1[NSLOT - 1])
        self.shift(TQ, D_TQ)
        self.push_above(0, TQ, D_TQ, N1[0])
        if cook and getattr(self, "_cook_pending", None):
            self.p.waitcnt(lgkm=0)
            self.cook_math(*self._cook_pending)
            self.cook_writes(*self._cook_pending)
            self._cook_pending = None
        for j in range(NSLOT - 1, -1, -1):
            vq = N1[j]
            tq = D_SLOT.get(j)
            if ev == j:
                self.p.waitcnt(lgkm=0)
                self.take_event()
                self.retire(j, vq)
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
        # the pieces this wave requested three steps ago are cooked in the next step: they must have landed before this step's barrier
        nd = lambda x: len(self.roles[x % LV]["dma"])
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
        e("raw", (), ["v_add_u32_e32 v8, s%d, v1" % T[0].i])
        e("raw", (), ["s_mov_b64 exec, 0x1f"])
        e("raw", (), ["global_store_dword v8, v9, s[26:27]"])
        e("raw", (), ["s_mov_b64 exec, -1"])
        e("raw", (), ["s_add_u32 s26, s26, %d" % (NW * TRACE_BYTES)])
        e("raw", (), ["s_addc_u32 s27, s27, 0"])

    # ---------------------------------------------------------------------------------- DMA of raw rows
    def dma_fetch(self, c, dma):
        """descriptor dwords the wave's pieces need (goff for guidance planes, boff for blur / sparse), one fetch per row"""
        rows = []
        for d, pc in dma:
            key = (d, pc >= 8)
            if key not in rows:
                rows.append(key)
        assert len(rows) <= 2
        out = {}
        for i, (d, plain) in enumerate(rows):
            reg = FETCH[i]
            self.mov(reg[0], S_QTA)
            if plain:
                self.e("ds_read_b32", reg[0], [reg[0]], offset=d * DESC_BYTES + 8, at=0.0)
            else:
                self.e("ds_read_b64", reg, [reg[0]], offset=d * DESC_BYTES, at=0.0)
            out[(d, plain)] = (reg, S_DD[2 * i], S_DD[2 * i + 1])
        return out

    def dma_issue(self, c, dma, fetched):
        e = self.e
        for (d, plain), (reg, s0, s1) in fetched.items():
            e("v_readfirstlane_b32", s0, [reg[0]])
            if not plain:
                e("v_readfirstlane_b32", s1, [reg[1]])
        n = len(dma)
        for i, (d, pc) in enumerate(dma):
            reg, s0, s1 = fetched[(d, pc >= 8)]
            P = S_P[i]
            if pc < 8:
                e("s_add_u32", P[0], [S_GDK[pc][0], s0])
                e("s_addc_u32", P[1], [S_GDK[pc][1], s1])
            else:
                base = S_BLUR if pc == 8 else S_SP
                e("s_add_u32", P[0], [base[0], s0])
                e("s_addc_u32", P[1], [base[1], 0])
            grp, off = self.slot_of(c, d)
            at = self.cfg.get("dma_at", 0.1) + self.cfg.get("dma_span", 0.6) * i / max(1, n)
            e("s_add_u32", M0, [grp, off + pc * 1024], at=at)
            if "nodmaload" not in self.ab:
                e("global_load_lds_dwordx4", (), [V_L16, P, M0], cache=self.cfg.get("ld_cache"), at=at)

    # ---------------------------------------------------------------------------------- cooking: one pixel per lane
    def cook_reads(self, c, d, t):
        e = self.e
        grp, off = self.slot_of(c, d)
        self.mov(V_D[0], S_QTA)
        e("ds_read_b32", V_D[0], [V_D[0]], offset=d * DESC_BYTES + 12, at=0.0)
        last = CK_SP if self.sparse else CK_H0
        e("s_add_u32", T[3], [grp, off + 256 * t])
        e("v_add_u32", last, [T[3], V_L4])
        if "nocookread" in self.ab:
            return
        for k in range(8):
            e("ds_read_b32", CK_G[k], [last], offset=k * 1024, at=0.0)
        if self.sparse:
            e("ds_read_b32", CK_H0, [last], offset=RAW_BLUR * 1024, at=0.0)
            e("ds_read_b32", CK_SP, [last], offset=RAW_SP * 1024, at=0.0)
        else:
            e("ds_read_b32", CK_H0, [last], offset=RAW_BLUR * 1024, at=0.0)

    def cook_math(self, c, d, t):
        """normalise + fold one pixel per lane (cspn.py:85-144, :76, :81): CK_G[k] <- w'_k, CK_T[1] <- c', CK_H0 keeps the level-0 value"""
        e, g, norm = self.e, CK_G, self.norm
        e("v_readfirstlane_b32", S_CF, [V_D[0]])
        if "nocookmath" in self.ab:
            return
        stub, back = self.p.newlabel("ckfix"), self.p.newlabel("ckfixb")
        e("s_bitcmp1_b32", (), [S_CF, F_PLAIN])
        e("s_cbranch_scc0", (), [stub])
        self.p.label(back)
        self.cstubs.append((stub, back, t))
        r, s_ = CK_RS[0], CK_RS[1]
        tl, th = CK_T[0], CK_T[1]
        h0, sp = CK_H0, CK_SP
        bc = dict(op_sel=[0, 0], op_sel_hi=[1, 0])   # src1's low half for both results
        if norm == 1:
            for k in range(8):
                e("v_and_b32", g[k], [0x7fffffff, g[k]])
        if norm == 3:
            # prenorm: the planes ARE the w_k(p) of cspn.py:138; what is left of the fold is the centre term (cspn.py:76)
            e("v_pk_add_f32", CK_T, [V(g[0].i, 2), V(g[2].i, 2)])
            e("v_pk_add_f32", CK_T, [CK_T, V(g[4].i, 2)])
            e("v_pk_add_f32", CK_T, [CK_T, V(g[6].i, 2)])
            e("v_add_f32", th, [tl, th])                                  # sigma
            e("v_fma_f32", th, [-th, h0, h0])                              # (1 - sigma) * H0
        elif norm != 2:
            e("v_add_f32", s_, [g[0].abs(), g[1].abs()])
            for k in range(2, 8):
                e("v_add_f32", s_, [s_, g[k].abs()])
            if norm == 0:
                e("v_pk_add_f32", CK_T, [V(g[0].i, 2), V(g[2].i, 2)])
                e("v_pk_add_f32", CK_T, [CK_T, V(g[4].i, 2)])
                e("v_pk_add_f32", CK_T, [CK_T, V(g[6].i, 2)])
                e("v_add_f32", tl, [tl, th])
            else:
                self.mov(tl, s_)
            e("v_rcp_f32", r, [s_])
            e("v_mul_f32", th, [tl, r])                                    # sigma (8sum_abs: S / S = 1, NaN where S = 0)
            e("v_fma_f32", th, [-th, h0, h0])                              # (1 - sigma) * H0
        else:
            self.mov(th, 0)
        if self.sparse:
            # m = sign(sparse) (NaN / 0 pass through), cspn.py:64,81; s_ <- m, tl <- 1 - m
            mm, om = s_, tl
            self.mov(mm, sp)
            e("v_cmp_gt_f32", S(T[4].i, 2), [sp, 0])
            e("v_cndmask_b32", mm, [mm, 1.0, S(T[4].i, 2)])
            e("v_cmp_lt_f32", S(T[6].i, 2), [sp, 0])
            e("v_cndmask_b32", mm, [mm, -1.0, S(T[6].i, 2)])
            e("v_sub_f32", om, [1.0, mm])
            if not self.given:
                e("v_mul_f32", r, [r, om])
            else:
                self.mov(r, om)
            e("v_mul_f32", mm, [mm, h0])
            e("v_fma_f32", th, [om, th, mm])
        if not self.given or self.sparse:
            for k in range(0, 8, 2):
                e("v_pk_mul_f32", V(g[k].i, 2), [V(g[k].i, 2), CK_RS], **bc)

    def cook_writes(self, c, d, t):
        e, g = self.e, CK_G
        if "nocookwrite" in self.ab:
            return
        grp, off = self.slot_of(c, d)
        e("s_add_u32", T[4], [grp, off + 256 * t])
        e("v_add_u32", CK_TM, [T[4], V_CW])
        e("v_add_u32", CK_TR, [CK_TM, V_HI3K])                 # R quads for columns 0, 1 -- L quads (3 KiB further) for columns 2, 3
        e("v_sub_u32", CK_TL, [CK_TR, V_HI3K])
        e("v_sub_u32", CK_TL, [CK_TL, V_HI3K])
        e("v_add_u32", CK_TL, [3072, CK_TL])                   # and the other way round
        items = [(CK_TR, g[0], 0), (CK_TR, g[3], 1), (CK_TR, g[5], 2), (CK_TL, g[2], 0), (CK_TL, g[4], 1), (CK_TL, g[7], 2),
                 (CK_TM, g[1], 6), (CK_TM, g[6], 7), (CK_TM, CK_T[1], 8), (CK_TM, CK_H0, 9)]
        a0, sp = self.cfg.get("cw_at", 0.5), self.cfg.get("cw_span", 0.4)
        for i, (addr, reg, kib) in enumerate(items):
            e("ds_write_b32", (), [addr, reg], offset=kib * 1024, at=a0 + sp * i / len(items))

    def emit_cook_stub(self, stub, back, t):
        """rows that need patching: an inactive (separator / padding) row cooks to zeros; the row above / below the image and the columns
        left / right of it count as zero (the raw reads fetched whatever lies there in the tensor: always inside it)"""
        e, g = self.e, CK_G
        self.p.label(stub)
        l_act = self.p.newlabel("ckact")
        e("s_bitcmp1_b32", (), [S_CF, F_ACTIVE])
        e("s_cbranch_scc1", (), [l_act])
        # inputs that cook to all-zero coefficients, c' = 0, level 0 = 0 without a 0 / 0 (the slot is pinned to zero anyway, but it
        # pushes its level-0 value in the step it enters)
        for k in range(8):
            self.mov(g[k], 0)
        self.mov(CK_H0, 0)
        if self.sparse:
            self.mov(CK_SP, 0)
        if not self.given:
            self.mov(g[0], 1.0)
        e("s_branch", (), [back])
        self.p.label(l_act)
        if self.sited:
            for flag, planes in ((F_UP, (0, 1, 2)), (F_DN, (5, 6, 7))):
                lab = self.p.newlabel("edge")
                e("s_bitcmp1_b32", (), [S_CF, flag])
                e("s_cbranch_scc1", (), [lab])
                for k in planes:
                    self.mov(g[k], 0)
                self.p.label(lab)
            if t == 0:      # band column 0 = image column 0 of a first band: the dx = -1 planes
                lab = self.p.newlabel("nofirst")
                e("s_bitcmp1_b32", (), [S_CF, F_FIRST])
                e("s_cbranch_scc0", (), [lab])
                for k in (2, 4, 7):
                    e("v_cndmask_b32", g[k], [g[k], 0, S_M0L])
                self.p.label(lab)
            if t == 3:      # band column 255 = the image's last column in a last band: the dx = +1 planes
                lab = self.p.newlabel("nolast")
                e("s_bitcmp1_b32", (), [S_CF, F_LAST])
                e("s_cbranch_scc0", (), [lab])
                for k in (0, 3, 5):
                    e("v_cndmask_b32", g[k], [g[k], 0, S_M63])
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
        e("v_lshlrev_b32", V_L4, [2, V_LANE])
        # position of column c in the quad (c0, c3, c1, c2): 0, 2, 3, 1 = 2 * ((c & 1) ^ (c >> 1)) + (c >> 1)
        e("v_and_b32", BQ[0], [1, V_LANE])
        e("v_lshrrev_b32", BQ[1], [1, V_LANE])
        e("v_and_b32", BQ[1], [1, BQ[1]])
        e("v_xor_b32", BQ[0], [BQ[0], BQ[1]])
        e("v_lshlrev_b32", BQ[0], [1, BQ[0]])
        e("v_add_u32", BQ[0], [BQ[0], BQ[1]])
        e("v_lshlrev_b32", BQ[0], [2, BQ[0]])                  # 4 * pos
        e("v_lshrrev_b32", V_CW, [2, V_LANE])
        e("v_lshlrev_b32", V_CW, [4, V_CW])                    # 16 * (lane / 4): the consumer lane's quad
        e("v_add_u32", V_CW, [V_CW, BQ[0]])
        e("v_and_b32", V_HI3K, [2, V_LANE])
        e("v_mul_u32_u24", V_HI3K, [1536, V_HI3K])             # 3072 for columns 2, 3
        e("v_lshlrev_b32", V_L16, [4, V_LANE])                 # (V_L16 is v0: the lane number is gone from here on)
        for r in range(ACC_BASE, WT_BASE + NSLOT * 36):
            self.mov(V(r), 0)
        for i in range(4):
            self.mov(HN[i], 0)
            self.mov(HA[i], 0)
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
        e("s_add_i32", S_RB, [T[1], S_LDSB])                   # the row below my rows: the next wave's top row
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
