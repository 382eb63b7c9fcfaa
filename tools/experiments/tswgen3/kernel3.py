"""tools/tswgen/kernel3.py -- generator of the round-3 main loop of the fused CSPN kernel: the "time-skewed wave ring" of
kernel.py with a different way of getting the guidance to the ring.

What stays (see kernel.py / DESIGN.md 3.1b): 8 waves x 4 resident rows x 4 columns per lane, 24 unrolled phases of a wave's
ring counter, the register pairing X = (c0,c3), Y = (c1,c2), one s_barrier per step, boundary rows through LDS.

What is new (cspn.py:85-144 normalisation, :76 centre term, :81 mask -- "cooking" -- and how its inputs arrive):
  * LDS-DMA: the raw rows (8 guidance planes neighbour-sited, blur, sparse / previous-pass depth) are fetched with
    global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR destination) TWO cooking groups = six steps ahead of their
    use, so HBM requests are in flight all the time and no compute wave ever waits on vmcnt for data it needs now
    (profiles/r03_ubench_dma.txt: 6.0 TB/s beside a VALU-bound barrier-stepped loop from 3 steps of lead on).
  * One pool of 12 row slots of 10 KiB in LDS (slot = stream row mod 12).  A slot holds a row first raw, as the DMA wrote it
    -- pair records [planes (kr, kl)][half][kr: 32 lanes x 16 B | kl: 32 lanes x 16 B] so that one ds_read2_b32 fetches
    (kr[p0], kl[p3]) --, then cooked IN PLACE: ten quads per consumer lane in the consumer's register order
    ([quad q][lane][4 floats]), so that an event is ten ds_read_b128 instead of 22 ds_read2_b32.
  * A cooking task = one wave, half a row; lane i holds pixels i and i + 64 of the half, so that every raw read and every
    cooked write of a wave covers 64 consecutive dwords (no LDS bank conflicts; the first version read 4 bytes at a 16-byte lane
    stride -- 4-way conflicts -- and lost 0.03 ms to that, profiles/r03_ablations_a.txt).
  * Cooking is spread over the three steps of a group: step = 0 (mod 3) issues the DMA of the group two ahead, step = 1
    reads the raw values of the next group from LDS and does the arithmetic, step = 2 writes the cooked quads back.
  * Row descriptors are 4 bytes (flags | image << ybits | y) instead of 16: the table shrinks from 44 KB to 8 KB; the
    DMA-issuing wave leaves "output byte offset | owned | active" in a per-slot header for the event that injects the row.
Variants: norm x (sparse | hin | none).  History / adjoint / sparse+hin variants stay on kernel.py's loop.
"""
from tools.tswgen.isa import Prog, V, S, R, EXEC, VCC, M0, I, schedule, check_hazards, expand_pseudos

R_EXEC_LO, R_EXEC_HI = R("exec", 0, 1), R("exec", 1, 1)

NW, NSLOT, LV = 8, 4, 24
PADF, PADB = 36, 64          # inactive descriptor rows before / after a workgroup's stream
DESC_BYTES = 4
# descriptor dword: flags | (image << ybits | y) << 4
F_ACTIVE, F_UP, F_DN, F_OWNED = 0, 1, 2, 3
# slot header dword (written by the DMA-issuing wave, read by the event): byte offset in a 1-channel tensor (multiple of 16)
H_OWNED, H_ACTIVE = 0, 1
# band flags in S_GEOM
G_FIRST, G_LAST = 8, 9

SLOT, NSLOTS = 10240, 12
LDS_BND, LDS_ROWS = 0, 32768
LDS_HDR = LDS_ROWS + NSLOTS * SLOT          # 12 slot headers (+ 4 spare dwords)
LDS_TAB = LDS_HDR + 64
TAB_MAX_ROWS = (163840 - LDS_TAB) // DESC_BYTES
LDS_BYTES = LDS_TAB + TAB_MAX_ROWS * DESC_BYTES
assert LDS_BYTES == 163840

DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]
# pair records of a raw row: (plane in the first 512 bytes, plane in the second) -- the dx = +1 / dx = -1 planes of a row of taps,
# and the two dx = 0 planes together
PAIRS = [(0, 2), (3, 4), (5, 7), (1, 6)]
REC_BLUR, REC_AUX = 8, 9     # plain 1 KiB rows: blur; sparse or previous-pass depth

# ---- VGPR map ----
V_LANE, V_L16 = V(0), V(2)
V_WR, V_RT, V_RB = [V(3), V(4)], [V(5), V(6)], [V(7), V(8)]
# cooking task = half a row (128 pixels), lane i holds pixels i and i + 64 of the half: consumer lanes i / 4 and i / 4 + 16, column
# i % 4.  Lane bases inside a slot: raw reads of the plane that belongs into the R / L quad for the lane's column (the dx = +1
# plane for columns 0,1 in R, the dx = -1 plane for columns 2,3, and vice versa for L) and of a plain row: 4 * i (+ 512 for the
# second plane of a pair record); cooked write: 16 * (i / 4) + 4 * position of the column in the (c0,c3,c1,c2) quad
V_CKR, V_CKL, V_CKW = V(9), V(10), V(1)
V_RINGE = V(11)             # event: address of the lane's quads in the slot group of the wave's current burst
V_OFFX = [V(12 + i) for i in range(4)]   # DMA: per-lane byte offsets of this wave's four guidance pieces
V_ADR = [V(16), V(17)]      # cooking: address temporaries
V_HDR = V(17)               # DMA: slot-header value (lane-uniform; no cooking read in a DMA step)
# cooking arithmetic: temporaries of its own, so that the scheduler may put it anywhere in the step
CK_SX, CK_SY, CK_TT, CK_SCALE, CK_T2, CK_M = V(20), V(253), V(18, 2), V(22, 2), V(254, 2), V(74, 2)
V_TMP = V(21)
V_DE = V(21)                # event: slot header fetch
V_DC = V(252)               # cooking / DMA: descriptor fetch
TQ, BQ, TA, TB, HN, HA, OUTQ = V(24, 4), V(28, 4), V(32, 4), V(36, 4), V(40, 4), V(44, 4), V(48, 4)
D_BQ, D_TQ, D_SLOT, D_TAIL = OUTQ.sub(0, 2), TB.sub(2, 2), {3: TA.sub(0, 2), 2: TA.sub(2, 2), 1: TB.sub(0, 2)}, OUTQ.sub(2, 2)
CK = V(24, 28)  # cooking temporaries alias the step temporaries
PEND_G = [V(52 + 2 * k, 2) for k in range(8)]     # quad halves q = 0..7 of the task in flight (raw, then cooked)
PEND_BLUR, PEND_HIN, PEND_SP = V(68, 2), V(70, 2), V(72, 2)
ACC_BASE, WT_BASE = 76, 108


def ACC(p, j):
    return V(ACC_BASE + (p * 4 + j) * 4, 4)


def WT(j, k):
    return V(WT_BASE + (j * 9 + k) * 4, 4)


# ---- SGPR map ----
S_GD, S_BLUR, S_HIN, S_SP, S_OUT, S_PLAN = S(16, 2), S(18, 2), S(20, 2), S(22, 2), S(24, 2), S(26, 2)
S_W4, S_HW4, S_LAST, S_WV = S(28), S(29), S(30), S(31)
S_LDSB = S(15)   # LDS base address of the kernel's __shared__ block
S_GEOM = S(14)   # input: ybits | first band << 8 | last band << 9
S_LOHI = S(13)   # input: owned columns of this workgroup's band, band relative: lo | hi << 16
S_P04 = S(12)    # input: byte offset of the band's first column inside an image row
S_OMASK = S(42, 2)  # lanes whose 4 columns lie inside [lo, hi)
S_TAU, S_QB, S_TG, S_CFLAGS = S(45), S(34), S(35), S(36)  # (s32 is reserved by the compiler: stack pointer)
# S_TG: LDS address of the descriptor of stream row 4 g, g = step number div 3 (the cooking group entering now)
S_EL, S_ER = S(38, 2), S(40, 2)
S_TABB = S(37)                  # LDS address of descriptor row 0 (table base + PADF rows)
S_EGRP = S(46)                  # LDS address of the slot group (4 slots) of the wave's current event burst
S_G3S, S_G3H = S(47), S(61)     # (g mod 3) * 4 slots as a byte offset into the slot pool / into the slot headers
S_CSLOT = S(50)                 # LDS address of the slot of the wave's cooking task
S_JJS, S_CK4 = S(51), S(44)     # per wave: (3 + (wv >> 1)) * SLOT and 12 + 4 * (wv >> 1): its task's slot / descriptor relative to g
S_H1K, S_H512 = S(66), S(67)    # per wave: 1024 * (wv & 1), 512 * (wv & 1): its half of a pair record / of a plain row or a quad
S_SLOTB = [S(52), S(53), S(54), S(55)]   # per ring slot of the wave: the resident row's header (output offset | owned | active)
S_AM = [S(0), S(1), S(4), S(5)]          # per ring slot: -1 if it holds a real row, 0 for a separator / padding row
S_YBFE, S_BSH, S_G8 = S(56), S(57), S(58)
S_ROWE = S(59)                  # LDS address behind the last row slot
S_HDRE = S(63)                  # LDS address behind the last slot header
S_CD = S(60)                    # descriptor of the row whose DMA is being issued
S_EHDR = S(62)                  # LDS address of the headers of the slot group S_EGRP
S_DMB = S(64)                   # LDS address of the slot of the row whose DMA is being issued
S_POOL = S(65)                  # LDS address of the first row slot
S_HDRB = S(10)                  # LDS address of the first slot header
T = [S(68 + i) for i in range(12)]  # scalar temporaries s68..s79
S_ELC, S_ERC = S(80, 2), S(82, 2)   # per-wave constant lane masks (half 0: lane 0 / half 1: lane 63)
GB, BX, AX = S(2, 2), S(6, 2), S(48, 2)   # row bases of the DMA being issued (guidance; blur; sparse / previous-pass depth)
# LDS-DMA schedule: a wave issues a whole raw row (8 pair-record pieces + blur + aux) at four of its 24 counters, so that at
# most two waves (on different SIMDs) feed the CU's one vector-memory address path in any step: counter -> (row jj of a
# cooking group, groups ahead of the one entering now).  Rows of group g (they enter from step 3g on, are read at 3g - 2) are
# requested at steps 3g - 8 (jj 0), 3g - 7 (jj 1), 3g - 6 (jj 2, 3): their slots were vacated by the events of steps 3g - 9 ..
# 3g - 7.  The requesting wave waits for its row at the top of step 3g - 3 (DMA_WAIT counters), the barrier of that step
# publishes it.
# (Counter 15 shares its SIMD with the event wave at counter 3 -- waves w and w + 4 share one, their counters differ by 12;
# requesting at 21 instead, cfg dma_issue / dma_wait, keeps every SIMD to one heavy wave-step: measured, no difference, the
# scalar instructions of a request co-issue with the partner's FMAs; profiles/r03_ablations_raw.txt r3n.)
DMA_ISSUE = {6: (2, 2), 15: (3, 2), 10: (0, 3), 20: (1, 3)}
DMA_WAIT = (9, 18, 15, 0)


class Gen(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.p = Prog()
        self.norm, self.sparse, self.hin = cfg.get("norm", 0), cfg.get("sparse", False), cfg.get("hin", False)
        assert not (self.sparse and self.hin), "sparse + continuation pass: kernel.py's loop"
        self.sited = self.norm != 2   # guidance plane k is read at (y + dy_k, x + dx_k)
        self.aux = self.sparse or self.hin
        assert cfg.get("n_iter", 24) == 24
        self.stubs = []
        self.cstubs = []
        self.ab = set(cfg.get("ablate", ()))  # timing experiments only (results are wrong)
        self.dma_issue = cfg.get("dma_issue", DMA_ISSUE)   # (A/B builds of other request schedules)
        self.dma_wait = cfg.get("dma_wait", DMA_WAIT)

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

    # Register layout of a row (4 columns c0..c3 per lane): the quad (c0, c3, c1, c2), i.e. the pairs X = (c0,c3) and
    # Y = (c1,c2).  With D = (c3 of lane-1, c0 of lane+1) -- two DPP moves -- every tap of every column is one half of a
    # v_pk_fma_f32 whose two h operands sit in ONE aligned register pair (kernel.py has the table).  Coefficient quads
    # WT(j, q), q = 0..8: for a row of taps with planes (kr, km, kl) = (dx = +1, 0, -1):
    #   R = (kr0, kl3, kr1, kl2),  M = (km0, km3, km1, km2),  L = (kl0, kr3, kl1, kr2)
    #   q = 0,1,2: below taps (planes 0,1,2)   q = 3,4: self taps R, L (planes 3,4)   q = 5,6,7: above taps   q = 8: c'
    def shift(self, q, t):
        """q = (c0,c3,c1,c2); t[0:1] <- D = (c3 of lane-1, c0 of lane+1)"""
        if "nostep" in self.ab:
            return
        self.e("v_mov_b32", t[0], q[1], dpp="wave_shr:1")
        self.e("v_mov_b32", t[1], q[0], dpp="wave_shl:1")

    SWAP = dict(op_sel=[0, 1, 0], op_sel_hi=[1, 0, 1])   # src1 halves exchanged

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
        """dst quad <- cooked quad q (0..7 coefficients, 8 = c', 9 = level-0 value) of the row in slot ev of the current burst"""
        if "noevlds" in self.ab:
            return
        self.e("ds_read_b128", dst, [V_RINGE], offset=ev * SLOT + q * 1024, **m)

    def wrap_slot(self, reg, add):
        """reg <- reg + add, wrapped into the pool of row slots"""
        self.e("s_add_u32", reg, [reg, add])
        self.e("s_cmp_ge_u32", (), [reg, S_ROWE])
        self.e("s_cselect_b32", T[11], [NSLOTS * SLOT, 0])
        self.e("s_sub_u32", reg, [reg, T[11]])

    def wrap(self, reg, add, mod):
        """reg <- (reg + add) mod `mod` for 0 <= reg + add < 2 mod (plain byte offsets)"""
        if add:
            self.e("s_add_u32", reg, [reg, add])
        self.e("s_cmp_ge_u32", (), [reg, mod])
        self.e("s_cselect_b32", T[11], [mod, 0])
        self.e("s_sub_u32", reg, [reg, T[11]])

    def wrap_hdr(self, reg, add):
        """reg <- reg + add, wrapped into the 12 slot headers"""
        self.e("s_add_u32", reg, [reg, add])
        self.e("s_cmp_ge_u32", (), [reg, S_HDRE])
        self.e("s_cselect_b32", T[11], [4 * NSLOTS, 0])
        self.e("s_sub_u32", reg, [reg, T[11]])

    # ---------------------------------------------------------------------------------- events
    def fetch_event(self, ev):
        """LDS read of the header of the slot the entering row was cooked in (issued with the boundary-row reads)"""
        self.mov(V_DE, S_EHDR)
        self.e("ds_read_b32", V_DE, [V_DE], offset=4 * ev, at=0.0)

    def take_event(self, ev):
        self.e("v_readfirstlane_b32", T[1], [V_DE])

    def retire(self, j, vq):
        lab = self.p.newlabel("noret")
        self.e("s_and_b32", T[8], [S_SLOTB[j], (1 << H_OWNED) | (1 << H_ACTIVE)])   # (a header is only ever both or not owned)
        self.e("s_cmp_eq_u32", (), [T[8], (1 << H_OWNED) | (1 << H_ACTIVE)])
        self.e("s_cbranch_scc0", (), [lab])
        self.mov(OUTQ[0], vq[0])   # registers hold (c0,c3,c1,c2)
        self.mov(OUTQ[1], vq[2])
        self.mov(OUTQ[2], vq[3])
        self.mov(OUTQ[3], vq[1])
        self.e("s_and_b32", T[8], [S_SLOTB[j], 0xfffffff0])
        self.e("s_add_u32", T[8], [S_OUT[0], T[8]])
        self.e("s_addc_u32", T[9], [S_OUT[1], 0])
        self.e("s_mov_b64", EXEC, [S_OMASK])   # the owned columns are the same for every row of this workgroup's band
        if "nostore" not in self.ab:
            self.e("global_store_dwordx4", (), [V_L16, OUTQ, S(T[8].i, 2)], cache=self.cfg.get("st_cache"))
        self.e("s_mov_b64", EXEC, [-1])
        self.p.label(lab)

    def inject(self, j, vq, hn=HN, copy=True):
        for k in self.late_planes(j):
            self.ring_read(WT(j, k), j, k, at=0.0)
        if copy:   # slots 1..3 use the freshly read quad itself as the row's level-0 value (nothing reads vq before it is
            for i in (1, 0, 2, 3):   # re-initialised); slot 0's deferred tail needs it in the accumulator register
                self.mov(vq[i], hn[i])
        self.e("s_mov_b32", S_SLOTB[j], [T[1]])            # the entering row's header (take_event)
        self.e("s_bfe_i32", S_AM[j], [T[1], (1 << 16) | H_ACTIVE])
        self.e("s_and_b32", T[2], [S_AM[0], S_AM[1]])
        self.e("s_and_b32", T[9], [S_AM[2], S_AM[3]])
        self.e("s_and_b32", T[2], [T[2], T[9]])
        self.e("s_cmp_lg_u32", (), [T[2], -1])            # vcc != 0 <=> some slot holds a separator / padding row: those
        self.e("s_cselect_b64", VCC, [1, 0])              # steps run the body that pins such slots to zero
        if j == 3:
            self.wrap_slot(S_EGRP, 8 * SLOT)               # the next burst's rows are 32 further down the stream: 32 mod 12 = 8 slots
            self.wrap_hdr(S_EHDR, 32)

    # event planes: the coefficient quads of slot j that are dead when the event step starts (the row in the slot only
    # needs its below taps -- and, for slot 0, its above taps -- to finish its last level) can be replaced at the top of
    # the step, together with the boundary-row reads; the others right after the row completed.  Nothing waits mid-step.
    @staticmethod
    def early_planes(j):
        return (3, 4, 8) if j == 0 else (3, 4, 5, 6, 7, 8)

    @staticmethod
    def late_planes(j):
        return (0, 1, 2, 5, 6, 7) if j == 0 else (0, 1, 2)

    def act_check(self, j, vq):
        """a slot holding a separator / padding row is pinned to zero (0 x NaN from a neighbour must not leak into it): the
        completed value is ANDed with the slot's 0 / -1 mask.  Straight-line: the first version branched to a stub per slot,
        and its ~8 taken branches per step in the ~26 % of steps with such a row resident (ring fill / drain, image
        boundaries) cost 0.04 ms of a 0.29 ms forward (profiles/r03_ablations.md)"""
        for k in (0, 3, 2, 1):  # the DPP sources first (VALU -> DPP distance)
            self.e("v_and_b32", vq[k], [S_AM[j], vq[k]])

    # ---- cfg trace (timing instrumentation, tools/tsw_trace.py): six s_memtime stamps per step
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
                e("raw", (), ["v_writelane_b32 v74, s%d, 1" % self.TRACE_REGS[2].i])   # no DMA wait in this step: stamp 1 = stamp 2
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

    def step(self, c, slow=False):
        """one step of a wave whose ring counter is c.  Two bodies per counter: the fast one assumes that all four slots hold
        real rows (vcc == 0) and carries no per-slot checks; `slow` (out of line, entered by one branch at the top of the fast
        body) pins slots that hold a separator / padding row to zero"""
        p = c & 1
        N1 = [ACC(p, j) for j in range(4)]
        N2 = [ACC(p ^ 1, j) for j in range(4)]
        ev = c if c < 4 else None
        if "noevents" in self.ab:
            ev = None
        # Cooking, spread over the three steps of a group (the wave's counter = the step number mod 3): the group of four
        # rows that enters from step 3g on has its raw data requested at step 3g - 6, read at 3g - 2, normalised and its
        # cooked quads written at 3g - 1.
        nocook = "nocook" in self.ab
        ph = c % 3
        cookr, cookw = (ph == 1 and not nocook), (ph == 2 and not nocook)
        dma = None if nocook else self.dma_issue.get(c)                     # (jj, groups ahead) of the row this wave requests now
        dma_next = None if nocook else self.dma_issue.get((c + 1) % LV)    # ... at the top of the next step: fetch its descriptor
        dma_wait = c in self.dma_wait and not nocook
        act_fast = self.cfg.get("act_fast", True) and "noact" not in self.ab
        cook_top = self.cfg.get("cook_top", True)
        tau3 = self.cfg.get("tau3", True)
        nxt = ".LS%d_%%=" % ((c + 1) % LV)
        if slow:
            self.p.label(".LSs%d_%%=" % c)
        else:
            self.p.label(".LS%d_%%=" % c)
            if act_fast:
                self.e("s_cbranch_vccnz", (), [".LSs%d_%%=" % c])
        # an event makes this wave the slowest of the step while the wave it shares its SIMD with has slack: let it issue first
        prio = self.cfg.get("prio", 1) if ev is not None else 0
        if prio:
            self.e("raw", (), ["s_setprio %d" % prio])
        yp = self.cfg.get("youngprio", 0)   # experiment: the younger wave of a SIMD (wv >= 4) leads for the first part of the chain
        if yp:
            lab = self.p.newlabel("old")
            self.e("s_bitcmp1_b32", (), [S_WV, 2])
            self.e("s_cbranch_scc0", (), [lab])
            self.e("raw", (), ["s_setprio 2"])
            self.p.label(lab)
        self.probe(0)
        partial = not self.cfg.get("trace", False)
        # slim events: the H0 quad read for the row entering slot ev is, one step later, the "row above" of the row entering
        # slot ev + 1 (a wave's four events are consecutive steps): two quads alternate, nothing is read twice; and slots
        # 1..3 use the quad as the row's level-0 value directly
        hn, ha = (HN, HA) if (ev is None or ev % 2 == 0) else (HA, HN)
        # ---- top: everything that travels through LDS is requested first; what the chain needs at once comes first, because
        # the LDS operations of a wave complete in order and the waits below count the requests that may stay outstanding
        if "nolds" not in self.ab:
            self.e("ds_read_b128", BQ, [V_RB[p]], at=0.0)
            self.e("ds_read_b128", TQ, [V_RT[p]], at=0.0)
        if ph == 0 and not nocook:   # a new cooking group enters: g = step number div 3
            self.e("s_add_u32", S_TG, [S_TG, 16])
            self.wrap(S_G3S, 4 * SLOT, NSLOTS * SLOT)
            self.wrap(S_G3H, 16, 4 * NSLOTS)
        n_after = 0   # LDS requests behind those the mid-step wait needs
        if ev is not None:
            self.fetch_event(ev)
            n_after += 1
            n_after += self.top_ring_reads(ev, hn)
            if "noevlds" in self.ab:
                n_after = 1
        # work that does not depend on the boundary rows goes in front of their wait: this wave's (and its SIMD partner's) LDS
        # round trip is otherwise covered by the 14 instructions of the deferred tail only
        if dma_wait and "nocookwait" not in self.ab:
            self.p.waitcnt(vm=0)    # the row this wave requested 3 .. 5 steps ago has landed: this step's barrier publishes it
            self.probe(1)
        dma_top = self.cfg.get("dma_top", False)
        if dma and dma_top:
            self.issue_row(*dma)    # descriptor fetched at the end of the step before
        if cookw and cook_top:
            self.cook_math()
        self.tail((c - 1) % LV, skip_above1=(ev == 1))
        if ev == 0:  # slot 0's self taps were still needed by the deferred tail
            for k in self.early_planes(0):
                self.ring_read(WT(0, k), 0, k, at=0.0)
            if "noevlds" not in self.ab:
                n_after += len(self.early_planes(0))
        self.p.waitcnt(lgkm=min(n_after, 15) if partial else 0)
        self.probe(2)
        if ev is not None and not partial:
            self.take_event(ev)
        if dma and not dma_top:
            self.issue_row(*dma)    # (its ~30 scalar instructions spread between the chain's FMAs)
        if cookw:
            if not cook_top:
                self.cook_math()
            self.cook_writes()
        if "maskhack" in self.ab:   # timing experiment (WRONG RESULTS): 12 of 64 lanes switched off for the chain -- does the
            self.e("s_mov_b32", R_EXEC_LO, [0xffffffc0])   # forward get faster when the VALU moves less (power), at equal issue?
            self.e("s_mov_b32", R_EXEC_HI, [0x03ffffff])
        # received boundary rows
        self.shift(BQ, D_BQ)
        self.push_below(3, BQ, D_BQ, N1[3])
        self.shift(TQ, D_TQ)
        self.push_above(0, TQ, D_TQ, N1[0])
        for j in (3, 2, 1, 0):
            vq = N1[j]
            tq = D_SLOT.get(j)
            if ev == j:
                if partial:   # the header and the ring quads of the event were requested at the top of the step
                    self.p.waitcnt(lgkm=0)
                    self.take_event(ev)
                self.retire(j, vq)
                self.inject(j, vq, hn, copy=not (j > 0))
                if j > 0:
                    vq = hn
            elif "noact" not in self.ab and (slow or not act_fast):
                self.act_check(j, vq)
            if j == 3 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR[p], vq], offset=1024, at=0.0)
            if j == 0 and "nolds" not in self.ab:
                self.e("ds_write_b128", (), [V_WR[p], vq], offset=0, at=0.0)
            if j == 0:
                break  # slot 0's own pushes: tail(), at the top of the next step
            if yp and j == yp:
                self.e("raw", (), ["s_setprio %d" % prio])
            self.shift(vq, tq)
            self.push_below(j - 1, vq, tq, N1[j - 1])
            if ev == j:
                self.push_self(j, vq, tq, N2[j], init=WT(j, 8))
                self.shift(ha, D_BQ)
                self.push_above(j, ha, D_BQ, N2[j])
            else:
                self.push_self(j, vq, tq, N2[j])
            if j < 3:
                self.push_above(j + 1, vq, tq, N1[j + 1], init=WT(j + 1, 8))
        if "maskhack" in self.ab:
            self.e("s_mov_b64", EXEC, [-1])
        if cookr:
            # the raw values of the task this wave cooks in the next step (group g + 1); nobody waits for them before the barrier
            self.e("s_add_u32", T[2], [S_TG, S_CK4])
            self.mov(V_DC, T[2])
            self.e("ds_read_b32", V_DC, [V_DC])
            self.e("s_add_u32", S_CSLOT, [S_G3S, S_JJS])
            self.wrap(S_CSLOT, 0, NSLOTS * SLOT)
            self.e("s_add_u32", S_CSLOT, [S_CSLOT, S_POOL])
            self.raw_reads()
        if dma_next:
            # descriptor of the row whose DMA this wave issues at the top of the next step (g is one further there if that is a
            # step = 0 mod 3)
            jj, dg = dma_next
            self.e("s_add_u32", T[2], [S_TG, 16 * (dg + (1 if (c + 1) % 3 == 0 else 0)) - 4 + 4 * jj])
            self.mov(V_TMP, T[2])
            self.e("ds_read_b32", V_TMP, [V_TMP])
        self.probe(3)
        if prio:
            self.e("raw", (), ["s_setprio 0"])
        self.p.waitcnt(lgkm=0)
        if cookr:
            self.e("v_readfirstlane_b32", S_CFLAGS, [V_DC])
        if dma_next:
            self.e("v_readfirstlane_b32", S_CD, [V_TMP])
        self.probe(4)
        if "nobar" not in self.ab:
            self.e("s_barrier")
        self.probe(5)
        self.trace_flush(c, dma_wait)
        if not tau3 or ph == 2 or nocook:
            self.e("s_sub_u32", S_TAU, [S_TAU, 1])       # S_TAU counts the remaining steps (tau3: triples of steps) down
            self.e("s_cbranch_scc1", (), [".Lexit_%="])  # the borrow ends the loop
        if slow or c == LV - 1:
            self.e("s_branch", (), [nxt])

    def top_ring_reads(self, ev, hn):
        """the ring requests an event step makes before its mid-step wait"""
        m = {"at": 0.0}
        n = 0
        self.e("v_add_u32", V_RINGE, [S_EGRP, V_L16])
        self.ring_read(hn, ev, 9, **m)
        n += 1
        if ev > 0:
            for k in self.early_planes(ev):
                self.ring_read(WT(ev, k), ev, k, **m)
            n += len(self.early_planes(ev))
        return n

    # ---------------------------------------------------------------------------------- DMA of a raw row
    def issue_row(self, jj, dg, prologue=False):
        """the whole raw row with descriptor S_CD -> its slot.  Loop: row jj of the cooking group dg ahead of the one entering
        now (slot / header offsets from S_G3S / S_G3H); prologue: T[0] = slot offset in the pool, T[1] = header offset.
        Pair records (planes kr | kl, 32 lanes each) for both halves -- the second half through the instruction's immediate
        offset, which moves the global source AND the LDS destination (profiles/r03_ubench_dma.txt) --, then blur, then sparse /
        previous-pass depth.  No clamping: a row above / below the image, or the all-zero descriptor of an inactive row, still
        addresses memory inside the tensors (the cooking task zeroes what lies outside the image)."""
        e = self.e
        n_p = 9 + (1 if self.aux else 0)
        m = (lambda i: {}) if prologue else (lambda i: {"at": self.cfg.get("dma_at", 0.02) + self.cfg.get("dma_span", 0.9) * i / n_p})
        if not prologue:
            e("s_add_u32", T[0], [S_G3S, ((4 * dg + jj - 1) % NSLOTS) * SLOT])
            self.wrap(T[0], 0, NSLOTS * SLOT)
            e("s_add_u32", T[1], [S_G3H, 4 * ((4 * dg + jj - 1) % NSLOTS)])
            self.wrap(T[1], 0, 4 * NSLOTS)
        e("s_bfe_u32", T[3], [S_CD, S_YBFE])               # y
        e("s_lshr_b32", T[4], [S_CD, S_BSH])               # image
        e("s_mul_i32", T[3], [T[3], S_W4])
        e("s_add_u32", T[3], [T[3], S_P04])                # y * 4W + 4 * p0
        e("s_mul_i32", T[5], [T[4], S_G8])
        e("s_mul_hi_u32", T[6], [T[4], S_G8])              # image * 8 planes (64 bit)
        e("s_add_u32", T[5], [T[5], T[3]])
        e("s_addc_u32", T[6], [T[6], 0])
        e("s_add_u32", GB[0], [S_GD[0], T[5]])
        e("s_addc_u32", GB[1], [S_GD[1], T[6]])
        e("s_mul_i32", T[4], [T[4], S_HW4])
        e("s_add_u32", T[3], [T[3], T[4]])                 # byte offset of the row in a 1-channel tensor
        e("s_add_u32", BX[0], [S_BLUR[0], T[3]])
        e("s_addc_u32", BX[1], [S_BLUR[1], 0])
        if self.aux:
            src = S_SP if self.sparse else S_HIN
            e("s_add_u32", AX[0], [src[0], T[3]])
            e("s_addc_u32", AX[1], [src[1], 0])
        # slot header for the event that will inject the row: offset | owned | active
        e("s_bfe_u32", T[7], [S_CD, (1 << 16) | F_OWNED])
        e("s_or_b32", T[3], [T[3], T[7]])
        e("s_and_b32", T[7], [S_CD, 1])
        e("s_lshl_b32", T[7], [T[7], H_ACTIVE])
        e("s_or_b32", T[3], [T[3], T[7]])
        self.mov(V_HDR, T[3])
        e("s_add_u32", T[1], [T[1], S_HDRB])
        self.mov(V_ADR[0], T[1])
        e("ds_write_b32", (), [V_ADR[0], V_HDR])   # (every lane the same dword)
        if "nocookload" in self.ab:
            return
        if "sameload" in self.ab:      # timing experiment: every request reads the same (cache-resident) rows
            e("s_mov_b64", GB, [S_GD])
            e("s_mov_b64", BX, [S_BLUR])
        if "dma1lane" in self.ab:      # timing experiment: the requests move 16 bytes instead of 1 KiB
            e("s_mov_b64", EXEC, [1])
        e("s_add_u32", S_DMB, [T[0], S_POOL])
        i = 0
        for P in range(4):
            for h in (0, 1):
                e("s_add_u32", M0, [S_DMB, (2 * P + h) * 1024 - 512 * h], **m(i))
                e("global_load_lds_dwordx4", (), [V_OFFX[P], GB, M0], offset=512 * h, cache=self.cfg.get("ld_cache"), **m(i))
                i += 1
        e("s_add_u32", M0, [S_DMB, REC_BLUR * 1024], **m(i))
        e("global_load_lds_dwordx4", (), [V_L16, BX, M0], cache=self.cfg.get("ld_cache"), **m(i))
        if self.aux:
            i += 1
            e("s_add_u32", M0, [S_DMB, REC_AUX * 1024], **m(i))
            e("global_load_lds_dwordx4", (), [V_L16, AX, M0], cache=self.cfg.get("ld_cache"), **m(i))
        if "dma1lane" in self.ab:
            e("s_mov_b64", EXEC, [-1])

    # ---------------------------------------------------------------------------------- cooking
    def raw_reads(self):
        """raw values of the wave's task (slot S_CSLOT, half wv & 1) -> PEND_*.  64 lanes read 64 consecutive
        dwords per half-instruction: no bank conflicts.  ds_read2st64_b32 offsets count 256-byte units: pair record P starts at
        8 P (the half's 1 KiB is in the address), its second plane 2 units further for the plain M reads"""
        if "nocookread" in self.ab:
            return
        e = self.e
        e("s_add_u32", T[3], [S_CSLOT, S_H1K])
        e("v_add_u32", V_ADR[0], [T[3], V_CKR])
        e("v_add_u32", V_ADR[1], [T[3], V_CKL])
        for P, (qr, ql) in enumerate([(0, 2), (3, 4), (5, 7)]):
            e("ds_read2st64_b32", PEND_G[qr], [V_ADR[0]], offset0=8 * P, offset1=8 * P + 1)
            e("ds_read2st64_b32", PEND_G[ql], [V_ADR[1]], offset0=8 * P, offset1=8 * P + 1)
        e("v_lshl_add_u32", V_ADR[0], [V_LANE, 2, T[3]])
        e("ds_read2st64_b32", PEND_G[1], [V_ADR[0]], offset0=24, offset1=25)
        e("ds_read2st64_b32", PEND_G[6], [V_ADR[0]], offset0=26, offset1=27)
        e("s_add_u32", T[3], [S_CSLOT, S_H512])
        e("v_lshl_add_u32", V_ADR[1], [V_LANE, 2, T[3]])
        e("ds_read2st64_b32", PEND_BLUR, [V_ADR[1]], offset0=4 * REC_BLUR, offset1=4 * REC_BLUR + 1)
        if self.aux:
            e("ds_read2st64_b32", PEND_SP if self.sparse else PEND_HIN, [V_ADR[1]], offset0=4 * REC_AUX, offset1=4 * REC_AUX + 1)

    def cook_math(self):
        """normalise + fold the task whose raw values were requested at the top of this step (descriptor in V_DC):
        PEND_G[q] <- folded coefficient quad halves, PEND_SP <- c', PEND_BLUR / PEND_HIN keep the level-0 value"""
        g = PEND_G
        norm = self.norm
        e = self.e
        if "nocookmath" in self.ab:
            return
        # most tasks: nothing to patch (active row, the rows above and below inside the image, interior band).  The patches live
        # out of line, so that the arithmetic below shares one scheduling region with the step's chain.
        stub_r, back_r = self.p.newlabel("ckrow"), self.p.newlabel("ckrowb")
        stub_c, back_c = self.p.newlabel("ckcol"), self.p.newlabel("ckcolb")
        e("s_and_b32", T[0], [S_CFLAGS, 7])
        e("s_cmp_eq_u32", (), [T[0], 7])
        e("s_cbranch_scc0", (), [stub_r])
        self.p.label(back_r)
        if self.sited:
            e("s_and_b32", T[0], [S_GEOM, (1 << G_FIRST) | (1 << G_LAST)])
            e("s_cbranch_scc1", (), [stub_c])
            self.p.label(back_c)
        self.cstubs.append((stub_r, back_r, stub_c, back_c))
        sx, sy, tt, scale, t2 = CK_SX, CK_SY, CK_TT, CK_SCALE, CK_T2
        cc = PEND_SP
        ex, ey = t2[0], t2[1]
        om = CK_TT  # sparse only: reuse tt once it is dead
        h0 = PEND_BLUR
        if norm == 1:
            for k in range(8):
                e("v_and_b32", g[k][0], [0x7fffffff, g[k][0]])
                e("v_and_b32", g[k][1], [0x7fffffff, g[k][1]])
        if norm != 2:
            e("v_add_f32", sx, [g[0][0].abs(), g[1][0].abs()])
            e("v_add_f32", sy, [g[0][1].abs(), g[1][1].abs()])
            for k in range(2, 8):
                e("v_add_f32", sx, [sx, g[k][0].abs()])
                e("v_add_f32", sy, [sy, g[k][1].abs()])
            if norm == 0:
                e("v_pk_add_f32", tt, [g[0], g[1]])
                for k in range(2, 8):
                    e("v_pk_add_f32", tt, [tt, g[k]])
            e("v_rcp_f32", scale[0], [sx])
            e("v_rcp_f32", scale[1], [sy])
            if self.cfg.get("newton", False):   # v_rcp_f32 is 1 ulp: one Newton step buys nothing the 1e-4 gate can see
                e("v_fma_f32", ex, [-sx, scale[0], 1.0])
                e("v_fma_f32", ey, [-sy, scale[1], 1.0])
                e("v_fma_f32", scale[0], [ex, scale[0], scale[0]])
                e("v_fma_f32", scale[1], [ey, scale[1], scale[1]])
            if norm == 0:
                e("v_pk_mul_f32", t2, [tt, scale])
            else:
                self.mov(tt[0], sx)
                self.mov(tt[1], sy)
                e("v_pk_mul_f32", t2, [tt, scale])
        if self.sparse:
            # m = sign(sparse) (NaN / 0 pass through), cspn.py:64,81 -- before c' takes PEND_SP over
            mreg = CK_M
            for i in (0, 1):
                self.mov(mreg[i], PEND_SP[i])
                e("v_cmp_gt_f32", S(T[4].i, 2), [PEND_SP[i], 0])
                e("v_cndmask_b32", mreg[i], [mreg[i], 1.0, S(T[4].i, 2)])
                e("v_cmp_lt_f32", S(T[6].i, 2), [PEND_SP[i], 0])
                e("v_cndmask_b32", mreg[i], [mreg[i], -1.0, S(T[6].i, 2)])
        if norm != 2:
            self.fma(cc, t2, h0, h0, neg_lo=[1, 0, 0], neg_hi=[1, 0, 0], keep=True)  # (1 - sigma) * H0
        else:
            self.mov(cc[0], 0)
            self.mov(cc[1], 0)
        if self.sparse:
            for i in (0, 1):
                e("v_sub_f32", om[i], [1.0, mreg[i]])
            if norm != 2:
                e("v_pk_mul_f32", scale, [scale, om])
            else:
                self.mov(scale[0], om[0])
                self.mov(scale[1], om[1])
            e("v_pk_mul_f32", t2, [mreg, h0])
            self.fma(cc, om, cc, t2, keep=True)
        for k in range(8):
            if norm != 2 or self.sparse:
                e("v_pk_mul_f32", g[k], [g[k], scale])

    def emit_cook_stubs(self, stub_r, back_r, stub_c, back_c):
        e, g = self.e, PEND_G
        # ---- rows: inactive (separator / padding) row, or the row above / below lies outside the image
        self.p.label(stub_r)
        l_act = self.p.newlabel("ckact")
        e("s_bitcmp1_b32", (), [S_CFLAGS, F_ACTIVE])
        e("s_cbranch_scc1", (), [l_act])
        # an inactive row must come out as zeros (level-0 value and c' above all: the slot is pinned to zero, but it pushes
        # its value in the step it enters).  Inputs that cook to exactly that without a 0 / 0: one unit coefficient, no depth.
        for k in range(8):
            self.mov(g[k][0], 1.0 if k == 0 else 0)
            self.mov(g[k][1], 1.0 if k == 0 else 0)
        for q in (PEND_SP, PEND_BLUR, PEND_HIN):
            self.mov(q[0], 0)
            self.mov(q[1], 0)
        e("s_branch", (), [back_r])
        self.p.label(l_act)
        if self.sited:
            # rows above / below the image were read from whatever lies there in the tensor (always inside it: the planes
            # read at dy = -1 are channels 5..7, those at dy = +1 channels 0..2): they count as zero
            for flag, quads in ((F_UP, (0, 1, 2)), (F_DN, (5, 6, 7))):
                lab = self.p.newlabel("edge")
                e("s_bitcmp1_b32", (), [S_CFLAGS, flag])
                e("s_cbranch_scc1", (), [lab])
                for k in quads:
                    self.mov(g[k][0], 0)
                    self.mov(g[k][1], 0)
                self.p.label(lab)
        e("s_branch", (), [back_r])
        if self.sited:
            # ---- image-edge columns: the dx = -1 planes at column 0 (half 0, lane 0, first pixel) and the dx = +1 planes at column
            # 255 (half 1, lane 63, second pixel) both sit in the L quads (kl for columns 0,1; kr for columns 2,3)
            self.p.label(stub_c)
            e("s_bitcmp1_b32", (), [S_GEOM, G_FIRST])
            e("s_cselect_b64", S_EL, [S_ELC, 0])
            e("s_bitcmp1_b32", (), [S_GEOM, G_LAST])
            e("s_cselect_b64", S_ER, [S_ERC, 0])
            for q in (2, 4, 7):
                e("v_cndmask_b32", g[q][0], [g[q][0], 0, S_EL])
                e("v_cndmask_b32", g[q][1], [g[q][1], 0, S_ER])
            e("s_branch", (), [back_c])

    def cook_writes(self, spread=True):
        """the cooked quad halves of the task normalised in the step before -> slot S_CSLOT, in place of the raw records
        (every raw read of the row happened before the barrier in between)"""
        e = self.e
        e("s_add_u32", T[3], [S_CSLOT, S_H512])
        e("v_add_u32", V_ADR[0], [T[3], V_CKW])
        hv = PEND_HIN if self.hin else PEND_BLUR
        items = [(k, PEND_G[k]) for k in range(8)] + [(8, PEND_SP), (9, hv)]
        if "nocookwrite" not in self.ab:
            for i, (q, reg) in enumerate(items):   # 64 lanes write 64 different dwords of 256 consecutive bytes: no bank conflicts
                m = {"at": self.cfg.get("cw_at", 0.45) + self.cfg.get("cw_span", 0.45) * i / len(items)} if spread else {}
                e("ds_write2st64_b32", (), [V_ADR[0], reg[0], reg[1]], offset0=4 * q, offset1=4 * q + 1, **m)

    # ---------------------------------------------------------------------------------- prologue
    def prologue(self):
        e = self.e
        # LDS below the descriptor table (boundary rows, row slots, headers) was zeroed by the C++ part of the kernel
        e("v_lshlrev_b32", V_L16, [4, V_LANE])
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
        if self.cfg.get("tau3", True) and "nocook" not in self.ab:   # whole triples of steps: floor(last / 3) more after the first
            e("s_mul_hi_u32", S_TAU, [S_LAST, 0x55555556])            # (exact for last < 2^31; last = -1 never gets here)
        else:
            e("s_mov_b32", S_TAU, [S_LAST])
        for j in range(4):
            e("s_mov_b32", S_SLOTB[j], [0])
            e("s_mov_b32", S_AM[j], [0])
        e("s_mov_b64", VCC, [1])
        e("s_and_b32", T[2], [S_LOHI, 0xffff])
        e("s_lshr_b32", T[3], [S_LOHI, 16])
        e("v_lshlrev_b32", V_TMP, [2, V_LANE])
        e("v_cmp_ge_u32", S(T[4].i, 2), [V_TMP, T[2]])
        e("v_cmp_lt_u32", S(T[6].i, 2), [V_TMP, T[3]])
        e("s_and_b64", S_OMASK, [S(T[4].i, 2), S(T[6].i, 2)])
        e("s_and_b32", T[1], [S_WV, 1])            # T1 = wave parity (= task kind: 0 X, 1 Y)
        e("s_lshr_b32", T[2], [S_WV, 1])           # T2 = wv >> 1 (= which row of a cooking group)
        # descriptor decoding: y = bfe(d, 4, ybits), image = d >> (4 + ybits)
        e("s_and_b32", T[3], [S_GEOM, 0xff])
        e("s_lshl_b32", S_YBFE, [T[3], 16])
        e("s_or_b32", S_YBFE, [S_YBFE, 4])
        e("s_add_i32", S_BSH, [T[3], 4])
        e("s_lshl_b32", S_G8, [S_HW4, 3])
        e("s_add_i32", S_ROWE, [S_LDSB, LDS_ROWS + NSLOTS * SLOT])
        e("s_add_i32", S_HDRE, [S_LDSB, LDS_HDR + 4 * NSLOTS])
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
        # cooking lane bases
        e("v_lshlrev_b32", CK[2], [2, V_LANE])             # 4 * i
        e("v_and_b32", V_TMP, [2, V_LANE])                 # 2 for columns 2, 3
        e("v_lshlrev_b32", V_TMP, [8, V_TMP])              # 512 for columns 2, 3
        e("v_add_u32", V_CKR, [CK[2], V_TMP])
        e("v_xor_b32", V_TMP, [512, V_TMP])
        e("v_add_u32", V_CKL, [CK[2], V_TMP])
        # position of column c in the quad (c0, c3, c1, c2): 0, 2, 3, 1 = 2 * ((c & 1) ^ (c >> 1)) + (c >> 1)
        e("v_and_b32", CK[0], [1, V_LANE])                 # c & 1
        e("v_lshrrev_b32", CK[1], [1, V_LANE])
        e("v_and_b32", CK[1], [1, CK[1]])                  # c >> 1
        e("v_xor_b32", CK[0], [CK[0], CK[1]])
        e("v_lshlrev_b32", CK[0], [1, CK[0]])
        e("v_add_u32", CK[0], [CK[0], CK[1]])              # pos
        e("v_lshlrev_b32", CK[0], [2, CK[0]])
        e("v_lshrrev_b32", V_CKW, [2, V_LANE])
        e("v_lshlrev_b32", V_CKW, [4, V_CKW])
        e("v_add_u32", V_CKW, [V_CKW, CK[0]])
        e("s_lshl_b32", S_H1K, [T[1], 10])
        e("s_lshl_b32", S_H512, [T[1], 9])
        e("v_and_b32", V_TMP, [31, V_LANE])
        e("v_lshlrev_b32", V_TMP, [4, V_TMP])              # 16 * (lane & 31)
        # DMA lane offsets of pair record P (half 0; half 1 = + 512 bytes = the instruction's immediate offset): lanes 0..31 fetch 16
        # bytes of the first plane of the pair, lanes 32..63 of the second: plane * HW4 + dy * W4 + 4 * (4 * (lane & 31) + dx)
        e("v_cmp_lt_u32", S(T[4].i, 2), [V_LANE, 32])
        for P in range(4):
            for which in (0, 1):      # which plane of the pair
                k = PAIRS[P][which]
                dst = T[8 + which]
                e("s_mul_i32", dst, [S_HW4, k])
                if self.sited:
                    if DY[k] > 0:
                        e("s_add_i32", dst, [dst, S_W4])
                    if DY[k] < 0:
                        e("s_sub_i32", dst, [dst, S_W4])
                    if DX[k] != 0:
                        e("s_add_i32", dst, [dst, 4 * DX[k]])
            self.mov(CK[0], T[8])
            self.mov(CK[1], T[9])
            e("v_cndmask_b32", V_OFFX[P], [CK[1], CK[0], S(T[4].i, 2)])
            e("v_add_u32", V_OFFX[P], [V_OFFX[P], V_TMP])
        e("s_add_i32", S_POOL, [S_LDSB, LDS_ROWS])
        e("s_add_i32", S_HDRB, [S_LDSB, LDS_HDR])
        e("s_lshr_b32", T[2], [S_WV, 1])
        e("s_add_i32", T[3], [T[2], 3])
        e("s_mul_i32", S_JJS, [T[3], SLOT])
        e("s_lshl_b32", T[3], [T[2], 2])
        e("s_add_i32", S_CK4, [T[3], 12])
        e("s_cmp_eq_u32", (), [T[1], 0])
        # constant edge-lane masks (half 0: lane 0 = image column 0 of a first band; half 1: lane 63 (pixel 127) = its last column)
        e("s_cselect_b32", S_ELC[0], [1, 0])
        e("s_mov_b32", S_ELC[1], [0])
        e("s_mov_b32", S_ERC[0], [0])
        e("s_cselect_b32", S_ERC[1], [0, 0x80000000])
        # ring counters: wave 7's slot 3 fires at step 0 for the (inactive) row -1
        e("s_lshl_b32", S_QB, [S_WV, 2])
        e("s_cmp_eq_u32", (), [S_WV, 7])
        e("s_cselect_b32", S_QB, [-4, S_QB])
        # slot group of the first burst: rows S_QB .. S_QB + 3 -> slots (S_QB mod 12) ..
        e("s_add_i32", T[3], [S_QB, 12])                   # -4 -> 8
        e("s_cmp_ge_u32", (), [T[3], 24])
        e("s_cselect_b32", T[4], [24, 0])
        e("s_sub_u32", T[3], [T[3], T[4]])
        e("s_cmp_ge_u32", (), [T[3], 12])
        e("s_cselect_b32", T[4], [12, 0])
        e("s_sub_u32", T[3], [T[3], T[4]])                 # S_QB mod 12 (S_QB in -4 .. 28)
        e("s_lshl_b32", T[4], [T[3], 2])
        e("s_add_i32", T[4], [T[4], LDS_HDR])
        e("s_add_i32", S_EHDR, [T[4], S_LDSB])
        e("s_mul_i32", T[3], [T[3], SLOT])
        e("s_add_i32", T[3], [T[3], LDS_ROWS])
        e("s_add_i32", S_EGRP, [T[3], S_LDSB])
        # the workgroup's descriptor table: already in LDS (written by the C++ part of the kernel before this block)
        e("s_add_i32", T[3], [S_LDSB, LDS_TAB])
        e("s_add_i32", S_TABB, [T[3], PADF * DESC_BYTES])
        self.p.waitcnt(lgkm=0)
        e("s_barrier")                               # LDS zero-fill and the table are complete
        # cooking pipeline: the wave's task of group g = row 4g - 1 + (wv >> 1), half wv & 1.  The rows the loop expects to be on
        # their way when it starts (groups 0 and 1, rows 0 and 1 of group 2 = stream rows -1 .. 8) are requested here -- wave w row
        # w - 1, waves 0 and 1 also row 7 + w --, and group 0 is cooked synchronously.
        def request(rowreg):   # rowreg: stream row (>= -1); T[0], T[1] <- slot / header offset of row mod 12
            e("s_add_i32", T[0], [rowreg, 12])
            e("s_cmp_ge_u32", (), [T[0], 12])
            e("s_cselect_b32", T[1], [12, 0])
            e("s_sub_u32", T[0], [T[0], T[1]])
            e("s_lshl_b32", T[1], [T[0], 2])
            e("s_mul_i32", T[0], [T[0], SLOT])
            e("s_lshl_b32", T[2], [rowreg, 2])
            e("s_add_i32", T[2], [T[2], S_TABB])
            self.mov(V_DC, T[2])
            e("ds_read_b32", V_DC, [V_DC])
            self.p.waitcnt(lgkm=0)
            e("v_readfirstlane_b32", S_CD, [V_DC])
            self.issue_row(0, 0, prologue=True)
        e("s_add_i32", T[10], [S_WV, -1])
        request(T[10])
        l_one = self.p.newlabel("onerow")
        e("s_cmp_ge_u32", (), [S_WV, 2])
        e("s_cbranch_scc1", (), [l_one])
        e("s_add_i32", T[10], [S_WV, 7])
        request(T[10])
        self.p.label(l_one)
        # g = -1 until the first step of the loop counts it up
        e("s_add_i32", S_TG, [S_TABB, -16])
        e("s_mov_b32", S_G3S, [8 * SLOT])
        e("s_mov_b32", S_G3H, [32])
        self.p.waitcnt(vm=0, lgkm=0)
        e("s_barrier")
        # group 0: rows -1 .. 2 in slots 11, 0, 1, 2
        e("s_lshr_b32", T[2], [S_WV, 1])
        e("s_add_i32", T[3], [T[2], 11])
        e("s_cmp_ge_u32", (), [T[3], 12])
        e("s_cselect_b32", T[4], [12, 0])
        e("s_sub_u32", T[3], [T[3], T[4]])
        e("s_mul_i32", T[3], [T[3], SLOT])
        e("s_add_i32", S_CSLOT, [T[3], S_POOL])
        e("s_lshl_b32", T[2], [T[2], 2])
        e("s_add_i32", T[2], [T[2], S_TABB])
        e("s_add_i32", T[2], [T[2], -4])
        self.mov(V_DC, T[2])
        e("ds_read_b32", V_DC, [V_DC])
        self.raw_reads()
        self.p.waitcnt(lgkm=0)
        e("v_readfirstlane_b32", S_CFLAGS, [V_DC])
        self.cook_math()
        e("s_barrier")                               # every raw read of group 0 happened: cook in place
        self.cook_writes(spread=False)
        # waves whose first counter is a requesting one (step 0: g = 0) need that row's descriptor now
        for w in range(NW):
            c0 = (LV - 3 * w) % LV
            if c0 in self.dma_issue and "nocook" not in self.ab:
                jj, dg = self.dma_issue[c0]
                lab = self.p.newlabel("nofirst")
                e("s_cmp_lg_u32", (), [S_WV, w])
                e("s_cbranch_scc1", (), [lab])
                e("s_add_i32", T[2], [S_TABB, 16 * dg - 4 + 4 * jj])
                self.mov(V_TMP, T[2])
                e("ds_read_b32", V_TMP, [V_TMP])
                self.p.waitcnt(lgkm=0)
                e("v_readfirstlane_b32", S_CD, [V_TMP])
                self.p.label(lab)
        self.p.waitcnt(lgkm=0)
        e("s_barrier")
        if self.cfg.get("trace", False):
            e("s_mul_i32", T[3], [S_WV, self.TRACE_BYTES])
            e("v_lshlrev_b32", V(75), [2, V_LANE])
            e("v_add_u32", V(75), [T[3], V(75)])
            e("v_mov_b32", V(74), [0])
        for w in range(NW):
            c0 = (LV - 3 * w) % LV
            e("s_cmp_eq_u32", (), [S_WV, w])
            e("s_cbranch_scc1", (), [".LS%d_%%=" % c0])

    def build(self):
        self.prologue()
        for c in range(LV):
            self.step(c)
        self.p.label(".Lexit_%=")
        self.e("s_waitcnt", vmcnt=0)                  # no LDS-DMA may be in flight when the workgroup's LDS is released
        self.e("s_branch", (), [".Lend_%="])
        if self.cfg.get("act_fast", True) and "noact" not in self.ab:
            for c in range(LV):
                self.step(c, slow=True)
        for st in self.cstubs:
            self.emit_cook_stubs(*st)
        self.p.label(".Lend_%=")
        return self.p


def build(cfg, sched=True):
    from tools.tswgen import isa
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
    return p
