"""tools/tswgen/emu.py -- CPU interpreter for the instruction subset tools/tswgen/isa.py can emit.

One workgroup = NW waves executed round-robin between s_barrier's.  Besides the arithmetic it checks what a GPU would
not tell us politely: use of a register whose load has not been waited for (s_waitcnt placement), LDS races inside a
barrier epoch, out-of-range LDS / global accesses, misaligned multi-dword accesses.  fp32 FMA is evaluated in float64
and rounded once (double rounding is far below the 1e-4 parity tolerance).
"""
import numpy as np

from .isa import R, DS_OPS, VMEM_LD, VMEM_ST, SMEM, VMEM_LDS

F32 = np.float32
U32 = np.uint32


class EmuError(Exception):
    pass


class Wave(object):
    def __init__(self, wid):
        self.wid = wid
        self.v = np.zeros((256, 64), U32)
        self.s = np.zeros(128, U32)
        self.scc = 0
        self.m0 = 0
        self.vcc = np.uint64(0)
        self.exec = np.uint64(0xFFFFFFFFFFFFFFFF)
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vm_q = []    # in-order queue of sets of pending registers (global loads; stores = empty set)
        self.lds_q = []   # in-order queue (LDS)
        self.sm_pending = set()  # scalar loads return out of order: only lgkmcnt(0) clears them
        self.pending = {}  # reg -> count of outstanding loads
        self.icount = {}


def _mask_to_bool(m):
    return ((np.uint64(m) >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool)


def _bool_to_mask(b):
    return np.uint64(int(np.sum(b.astype(np.uint64) << np.arange(64, dtype=np.uint64))))


class Emu(object):
    def __init__(self, prog, mem, lds_bytes, nwaves=8):
        self.ins = prog.ins
        self.labels = {i.src[0]: k for k, i in enumerate(self.ins) if i.op == "label"}
        self.mem = mem  # np.uint8 flat global memory
        self.mem32 = mem.view(U32)
        self.lds = np.zeros(lds_bytes // 4, U32)
        self.lds_writer = np.full(lds_bytes // 4, -1, np.int64)  # wave that wrote the dword in the current epoch
        self.lds_epoch = np.full(lds_bytes // 4, -1, np.int64)
        self.lds_reader = np.full(lds_bytes // 4, -1, np.int64)  # wave that read it in the current epoch (-2: several)
        self.lds_repoch = np.full(lds_bytes // 4, -1, np.int64)
        self.lds_dma = np.zeros(lds_bytes // 4, bool)   # dwords an LDS-DMA load in flight will overwrite at an unknown time
        self.epoch = 0
        self.waves = [Wave(w) for w in range(nwaves)]
        self.nsteps = 0
        self.check_races = True
        self.sched_rng = None

    # ---- operand access -----------------------------------------------------------------------------
    def _chk(self, w, reg):
        if w.pending.get(reg, 0):
            raise EmuError("wave %d pc %d: %s%d read/written while a load into it is outstanding: %s" % (
                w.wid, w.pc, reg[0], reg[1], self.ins[w.pc].text()))

    def rs(self, w, o, k=0):
        """scalar source (dword k of a multi-dword operand) -> python int (u32)"""
        if isinstance(o, R):
            if o.kind == "s":
                self._chk(w, ("s", o.i + k))
                return int(w.s[o.i + k])
            if o.kind == "vcc":
                return int((int(w.vcc) >> (32 * k)) & 0xffffffff)
            if o.kind == "exec":
                return int((int(w.exec) >> (32 * k)) & 0xffffffff)
            if o.kind == "m0":
                return int(w.m0)
            raise EmuError("scalar read of vector register")
        if isinstance(o, float):
            return int(np.array([o], F32).view(U32)[0])
        return int(o) & 0xffffffff

    def rs64(self, w, o):
        if isinstance(o, R):
            return self.rs(w, o, 0) | (self.rs(w, o, 1) << 32)
        v = int(o)
        return v & 0xffffffffffffffff

    def ws(self, w, o, val, k=0):
        val = int(val) & 0xffffffff
        if o.kind == "s":
            self._chk(w, ("s", o.i + k))
            w.s[o.i + k] = val
        elif o.kind == "vcc":
            cur = int(w.vcc)
            cur = (cur & ~(0xffffffff << (32 * k))) | (val << (32 * k))
            w.vcc = np.uint64(cur)
        elif o.kind == "exec":
            cur = int(w.exec)
            cur = (cur & ~(0xffffffff << (32 * k))) | (val << (32 * k))
            w.exec = np.uint64(cur)
        elif o.kind == "m0":
            w.m0 = val
        else:
            raise EmuError("scalar write to vector register")

    def ws64(self, w, o, val):
        self.ws(w, o, val & 0xffffffff, 0)
        self.ws(w, o, (val >> 32) & 0xffffffff, 1)

    def rv(self, w, o, k=0):
        """vector source dword k as uint32[64]"""
        if isinstance(o, R):
            if o.kind == "v":
                self._chk(w, ("v", o.i + k))
                return w.v[o.i + k]
            return np.full(64, self.rs(w, o, k), U32)
        return np.full(64, self.rs(w, o), U32)

    def rvf(self, w, o, k=0):
        x = self.rv(w, o, k).view(F32)
        if isinstance(o, R):
            if o.absf:
                x = np.abs(x)
            if o.neg:
                x = -x
        return x

    def wv(self, w, o, val, k=0, mask=None):
        self._chk(w, ("v", o.i + k))
        val = np.asarray(val)
        if val.dtype == F32:
            val = val.view(U32)
        else:
            val = val.astype(U32)
        m = _mask_to_bool(w.exec) if mask is None else mask
        w.v[o.i + k] = np.where(m, val, w.v[o.i + k])

    # ---- memory helpers -----------------------------------------------------------------------------
    def lds_access(self, w, addr, ndw, write, lanes, align):
        """addr: uint32[64] byte addresses; returns index array [64][ndw] of dwords"""
        a = addr.astype(np.int64)
        if np.any((a[lanes] % align) != 0):
            raise EmuError("wave %d pc %d: misaligned LDS access %s" % (w.wid, w.pc, self.ins[w.pc].text()))
        idx = (a[:, None] // 4) + np.arange(ndw)[None, :]
        if np.any(idx[lanes] < 0) or np.any(idx[lanes] >= self.lds.size):
            raise EmuError("wave %d pc %d: LDS access out of range %s" % (w.wid, w.pc, self.ins[w.pc].text()))
        idx = np.where(lanes[:, None], idx, 0)
        if np.any(self.lds_dma[idx[lanes].ravel()]):
            raise EmuError("wave %d pc %d: LDS access to bytes an LDS-DMA load in flight is still writing: %s" % (
                w.wid, w.pc, self.ins[w.pc].text()))
        if self.check_races:
            fl = idx[lanes].ravel()
            same = self.lds_epoch[fl] == self.epoch
            other = self.lds_writer[fl] != w.wid
            if np.any(same & other):
                bad = fl[same & other][0]
                raise EmuError("wave %d pc %d: LDS race on dword %d (written by wave %d in this barrier epoch): %s" % (
                    w.wid, w.pc, bad, self.lds_writer[bad], self.ins[w.pc].text()))
            if write:
                rd = (self.lds_repoch[fl] == self.epoch) & (self.lds_reader[fl] != w.wid)
                if np.any(rd):
                    bad = fl[rd][0]
                    raise EmuError("wave %d pc %d: LDS write-after-read race on dword %d (read by wave %d in this epoch): %s" % (
                        w.wid, w.pc, bad, self.lds_reader[bad], self.ins[w.pc].text()))
                self.lds_writer[fl] = w.wid
                self.lds_epoch[fl] = self.epoch
            else:
                cur = self.lds_repoch[fl] == self.epoch
                self.lds_reader[fl] = np.where(cur & (self.lds_reader[fl] != w.wid), -2, w.wid)
                self.lds_repoch[fl] = self.epoch
        return idx

    def gaddr(self, w, ins, voff, sbase, ndw, lanes):
        base = self.rs64(w, sbase)
        a = base + voff.astype(np.int64) + int(ins.mods.get("offset", 0))
        if np.any(a[lanes] % 4):
            raise EmuError("misaligned global access")
        if np.any(a[lanes] < 4096) or np.any(a[lanes] + 4 * ndw > self.mem.size):
            raise EmuError("wave %d pc %d: global access out of range (%d..%d): %s" % (
                w.wid, w.pc, a[lanes].min(), a[lanes].max(), ins.text()))
        idx = (a[:, None] // 4) + np.arange(ndw)[None, :]
        return np.where(lanes[:, None], idx, 1024)

    def lds_dma_load(self, w, ins, ex):
        """global_load_lds_dwordx4 voff, sbase (M0 = LDS byte address of lane 0's 16 bytes): the data is written some time
        between now and the vmcnt wait of this wave that covers the load; until then nobody may touch the destination, and
        other waves only after a barrier behind that wait (modelled as an LDS write of this wave in the epoch of the wait)"""
        s = ins.src
        gidx = self.gaddr(w, ins, self.rv(w, s[0]), s[1], 4, ex)
        dst = int(w.m0) + int(ins.mods.get("offset", 0)) + 16 * np.arange(64, dtype=np.int64)   # (measured: the offset moves both)
        if int(dst[0]) % 16:
            raise EmuError("LDS-DMA destination not 16-byte aligned")
        didx = (dst[:, None] // 4) + np.arange(4)[None, :]
        if didx.max() >= self.lds.size:
            raise EmuError("wave %d pc %d: LDS-DMA destination out of range" % (w.wid, w.pc))
        lanes = ex
        fl = didx[lanes].ravel()
        if np.any(self.lds_dma[fl]):
            raise EmuError("wave %d pc %d: two LDS-DMA loads in flight into the same bytes" % (w.wid, w.pc))
        if self.check_races:
            rd = (self.lds_repoch[fl] == self.epoch) & (self.lds_reader[fl] != w.wid)
            wr = (self.lds_epoch[fl] == self.epoch) & (self.lds_writer[fl] != w.wid)
            if np.any(rd) or np.any(wr):
                raise EmuError("wave %d pc %d: LDS-DMA into bytes another wave accessed in this barrier epoch: %s" % (
                    w.wid, w.pc, ins.text()))
        data = self.mem32[gidx[lanes]].copy()
        self.lds_dma[fl] = True

        def land(fl=fl, data=data.ravel(), wid=w.wid):
            self.lds[fl] = data
            self.lds_dma[fl] = False
            self.lds_writer[fl] = wid
            self.lds_epoch[fl] = self.epoch
        w.vm_q.append(land)

    def _pend(self, w, regs, q):
        for r in regs:
            self._chk(w, r)
        for r in regs:
            w.pending[r] = w.pending.get(r, 0) + 1
        if q is not None:
            q.append(list(regs))

    def _retire(self, w, regs):
        if callable(regs):   # an LDS-DMA load lands when the wave's vmcnt wait covers it
            regs()
            return
        for r in regs:
            w.pending[r] -= 1

    # ---- execution ----------------------------------------------------------------------------------------
    def step_wave(self, w):
        ins = self.ins[w.pc]
        o, d, s, m = ins.op, ins.dst, ins.src, ins.mods
        w.icount[o] = w.icount.get(o, 0) + 1
        nxt = w.pc + 1
        ex = _mask_to_bool(w.exec)
        if o == "label":
            pass
        elif o == "raw":
            pass
        elif o == "s_nop":
            pass
        elif o == "s_endpgm":
            w.done = True
        elif o == "s_barrier":
            w.at_barrier = True
        elif o == "s_sleep":   # elastic kernels poll LDS tags: give the other waves a turn
            w.yielded = True
        elif o == "s_waitcnt":
            if "vmcnt" in m:
                while len(w.vm_q) > m["vmcnt"]:
                    self._retire(w, w.vm_q.pop(0))
            if "lgkmcnt" in m:
                if w.sm_pending and m["lgkmcnt"] == 0:
                    self._retire(w, list(w.sm_pending))
                    w.sm_pending = set()
                    # scalar and LDS share the counter: waiting for 0 drains both
                # with scalar loads outstanding only lgkmcnt(0) is meaningful
                if w.sm_pending and m["lgkmcnt"] != 0:
                    raise EmuError("lgkmcnt(%d) with scalar loads outstanding" % m["lgkmcnt"])
                while len(w.lds_q) > m["lgkmcnt"]:
                    self._retire(w, w.lds_q.pop(0))
        elif o in ("s_branch",):
            nxt = self.labels[s[0]]
        elif o == "s_cbranch_scc0":
            if not w.scc:
                nxt = self.labels[s[0]]
        elif o == "s_cbranch_scc1":
            if w.scc:
                nxt = self.labels[s[0]]
        elif o == "s_cbranch_vccnz":
            if int(w.vcc) != 0:
                nxt = self.labels[s[0]]
        elif o == "s_cbranch_vccz":
            if int(w.vcc) == 0:
                nxt = self.labels[s[0]]
        elif o == "s_cbranch_execz":
            if int(w.exec) == 0:
                nxt = self.labels[s[0]]
        elif o.startswith("s_") and o not in SMEM:
            self.salu(w, ins)
        elif o.startswith("v_"):
            self.valu(w, ins, ex)
        elif o in DS_OPS:
            self.ds(w, ins, ex)
        elif o in VMEM_LD:
            ndw = {"global_load_dword": 1, "global_load_dwordx2": 2, "global_load_dwordx4": 4}[o]
            idx = self.gaddr(w, ins, self.rv(w, s[0]), s[1], ndw, ex)
            regs = [("v", d[0].i + k) for k in range(ndw)]
            if m.get("dummy"):   # a touch whose result nobody reads: several may be outstanding into the same register
                w.vm_q.append([])
            else:
                for k in range(ndw):
                    self.wv(w, d[0], self.mem32[idx[:, k]], k)
                self._pend(w, regs, w.vm_q)
        elif o in VMEM_LDS:
            self.lds_dma_load(w, ins, ex)
        elif o in VMEM_ST:
            ndw = {"global_store_dword": 1, "global_store_dwordx2": 2, "global_store_dwordx4": 4}[o]
            idx = self.gaddr(w, ins, self.rv(w, s[0]), s[2], ndw, ex)
            for k in range(ndw):
                val = self.rv(w, s[1], k)
                self.mem32[idx[ex, k]] = val[ex]
            w.vm_q.append([])
        elif o in SMEM:
            ndw = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8}[o]
            a = self.rs64(w, s[0]) + self.rs(w, s[1])
            if a % 4 or a < 4096 or a + 4 * ndw > self.mem.size:
                raise EmuError("wave %d pc %d: scalar load out of range: %s (addr %d)" % (w.wid, w.pc, ins.text(), a))
            regs = [("s", d[0].i + k) for k in range(ndw)]
            for k in range(ndw):
                self.ws(w, d[0], self.mem32[a // 4 + k], k)
            self._pend(w, regs, None)
            w.sm_pending.update(regs)
        else:
            raise EmuError("unknown op " + o)
        w.pc = nxt

    def salu(self, w, ins):
        o, d, s = ins.op, ins.dst, ins.src
        rs = self.rs

        def i32(x):
            x &= 0xffffffff
            return x - (1 << 32) if x & 0x80000000 else x
        if o == "s_mov_b32":
            self.ws(w, d[0], rs(w, s[0]))
        elif o == "s_mov_b64":
            self.ws64(w, d[0], self.rs64(w, s[0]))
        elif o in ("s_add_u32", "s_add_i32"):
            a, b = rs(w, s[0]), rs(w, s[1])
            r = a + b
            self.ws(w, d[0], r)
            w.scc = int(r > 0xffffffff) if o == "s_add_u32" else int(i32(a) + i32(b) != i32(r))
        elif o == "s_addc_u32":
            r = rs(w, s[0]) + rs(w, s[1]) + w.scc
            self.ws(w, d[0], r)
            w.scc = int(r > 0xffffffff)
        elif o in ("s_sub_i32", "s_sub_u32"):
            a, b = rs(w, s[0]), rs(w, s[1])
            self.ws(w, d[0], a - b)
            w.scc = int(b > a) if o == "s_sub_u32" else int(i32(a) - i32(b) != i32(a - b))
        elif o == "s_subb_u32":
            a, b = rs(w, s[0]), rs(w, s[1]) + w.scc
            self.ws(w, d[0], a - b)
            w.scc = int(b > a)
        elif o == "s_min_u32":
            a, b = rs(w, s[0]), rs(w, s[1])
            self.ws(w, d[0], min(a, b))
            w.scc = int(a < b)
        elif o == "s_mul_hi_u32":
            self.ws(w, d[0], (rs(w, s[0]) * rs(w, s[1])) >> 32)
        elif o == "s_mul_i32":
            self.ws(w, d[0], i32(rs(w, s[0])) * i32(rs(w, s[1])))
        elif o == "s_lshl_b32":
            r = (rs(w, s[0]) << (rs(w, s[1]) & 31)) & 0xffffffff
            self.ws(w, d[0], r)
            w.scc = int(r != 0)
        elif o == "s_lshr_b32":
            r = rs(w, s[0]) >> (rs(w, s[1]) & 31)
            self.ws(w, d[0], r)
            w.scc = int(r != 0)
        elif o == "s_ashr_i32":
            r = i32(rs(w, s[0])) >> (rs(w, s[1]) & 31)
            self.ws(w, d[0], r)
            w.scc = int((r & 0xffffffff) != 0)
        elif o == "s_bfe_u32":
            a, b = rs(w, s[0]), rs(w, s[1])
            off, width = b & 31, (b >> 16) & 0x7f
            r = (a >> off) & ((1 << width) - 1)
            self.ws(w, d[0], r)
            w.scc = int(r != 0)
        elif o == "s_bfe_i32":
            a, b = rs(w, s[0]), rs(w, s[1])
            off, width = b & 31, (b >> 16) & 0x7f
            r = (a >> off) & ((1 << width) - 1)
            if width and (r >> (width - 1)) & 1:
                r -= 1 << width
            self.ws(w, d[0], r)
            w.scc = int((r & 0xffffffff) != 0)
        elif o in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_andn2_b32"):
            a, b = rs(w, s[0]), rs(w, s[1])
            r = {"s_and_b32": a & b, "s_or_b32": a | b, "s_xor_b32": a ^ b, "s_andn2_b32": a & ~b}[o] & 0xffffffff
            self.ws(w, d[0], r)
            w.scc = int(r != 0)
        elif o in ("s_and_b64", "s_or_b64", "s_andn2_b64"):
            a, b = self.rs64(w, s[0]), self.rs64(w, s[1])
            r = {"s_and_b64": a & b, "s_or_b64": a | b, "s_andn2_b64": a & ~b}[o] & 0xffffffffffffffff
            self.ws64(w, d[0], r)
            w.scc = int(r != 0)
        elif o == "s_and_saveexec_b64":
            old = int(w.exec)
            self.ws64(w, d[0], old)
            w.exec = np.uint64(old & self.rs64(w, s[0]))
            w.scc = int(int(w.exec) != 0)
        elif o.startswith("s_cmp_"):
            a, b = rs(w, s[0]), rs(w, s[1])
            kind, ty = o[6:8], o[-3:]
            if ty == "i32":
                a, b = i32(a), i32(b)
            w.scc = int({"eq": a == b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[kind])
        elif o == "s_bitcmp1_b32":
            w.scc = int((rs(w, s[0]) >> (rs(w, s[1]) & 31)) & 1)
        elif o == "s_bitcmp0_b32":
            w.scc = int(not ((rs(w, s[0]) >> (rs(w, s[1]) & 31)) & 1))
        elif o == "s_cselect_b32":
            self.ws(w, d[0], rs(w, s[0]) if w.scc else rs(w, s[1]))
        elif o == "s_cselect_b64":
            self.ws64(w, d[0], self.rs64(w, s[0]) if w.scc else self.rs64(w, s[1]))
        elif o == "s_bfm_b64":
            self.ws64(w, d[0], (((1 << (rs(w, s[0]) & 63)) - 1) << (rs(w, s[1]) & 63)) & 0xffffffffffffffff)
        else:
            raise EmuError("unknown SALU op " + o)

    def valu(self, w, ins, ex):
        o, d, s, m = ins.op, ins.dst, ins.src, ins.mods
        rv, rvf = self.rv, self.rvf
        with np.errstate(all="ignore"):
            if o == "v_mov_b32":
                x = rv(w, s[0])
                if ins.is_dpp():
                    if int(w.exec) != 0xFFFFFFFFFFFFFFFF:
                        raise EmuError("DPP with partial exec not modelled")
                    z = np.zeros(1, U32)
                    if m["dpp"] == "wave_shr:1":
                        x = np.concatenate([z, x[:-1]])
                    elif m["dpp"] == "wave_shl:1":
                        x = np.concatenate([x[1:], z])
                    else:
                        raise EmuError("dpp ctrl " + m["dpp"])
                self.wv(w, d[0], x)
            elif o == "v_pk_mov_b32":
                # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]] (LLVM SIInstrInfo::copyPhysReg uses op_sel:[0,1] for a 64-bit copy)
                osl = m.get("op_sel", [0, 0])
                lo, hi = rv(w, s[0], osl[0]).copy(), rv(w, s[1], osl[1]).copy()
                self.wv(w, d[0], lo, 0)
                self.wv(w, d[0], hi, 1)
            elif o in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
                nsrc = len(s)
                osl = m.get("op_sel", [0] * nsrc)
                osh = m.get("op_sel_hi", [1] * nsrc)
                ngl = m.get("neg_lo", [0] * nsrc)
                ngh = m.get("neg_hi", [0] * nsrc)
                res = []
                for half, sel, ng in ((0, osl, ngl), (1, osh, ngh)):
                    ops = []
                    for k in range(nsrc):
                        if isinstance(s[k], R) and s[k].n >= 2:
                            x = rvf(w, s[k], sel[k])
                        else:
                            x = rvf(w, s[k], 0)
                        ops.append(-x if ng[k] else x)
                    a64 = [x.astype(np.float64) for x in ops]
                    if o == "v_pk_fma_f32":
                        r = (a64[0] * a64[1] + a64[2]).astype(F32)
                    elif o == "v_pk_mul_f32":
                        r = (a64[0] * a64[1]).astype(F32)
                    else:
                        r = (a64[0] + a64[1]).astype(F32)
                    res.append(r)
                self.wv(w, d[0], res[0], 0)
                self.wv(w, d[0], res[1], 1)
            elif o == "v_fma_f32":
                r = (rvf(w, s[0]).astype(np.float64) * rvf(w, s[1]).astype(np.float64) + rvf(w, s[2]).astype(np.float64)).astype(F32)
                self.wv(w, d[0], r)
            elif o in ("v_add_f32", "v_sub_f32", "v_mul_f32"):
                a, b = rvf(w, s[0]), rvf(w, s[1])
                r = {"v_add_f32": a + b, "v_sub_f32": a - b, "v_mul_f32": a * b}[o]
                self.wv(w, d[0], r.astype(F32))
            elif o == "v_rcp_f32":
                self.wv(w, d[0], (F32(1) / rvf(w, s[0])).astype(F32))
            elif o == "v_cndmask_b32":
                mask = _mask_to_bool(self.rs64(w, s[2]))
                self.wv(w, d[0], np.where(mask, rv(w, s[1]), rv(w, s[0])))
            elif o.startswith("v_cmp_"):
                kind, ty = o[6:8], o[-3:]
                if ty == "f32":
                    a, b = rvf(w, s[0]), rvf(w, s[1])
                else:
                    a, b = rv(w, s[0]).astype(np.int64), rv(w, s[1]).astype(np.int64)
                r = {"eq": a == b, "ne": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[kind] & ex
                self.ws64(w, d[0], int(_bool_to_mask(r)))
            elif o == "v_readfirstlane_b32":
                lanes = np.nonzero(ex)[0]
                self.ws(w, d[0], int(rv(w, s[0])[lanes[0] if len(lanes) else 0]))
            elif o == "v_swap_b32":
                a, b = rv(w, s[0]).copy(), rv(w, s[1]).copy()
                self.wv(w, d[0], a)
                self.wv(w, d[1], b)
            elif o == "v_lshlrev_b32":
                self.wv(w, d[0], (rv(w, s[1]).astype(np.uint64) << np.uint64(self.rs(w, s[0]) & 31)).astype(U32))
            elif o == "v_lshrrev_b32":
                self.wv(w, d[0], rv(w, s[1]) >> U32(self.rs(w, s[0]) & 31))
            elif o in ("v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32"):
                a, b = rv(w, s[0]).astype(np.int64), rv(w, s[1]).astype(np.int64)
                r = {"v_add_u32": a + b, "v_sub_u32": a - b, "v_and_b32": a & b, "v_or_b32": a | b, "v_xor_b32": a ^ b}[o]
                self.wv(w, d[0], (r & 0xffffffff).astype(U32))
            elif o == "v_mul_u32_u24":
                a, b = rv(w, s[0]).astype(np.int64) & 0xffffff, rv(w, s[1]).astype(np.int64) & 0xffffff
                self.wv(w, d[0], ((a * b) & 0xffffffff).astype(U32))
            elif o == "v_lshl_add_u32":   # (src0 << src1) + src2
                a = rv(w, s[0]).astype(np.int64) << (self.rs(w, s[1]) & 31)
                self.wv(w, d[0], ((a + rv(w, s[2]).astype(np.int64)) & 0xffffffff).astype(U32))
            elif o == "v_mad_u32_u24":
                a, b = rv(w, s[0]).astype(np.int64) & 0xffffff, rv(w, s[1]).astype(np.int64) & 0xffffff
                self.wv(w, d[0], ((a * b + rv(w, s[2]).astype(np.int64)) & 0xffffffff).astype(U32))
            else:
                raise EmuError("unknown VALU op " + o)

    def ds(self, w, ins, ex):
        o, d, s, m = ins.op, ins.dst, ins.src, ins.mods
        if not (0 <= m.get("offset", 0) < 65536 and 0 <= m.get("offset0", 0) < 256 and 0 <= m.get("offset1", 0) < 256):
            raise EmuError("DS offset out of range: %r" % (m,))
        addr = self.rv(w, s[0]).astype(np.int64)
        if o in ("ds_read_b128", "ds_read_b64", "ds_read_b32"):
            ndw = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 1}[o]
            idx = self.lds_access(w, addr + m.get("offset", 0), ndw, False, ex, 4 * min(ndw, 4) if ndw < 4 else 16)
            regs = [("v", d[0].i + k) for k in range(ndw)]
            for k in range(ndw):
                self.wv(w, d[0], self.lds[idx[:, k]], k)
            self._pend(w, regs, w.lds_q)
        elif o in ("ds_write_b128", "ds_write_b64", "ds_write_b32"):
            ndw = {"ds_write_b128": 4, "ds_write_b64": 2, "ds_write_b32": 1}[o]
            idx = self.lds_access(w, addr + m.get("offset", 0), ndw, True, ex, 16 if ndw == 4 else 4 * ndw)
            for k in range(ndw):
                self.lds[idx[ex, k]] = self.rv(w, s[1], k)[ex]
            w.lds_q.append([])
        elif o in ("ds_write2_b32", "ds_write2st64_b32"):
            unit = 4 if o == "ds_write2_b32" else 256
            for src, off in ((s[1], m.get("offset0", 0)), (s[2], m.get("offset1", 0))):
                idx = self.lds_access(w, addr + unit * off, 1, True, ex, 4)
                self.lds[idx[ex, 0]] = self.rv(w, src)[ex]
            w.lds_q.append([])
        elif o in ("ds_read2_b32", "ds_read2st64_b32"):
            unit = 4 if o == "ds_read2_b32" else 256
            regs = [("v", d[0].i), ("v", d[0].i + 1)]
            vals = []
            for off in (m.get("offset0", 0), m.get("offset1", 0)):
                idx = self.lds_access(w, addr + unit * off, 1, False, ex, 4)
                vals.append(self.lds[idx[:, 0]])
            for k in range(2):
                self.wv(w, d[0], vals[k], k)
            self._pend(w, regs, w.lds_q)
        else:
            raise EmuError("unknown DS op " + o)

    def run(self, max_instr=50_000_000):
        n = 0
        while True:
            progressed = False
            order = list(self.waves)
            if self.sched_rng is not None:   # adversarial schedule for barrier-free kernels: random order, random stalls
                self.sched_rng.shuffle(order)
                order = [w for w in order if self.sched_rng.random() < 0.5] or order[:1]
            for w in order:
                w.yielded = False
                while not w.done and not w.at_barrier and not w.yielded:
                    if w.pc >= len(self.ins):
                        w.done = True
                        break
                    self.step_wave(w)
                    n += 1
                    progressed = True
                    if n > max_instr:
                        raise EmuError("instruction budget exceeded")
            live = [w for w in self.waves if not w.done]
            if not live:
                break
            if all(w.at_barrier for w in live):
                if len(live) != len(self.waves):
                    raise EmuError("barrier reached by %d of %d waves (others exited)" % (len(live), len(self.waves)))
                for w in live:
                    w.at_barrier = False
                self.epoch += 1
            elif not progressed and self.sched_rng is None:
                raise EmuError("deadlock")

        self.ninstr = n
        return n
