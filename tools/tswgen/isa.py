"""tools/tswgen/isa.py -- a very small gfx950 assembler DSL.

The fused CSPN kernel's main loop is emitted as one inline-asm block (cspn_amd/csrc/cspn2d_tsw_gen.inc) by
tools/tswgen/kernel.py.  This module holds the instruction objects, the text backend, a hazard-aware list scheduler
for straight-line regions and the static checks.  tools/tswgen/emu.py interprets the same objects on the CPU so that
the program logic (register allocation, schedule, addresses, waitcnt placement) is validated against the oracle
before it ever runs on a GPU.  Build-time tooling, not part of the product's run time.
"""


class R(object):
    """register operand: kind 'v' | 's' | 'vcc' | 'exec', first index, count"""
    __slots__ = ("kind", "i", "n", "neg", "absf")

    def __init__(self, kind, i, n=1, neg=False, absf=False):
        self.kind, self.i, self.n, self.neg, self.absf = kind, i, n, neg, absf

    def __getitem__(self, k):
        assert 0 <= k < self.n
        return R(self.kind, self.i + k, 1)

    def sub(self, k, n):
        assert 0 <= k and k + n <= self.n, (self, k, n)
        return R(self.kind, self.i + k, n)

    def __neg__(self):
        return R(self.kind, self.i, self.n, not self.neg, self.absf)

    def abs(self):
        return R(self.kind, self.i, self.n, self.neg, True)

    def regs(self):
        if self.kind in ("vcc", "exec"):
            return [(self.kind, 0), (self.kind, 1)] if self.n == 2 else [(self.kind, self.i)]
        if self.kind == "m0":
            return [("m0", 0)]
        return [(self.kind, self.i + k) for k in range(self.n)]

    def text(self):
        if self.kind in ("vcc", "exec") and self.n == 1:
            t = self.kind + ("_lo", "_hi")[self.i]
        elif self.kind in ("vcc", "exec", "m0"):
            t = self.kind
        elif self.n == 1:
            t = "%s%d" % (self.kind, self.i)
        else:
            t = "%s[%d:%d]" % (self.kind, self.i, self.i + self.n - 1)
        if self.absf:
            t = "|%s|" % t
        if self.neg:
            t = "-" + t
        return t

    def __repr__(self):
        return self.text()


def V(i, n=1):
    return R("v", i, n)


def S(i, n=1):
    return R("s", i, n)


VCC = R("vcc", 0, 2)
EXEC = R("exec", 0, 2)
M0 = R("m0", 0, 1)   # LDS destination base of the LDS-DMA loads (global_load_lds_*)


def optext(o):
    if isinstance(o, R):
        return o.text()
    if isinstance(o, float):
        assert o in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0), o
        return repr(o)
    if isinstance(o, int):
        return str(o) if -16 <= o <= 64 else hex(o & 0xffffffff)
    return str(o)


# opcode classes
VOP_PK = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32"}
VOP_TRANS = {"v_rcp_f32"}
VOP_E32 = {"v_mov_b32", "v_rcp_f32", "v_lshlrev_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32",
           "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mul_u32_u24", "v_xor_b32"}
VOP_E64 = {"v_fma_f32", "v_cndmask_b32", "v_cmp_ge_u32", "v_cmp_lt_u32", "v_cmp_gt_f32", "v_cmp_lt_f32", "v_cmp_eq_u32", "v_cmp_ne_u32",
           "v_mad_u32_u24", "v_lshl_add_u32"}
DS_OPS = {"ds_read_b128", "ds_write_b128", "ds_write2_b32", "ds_read_b64", "ds_write_b64", "ds_write_b32", "ds_read_b32",
          "ds_read2st64_b32", "ds_write2st64_b32", "ds_read2_b32"}
VMEM_LD = {"global_load_dwordx2", "global_load_dword", "global_load_dwordx4"}
# LDS-DMA: no VGPR destination; 64 lanes x 16 bytes land at M0 + 16 * lane.  Operands: [lane offset VGPR, scalar base pair, M0]
VMEM_LDS = {"global_load_lds_dwordx4"}
VMEM_ST = {"global_store_dwordx4", "global_store_dwordx2", "global_store_dword"}
SMEM = {"s_load_dwordx8", "s_load_dwordx4", "s_load_dwordx2", "s_load_dword"}
BRANCH = {"s_cbranch_scc0", "s_cbranch_scc1", "s_branch", "s_cbranch_execz", "s_cbranch_vccnz", "s_cbranch_vccz"}
SCC_WRITERS = {"s_add_u32", "s_addc_u32", "s_subb_u32", "s_add_i32", "s_sub_i32", "s_sub_u32", "s_lshl_b32", "s_lshr_b32", "s_and_b32",
               "s_or_b32", "s_xor_b32", "s_andn2_b32", "s_and_b64", "s_or_b64", "s_andn2_b64", "s_and_saveexec_b64",
               "s_bitcmp1_b32", "s_bitcmp0_b32", "s_ashr_i32", "s_bfe_u32", "s_bfe_i32", "s_min_u32"}
SCC_READERS = {"s_addc_u32", "s_subb_u32", "s_cselect_b32", "s_cselect_b64", "s_cbranch_scc0", "s_cbranch_scc1"}


class I(object):
    """one instruction"""

    def __init__(self, op, dst, src, **mods):
        self.op, self.dst, self.src, self.mods = op, list(dst), list(src), mods
        self.comment = mods.pop("comment", None)

    # --- classification
    def is_valu(self):
        return self.op.startswith("v_")

    def is_dpp(self):
        return "dpp" in self.mods

    def is_fence(self):
        o = self.op
        if o == "pseudo":
            return False
        return (o in BRANCH or o in ("label", "s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_endpgm", "raw") or
                o == "s_and_saveexec_b64" or any(isinstance(d, R) and d.kind == "exec" for d in self.dst))

    def is_mem(self):
        return self.op in DS_OPS or self.op in VMEM_LD or self.op in VMEM_ST or self.op in SMEM or self.op in VMEM_LDS

    # --- register sets (for the scheduler and the checks)
    def reads(self):
        if self.op == "pseudo":
            return list(self.mods["reads"])
        r = []
        for s in self.src:
            if isinstance(s, R):
                r += s.regs()
        o = self.op
        if o in SCC_READERS:
            r.append(("scc", 0))
        if self.is_valu() or self.is_mem() and o not in SMEM:
            r += [("exec", 0), ("exec", 1)]
        if o == "v_mov_b32" and self.is_dpp():
            r += self.dst[0].regs()  # bound_ctrl:1 zero-fills, but keep the old value dependency conservative
        return r

    def __repr__(self):
        return self.text()

    def writes(self):
        if self.op == "pseudo":
            return list(self.mods["writes"])
        w = []
        for d in self.dst:
            if isinstance(d, R):
                w += d.regs()
        o = self.op
        if o in SCC_WRITERS or o.startswith("s_cmp"):
            w.append(("scc", 0))
        if o == "s_and_saveexec_b64":
            w += [("exec", 0), ("exec", 1)]
        return w

    # --- text
    def text(self):
        o, d, s, m = self.op, self.dst, self.src, self.mods
        if o == "label":
            return "%s:" % s[0]
        if o == "pseudo":
            return "\n".join(i.text() for i in m["expand"])
        if o == "raw":
            return s[0]
        if o == "s_waitcnt":
            parts = []
            if "vmcnt" in m:
                parts.append("vmcnt(%d)" % m["vmcnt"])
            if "lgkmcnt" in m:
                parts.append("lgkmcnt(%d)" % m["lgkmcnt"])
            return "s_waitcnt " + " ".join(parts)
        if o in ("s_barrier", "s_endpgm"):
            return o
        if o in ("s_nop", "s_sleep"):
            return "%s %d" % (o, s[0])
        if o in BRANCH:
            return "%s %s" % (o, s[0])
        if o in DS_OPS:
            if o in ("ds_read2st64_b32", "ds_read2_b32"):
                assert 0 <= m.get("offset0", 0) < 256 and 0 <= m.get("offset1", 0) < 256, m
                return "%s %s, %s offset0:%d offset1:%d" % (o, optext(d[0]), optext(s[0]), m.get("offset0", 0), m.get("offset1", 0))
            if o.startswith("ds_read"):
                t = "%s %s, %s" % (o, optext(d[0]), optext(s[0]))
            elif o in ("ds_write2_b32", "ds_write2st64_b32"):
                assert 0 <= m.get("offset0", 0) < 256 and 0 <= m.get("offset1", 0) < 256, m
                return "%s %s, %s, %s offset0:%d offset1:%d" % (o, optext(s[0]), optext(s[1]), optext(s[2]),
                                                                 m.get("offset0", 0), m.get("offset1", 0))
            else:
                t = "%s %s, %s" % (o, optext(s[0]), optext(s[1]))
            if m.get("offset", 0):
                assert 0 <= m["offset"] < 65536, m["offset"]
                t += " offset:%d" % m["offset"]
            return t
        if o in VMEM_LD:
            t = "%s %s, %s, %s" % (o, optext(d[0]), optext(s[0]), optext(s[1]))
            if m.get("offset", 0):
                t += " offset:%d" % m["offset"]
            if m.get("cache"):
                t += " " + m["cache"]
            return t
        if o in VMEM_LDS:
            t = "%s %s, %s" % (o, optext(s[0]), optext(s[1]))
            if m.get("offset", 0):
                t += " offset:%d" % m["offset"]
            if m.get("cache"):
                t += " " + m["cache"]
            return t
        if o in VMEM_ST:
            t = "%s %s, %s, %s" % (o, optext(s[0]), optext(s[1]), optext(s[2]))
            if m.get("offset", 0):
                t += " offset:%d" % m["offset"]
            if m.get("cache"):
                t += " " + m["cache"]
            return t
        if o in SMEM:
            return "%s %s, %s, %s" % (o, optext(d[0]), optext(s[0]), optext(s[1]))
        if o.startswith("s_cmp") or o.startswith("s_bitcmp"):
            return "%s %s, %s" % (o, optext(s[0]), optext(s[1]))
        if o == "v_swap_b32":
            return "v_swap_b32 %s, %s" % (optext(d[0]), optext(d[1]))
        if o == "v_mov_b32" and self.is_dpp():
            return "v_mov_b32_dpp %s, %s %s row_mask:0xf bank_mask:0xf bound_ctrl:1" % (optext(d[0]), optext(s[0]), m["dpp"])
        if o in VOP_PK:
            t = "%s %s, %s" % (o, optext(d[0]), ", ".join(optext(x) for x in s))
            for key in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
                if key in m:
                    t += " %s:[%s]" % (key, ",".join(str(v) for v in m[key]))
            return t
        if o.startswith("v_"):
            name = o
            need64 = o in VOP_E64 or any(isinstance(x, R) and (x.neg or x.absf) for x in s) or m.get("e64", False)
            if o in VOP_E32 and not need64:
                name = o + "_e32"
            elif o in VOP_E32 or o.startswith("v_cmp") or o == "v_cndmask_b32":
                name = o + "_e64"
            return "%s %s, %s" % (name, optext(d[0]), ", ".join(optext(x) for x in s))
        # plain SALU
        if d:
            return "%s %s, %s" % (o, optext(d[0]), ", ".join(optext(x) for x in s))
        return "%s %s" % (o, ", ".join(optext(x) for x in s))


class Prog(object):
    def __init__(self):
        self.ins = []
        self.nlabel = 0

    def emit(self, op, dst=(), src=(), **mods):
        if isinstance(dst, R):
            dst = [dst]
        if isinstance(src, (R, int, float, str)):
            src = [src]
        i = I(op, dst, src, **mods)
        self.ins.append(i)
        return i

    def label(self, name):
        self.emit("label", (), [name])

    def newlabel(self, stem="L"):
        self.nlabel += 1
        return ".L%s_%d_%%=" % (stem, self.nlabel)

    # convenience wrappers ---------------------------------------------------------------
    def waitcnt(self, vm=None, lgkm=None):
        m = {}
        if vm is not None:
            m["vmcnt"] = vm
        if lgkm is not None:
            m["lgkmcnt"] = lgkm
        self.emit("s_waitcnt", **m)

    def text(self):
        return "\n".join(i.text() + ("  ; " + i.comment if i.comment else "") for i in self.ins)


# ------------------------------------------------------------------------------------------- scheduling
def _latency(prod, cons, reg):
    """minimum issue distance between producer and consumer of `reg` (1 = may be adjacent)"""
    if reg[0] == "v" and prod.is_valu() and cons.is_dpp() and cons.op == "v_mov_b32" and reg in cons.src[0].regs():
        return 3  # VALU write -> DPP read: 2 wait states
    if prod.op in VOP_TRANS and cons.is_valu():
        return 2  # trans result -> VALU: 1 wait state (gfx940)
    if prod.is_valu() and reg[0] == "s" and cons.op in VMEM_LD | VMEM_ST | VMEM_LDS:
        return 6  # VALU writes SGPR -> VMEM reads it
    if reg[0] == "m0" and cons.op in VMEM_LDS:
        return 2  # SALU writes M0 -> LDS-DMA uses it: 1 wait state
    return 1


SOFT_VALU_LATENCY = 1
# Issue model measured on gfx950 with two waves per SIMD (profiles/r03_ubench_issue.txt): a scalar instruction placed between a
# wave's vector instructions issues beside the partner wave's VALU work for free (up to one per two VALU instructions), a run of
# scalar instructions costs ~6.5 cycles each.  MIX_POLICY: the list scheduler spreads the non-VALU instructions of a region
# evenly between its VALU instructions instead of leaving them where their (short) dependency chains put them.
MIX_POLICY = False


def schedule_region(region):
    """List-schedule a straight-line region (no fences inside).  Memory operations keep their relative order."""
    n = len(region)
    if n <= 1:
        return list(region)
    preds = [dict() for _ in range(n)]  # j -> min distance
    soft = [dict() for _ in range(n)]   # j -> preferred distance (dependent VALU results are better not consumed next slot)
    last_write, readers = {}, {}
    last_mem = {}
    for j, ins in enumerate(region):
        rd, wr = ins.reads(), ins.writes()
        for r in rd:
            if r in last_write:
                i = last_write[r]
                preds[j][i] = max(preds[j].get(i, 0), _latency(region[i], ins, r))
                if SOFT_VALU_LATENCY > 1 and r[0] == "v" and region[i].is_valu() and ins.is_valu():
                    soft[j][i] = SOFT_VALU_LATENCY
        for r in wr:
            if r in last_write:
                i = last_write[r]
                preds[j][i] = max(preds[j].get(i, 0), 1)
            for i in readers.get(r, ()):
                if i != j:
                    lat = 2 if region[i].op in VMEM_ST and r[0] == "v" else 1  # store data: leave a gap before overwriting
                    # WAR: the reader must issue first (distance >= 1, same slot impossible anyway)
                    preds[j][i] = max(preds[j].get(i, 0), lat)
        if ins.is_mem():  # LDS and vector-memory operations each keep their own program order (separate counters)
            ch = "ds" if ins.op in DS_OPS else "vm"
            if last_mem.get(ch) is not None:
                preds[j][last_mem[ch]] = max(preds[j].get(last_mem[ch], 0), 1)
            last_mem[ch] = j
        for r in wr:
            last_write[r] = j
            readers[r] = []
        for r in rd:
            readers.setdefault(r, []).append(j)
    succs = [[] for _ in range(n)]
    for j in range(n):
        for i, d in preds[j].items():
            succs[i].append((j, d))
    # priority: longest latency-weighted path to the end
    prio = [0] * n
    for i in range(n - 1, -1, -1):
        prio[i] = 1 + max([d - 1 + prio[j] for j, d in succs[i]] + [0])
    out, slot_of, done = [], {}, [False] * n
    npred = [len(preds[j]) for j in range(n)]
    ready = [j for j in range(n) if npred[j] == 0]
    slot = 0
    is_v = [ins.is_valu() for ins in region]
    n_v = sum(is_v)
    n_o = n - n_v
    ratio = max(1, n_v // max(1, n_o)) if MIX_POLICY else 0   # VALU instructions between two others
    run_v = 0
    while len(slot_of) < n:
        best = None
        bestv = besto = None
        for j in ready:
            ok = all(slot - slot_of[i] >= d for i, d in preds[j].items())
            if not ok:
                continue
            # placement hint: an instruction nobody in the region waits for (publishing store, prefetch) is taken as soon
            # as `at` (fraction of the region) has been issued; otherwise the critical path decides
            at = region[j].mods.get("at")
            relaxed = all(slot - slot_of[i] >= d for i, d in soft[j].items())
            key = (1 if at is not None and slot >= at * n else 0, 1 if relaxed else 0, prio[j] if at is None else 0, -j)
            if best is None or key > bestkey:
                best, bestkey = j, key
            if MIX_POLICY and not (at is not None and slot < at * n):
                if is_v[j]:
                    if bestv is None or key > bestvkey:
                        bestv, bestvkey = j, key
                elif besto is None or key > bestokey:
                    besto, bestokey = j, key
        if MIX_POLICY and bestv is not None and besto is not None:
            # both kinds ready: a non-VALU instruction goes when it is due (every `ratio` VALU instructions) or was asked for
            # at this point of the region
            if bestokey[0] or run_v >= ratio:
                best = besto
            else:
                best = bestv
        if best is None:
            out.append(I("s_nop", (), [0]))
            slot += 1
            continue
        ready.remove(best)
        slot_of[best] = slot
        out.append(region[best])
        run_v = run_v + 1 if is_v[best] else 0
        slot += 1
        for j, _ in succs[best]:
            npred[j] -= 1
            if npred[j] == 0:
                ready.append(j)
    return out


def expand_pseudos(prog):
    out = []
    for ins in prog.ins:
        if ins.op == "pseudo":
            out += ins.mods["expand"]
        else:
            out.append(ins)
    prog.ins = out
    return prog


def schedule(prog):
    """Schedule every fence-free region of the program in place."""
    out, region = [], []
    for ins in prog.ins:
        if ins.is_fence() or ins.mods.get("pin", False):
            out += schedule_region(region)
            region = []
            out.append(ins)
        else:
            region.append(ins)
    out += schedule_region(region)
    prog.ins = out
    return prog


def check_hazards(prog):
    """Static re-check of the issue-distance rules over the final instruction order (across fences too)."""
    errs = []
    hist = []  # recent instructions with their slot number
    slot = 0
    for ins in prog.ins:
        if ins.op == "label":
            continue  # fall-through distance; a taken branch only adds cycles
        width = ins.src[0] + 1 if ins.op == "s_nop" else 1
        for ps, p in hist:
            for r in ins.reads():
                if r in p.writes():
                    need = _latency(p, ins, r)
                    if slot - ps < need:
                        errs.append("hazard: %s -> %s (distance %d < %d)" % (p.text(), ins.text(), slot - ps, need))
        hist.append((slot, ins))
        hist = [(s, p) for s, p in hist if slot - s < 8]
        slot += width
    return errs
