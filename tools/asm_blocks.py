#!/usr/bin/env python
"""tools/asm_blocks.py -- static instruction census of one kernel in hipcc's -S output.
Splits the kernel into basic blocks and prints per block: #VALU (pk_fma / dpp / other), #SALU, #LDS, #VMEM,
#waitcnt, #branches, and the successor labels.  Used to see what the compiler made of the fused step.
usage: asm_blocks.py fused.s <kernel-substring> [min_instrs]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.split(":")[0].endswith("E"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], {"name": "entry", "ins": [], "line": start}
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].rstrip()
    if not l.strip():
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "ins": [], "line": i}
        continue
    if l.startswith("\t") and not l.strip().startswith("."):
        cur["ins"].append(l.strip())
blocks.append(cur)
def cls(op):
    if op.startswith("v_pk_fma") or op.startswith("v_pk_mul") or op.startswith("v_pk_add"): return "pk"
    if op.startswith("v_"): return "v"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "s"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
tot = {}
for b in blocks:
    c = {}
    dpp = 0
    for ins in b["ins"]:
        op = ins.split()[0]
        k = cls(op)
        c[k] = c.get(k, 0) + 1
        if "dpp" in ins or "row_" in ins or "wave_sh" in ins: dpp += 1
    n = sum(c.values())
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
    if n < minn: continue
    succ = [ins.split()[-1] for ins in b["ins"] if ins.startswith("s_cbranch") or ins.startswith("s_branch")]
    print("%-12s L%-6d n=%4d pk=%3d v=%3d(dpp %2d) s=%3d lds=%2d vmem=%2d smem=%d wait=%2d br=%d bar=%d -> %s" % (
        b["name"], b["line"] + 1, n, c.get("pk", 0), c.get("v", 0), dpp, c.get("s", 0), c.get("lds", 0), c.get("vmem", 0),
        c.get("smem", 0), c.get("wait", 0), c.get("br", 0), c.get("bar", 0), ",".join(succ)))
print("total", tot)
