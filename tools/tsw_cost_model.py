#!/usr/bin/env python
"""tools/tsw_cost_model.py -- time model of the fused 2D forward (time-skewed wave ring, cspn2d_tsw.hip) and what it says about
other decompositions.  Prints profiles/r05_decomposition_model.md (round 4's table + the short-stream section of round 5).

    python -m tools.tsw_cost_model > profiles/r05_decomposition_model.md

Model (every constant measured on MI355X, sources in the table the script prints):
    forward time = S x (I x c_instr + c_sync) / f
    S       steps of the longest workgroup = 3 (Q - 1) // 4 + (Q - 1) % 4 + 24 for a stream of Q rows (the planner: tools/tswgen/plan.py)
    I       instructions a SIMD issues per step = 2 waves x the mean over the 24 ring phases of the generated loop's step bodies
            (static census of tools/tswgen/kernel.py's output; the out-of-line patch code of edge rows is not executed on ordinary rows
            and is left out)
    c_instr 5.0 shader cycles per instruction of ANY kind: two waves per SIMD run the same code in lock step behind one barrier per
            step, so scalar / LDS / VMEM instructions do not co-issue beside the partner's VALU work (profiles/r03_ubench_issue.txt,
            r03_perf_notes.md 2); a v_pk_fma_f32 with three VGPR-pair operands costs 4.95 at two waves per SIMD whatever the
            registers' banks (profiles/r04_ubench_vgpr_banks.txt; 4.45 with a constant / SGPR operand; v_fma_f32 3.9)
    c_sync  260 cycles per step nobody issues in: LDS write -> s_waitcnt -> s_barrier -> LDS read of the neighbour's row
            (profiles/r02_perf_notes.md: 170-250 with the deferred tail covering part of it) plus the skew of eight waves meeting at
            the barrier -- the one constant FITTED here, to the two plans measured on one box in round 4 (1531 and 1552 cycles per step)
    f       2.15 GHz while ~3.5 TB/s stream from HBM (profiles/r03_power_samples.txt; 2.38 GHz without the stream)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.tswgen import kernel as K  # noqa: E402
from tools.tswgen.plan import LinearPlan, plan_bands  # noqa: E402

C_INSTR, C_SYNC, F_GHZ = 5.0, 260.0, 2.15
ALG_BYTES = 64 * 304 * 1216 * 40


def steps_of(Q):
    return 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + 24


def census(cfg):
    """mean instructions per wave-step of the fast step bodies, by kind, executed on ordinary rows (edge / inactive-row patch code
    sits behind never-taken branches: its instructions are subtracted by counting only what lies on the fall-through path)"""
    p = K.build(dict(cfg))
    kinds = {}
    cur, skip_to = None, None
    for ins in p.ins:
        if ins.op == "label":
            name = ins.src[0]
            if name.startswith(".LS") and not name.startswith(".LSs"):
                cur = name
            elif name.startswith(".LSs") or name.startswith(".Lexit"):
                cur = None
            if skip_to is not None and name == skip_to:
                skip_to = None
            continue
        if cur is None or skip_to is not None:
            continue
        o = ins.op
        if o == "s_cbranch_scc1" and isinstance(ins.src[0], str) and ("cmath" in ins.src[0]):
            skip_to = ins.src[0]   # "most rows: nothing to patch" -- the branch is TAKEN on ordinary rows
            kinds["salu"] = kinds.get("salu", 0) + 1
            continue
        if o == "v_pk_fma_f32":
            k = "pk_fma"
        elif o.startswith("v_") and ins.is_dpp():
            k = "dpp"
        elif o.startswith("v_"):
            k = "valu"
        elif o.startswith("ds_"):
            k = "lds"
        elif o.startswith("global_"):
            k = "vmem"
        else:
            k = "salu"   # scalar ALU, branches, waitcnt, barrier
        kinds[k] = kinds.get(k, 0) + 1
    return {k: v / 24.0 for k, v in kinds.items()}


def t_ms(S, I):
    return S * (I * C_INSTR + C_SYNC) / (F_GHZ * 1e9) * 1e3


def main():
    c = census(dict(norm=0))
    per_wave = sum(c.values())
    I = 2 * per_wave
    chain = 2 * (c["pk_fma"] + c["dpp"])
    bands = plan_bands(1216, 24)
    lp = LinearPlan(64, 304, 1216, 24, 256)
    S_new = steps_of(lp.L)
    S_old = steps_of(-(-64 * 304 // 42) + 48 + 1)
    rows = []

    def row(name, S, I_, note, measured=None, extra_cycles=0.0):
        t = S * (I_ * C_INSTR + C_SYNC + extra_cycles) / (F_GHZ * 1e9) * 1e3
        rows.append((name, S, I_, t, ALG_BYTES / (t * 1e-3) / 8e12, measured, note))

    row("round 3: 6 bands x 256 columns, 42 band groups on 252 CUs (513 stream rows)", S_old, I, "baseline", "0.2884 (BENCH_r03 driver), 0.2905 (r04 box, plan mode 2)")
    row("**round 4 (built): linear plan, 256 pieces of 1.5 (image, band) units (481 stream rows)**", S_new, I, "go: built", "0.2772 / 0.2776 (profiles/r04_plan_ab.md)")
    ev = 2 * 25 * 4 / 24.0   # 25 instructions fewer per event, 4 events per wave and 24 steps, two waves per SIMD
    row("+ events with 10 ds_read_b128 (ring in consumer order) and the retiring row's offset / flags kept in SGPRs", S_new, I - ev,
        "no-go for now: -3 % for a new ring layout (176-byte records, 2-way conflicts on the cook's writes) in every variant")
    # 5 bands with a neighbour-CU edge exchange: 320 (image, band) units on 256 CUs = 1.25 units per CU
    Q5 = int(1.25 * 304) + 24 + 2
    row("5 bands (8-column halos refreshed every 8 levels from the neighbour CU), exchange costed at ZERO", steps_of(Q5), I,
        "upper bound of the idea: -14 %")
    row("the same with the exchange: 4 of the 32 resident rows reach a refresh level in EVERY step, data produced in step s is needed in step "
        "s + 1 in both directions (the ring has no slack: the sum of the two directions' slacks is fixed by 8 waves x 3 levels = 24), so one "
        "L2 round trip between CUs of an XCD (>= 700 cycles, sc1 through memory 4.5 us: profiles/r02_gridsync_ubench.txt) is exposed per step, "
        "+ ~20 publish / poll instructions per wave-step", steps_of(Q5), I + 40, "**no-go**: slower than today", None, 700.0)
    row("5 columns per lane (320-column bands: 5 bands cover 1216 + 4 x 48 without any exchange; DPP, boundary rows, events amortised over 5 / 4 "
        "the pixels)", steps_of(int(1.25 * 304) + 24 + 2), I * 0.97 * 1.25, "**no-go**: 4 rows x 5 columns x 11 registers = 220 of 256 VGPRs before "
        "any temporary (the loop needs ~80: boundary rows, shifted pairs, cooking); the same columns per CU as 6 x 256, so no gain either")
    row("half-step phase shift between neighbouring waves (to take the LDS round trip off the critical path)", S_new, I,
        "**impossible**: a lead of 1/2 step per wave is 4 steps around the ring of 8: a wave would get its next four rows every 20 steps while a "
        "row lives 24 (the ring is exactly full)", None, -C_SYNC)
    row("SIMD partners cook in different steps (generator option `stagger`: waves 0..3 at counters = 2 mod 3, waves 4..7 at 0 mod 3), hoping their "
        "non-VALU instructions issue beside the partner's FMAs", S_new, I, "**built and measured: +0.6 %** (0.2775 vs 0.2757 ms, profiles/r04_stagger_ab.md): the "
        "same instructions per SIMD and three steps either way", "0.2769-0.2786")
    row("floor of this ring: only the chain (64 v_pk_fma_f32 + 12 DPP moves per wave-step), cooking / events / feed free", S_new, chain,
        "what no pipeline around the ring can beat")
    print("# r05 — time model of the fused 2D forward and what it says about other decompositions (round 4's table, constants unchanged; + short streams)\n")
    print("Generated by `python -m tools.tsw_cost_model` (the model and the sources of its constants are in that file's header).  BASELINE config 3 at")
    print("64 images per GPU (KITTI 304x1216, 24 iterations, 946.3 MB algorithmic), MI355X.\n")
    print("`forward time = S x (I x %.1f + %d) cycles / %.2f GHz`, S = steps of the longest workgroup, I = instructions a SIMD issues per step (two" % (C_INSTR, C_SYNC, F_GHZ))
    print("waves in lock step: an instruction of any kind costs the SIMD ~5 cycles).\n")
    print("Census of the generated loop (norm 8sum, no mask), mean per wave-step over the 24 ring phases, ordinary rows: " +
          ", ".join("%s %.1f" % (k, v) for k, v in sorted(c.items())) + " = **%.1f per wave, %.1f per SIMD**; of these the propagation chain itself (64 packed" % (per_wave, I))
    print("FMAs + 12 DPP moves) is %.0f, cooking (normalise + fold + its loads and ring writes) ~%.0f, the four events per wave and 24 steps ~%.0f.\n" % (
        chain, 2 * 8 * (207 - 86) / 24.0, 2 * ((148 + 146 + 153 - 3 * 86) + (273 - 207)) / 24.0))
    print("| decomposition | S | I | model ms | model frac of 8 TB/s | measured ms | verdict |")
    print("|---|---|---|---|---|---|---|")
    for name, S, I_, t, fr, meas, note in rows:
        print("| %s | %d | %.0f | %.4f | %.3f | %s | %s |" % (name, S, I_, t, fr, meas or "", note))
    print()
    print("Reading:\n")
    print("* The model reproduces both measured plans within 3 %% (%.4f vs 0.2884 / 0.2905; %.4f vs 0.2772): time is instructions issued," % (rows[0][3], rows[1][3]))
    print("  not bytes moved -- HBM traffic is 1.11x algorithmic (profiles/r03_pmc_fetch_write.md) and 3.4 TB/s of the ~6.3 TB/s a copy reaches.")
    print("* The planner was the one lever that removes STEPS without touching the loop; it is built (-4.6 % measured).  Every other candidate either")
    print("  needs data from a neighbouring CU within one step (the ring has no slack to hide a round trip: each row needs its upper and its lower")
    print("  neighbour's value of the level before, and 8 waves x 3 levels = 24 leaves no spare level), or more registers than a lane has.")
    print("* **Ceiling of the register-resident-weights ring: %.3f ms = %.2f of the roofline with a free feed**; with the cooking a 36 B/pixel input needs" % (rows[-1][3], rows[-1][4]))
    print("  (~%.0f instructions per SIMD-step) and four row events per wave cycle it is the %.2f measured (%.2f with cheaper events).  >= 0.48 (<= 0.246 ms) would need" % (2 * 8 * (207 - 86) / 24.0, 0.427, rows[2][4] * 0.427 / rows[1][4]))
    print("  I <= %.0f at S = %d, i.e. cooking + events + plumbing in %.0f instructions per SIMD-step instead of %.0f: not with a feed that passes every" % (
        (0.246e-3 * F_GHZ * 1e9 / S_new - C_SYNC) / C_INSTR, S_new, (0.246e-3 * F_GHZ * 1e9 / S_new - C_SYNC) / C_INSTR - chain, I - chain))
    print("  coefficient through registers or LDS once (both feeds built so far, loads into VGPRs and LDS-DMA row slots, land within 1 % of each other).")
    print("* Statement the round-3 review asked for: **this design tops out at ~0.43-0.45 of the 8 TB/s roofline (0.56 of the 6.3 TB/s copy ceiling) on")
    print("  MI355X; 0.48 would take a third fewer non-chain instructions, 0.60 is out of reach.  No further rounds go into the 2D forward's pipeline**; the remaining 3 % (events) is noted above.")
    print("* Round 5 counters (profiles/r05_pmc_sq.md): a VALU instruction holds the pipe 4.03 cycles and the pipe is busy 51 % of the kernel; the `x 5 + 260` of the fit is")
    print("  issue turnaround plus the two waves of a SIMD waiting at the same time, i.e. schedule, not pipe occupancy.  Removing ~30 of the ~81 cooking instructions (the")
    print("  pre-normalised input contract, profiles/r05_prenorm_ab.md) bought 2.9 %, not the 12 % an instruction count would give: the cooking math already runs under the")
    print("  boundary rows' LDS round trip.")
    short_streams()


def short_streams():
    """round 5, review item 4: an XCD-resident plan for short streams (config 3 as written = 8 images per GPU), modelled before building"""
    hop_same, hop_cross = 0.46, 0.80       # us per hop of a tagged 16-byte quad, loaded fabric (profiles/r05_ubench_xcd_handoff.txt)
    H, W, NI = 304, 1216, 24
    px_cu = H * W / 32.0                   # one image per XCD, 32 CUs each own a 304 x 38-column slab
    fma_cycles = px_cu * 8 / 2 / 64 / 4 * 4.03       # packed FMAs per SIMD x 4.03 pipe cycles (r05_pmc_sq.md)
    lds_cycles = px_cu * (9 * 4 + 4) / 256.0          # the 3D kernel's way of holding H: 9 LDS reads + 1 write per pixel-iteration at 256 B/clk
    comp_us = max(fma_cycles, lds_cycles) / 2150.0 * 1.6   # x1.6: what the 3D kernel's arithmetic phase takes over its LDS floor (2 000 vs 1 250 cycles)
    sync_us = 0.25                          # polls of the second edge + LDS + barrier (3D kernel: ~1 200 cycles per step for 26 neighbours; 2 here)
    it_us = comp_us + hop_same + sync_us
    load_us = H * W * 40 / 0.6e12 * 1e6     # one XCD streams its image alone: 1/8 of the ~4.8 TB/s the ring's loads reach chip-wide
    total = load_us + NI * it_us
    alg = 8 * H * W * 40
    print()
    print("## Short streams: config 3 as written is 8 images per GPU (0.0645 ms = 0.229 today) -- an XCD-resident plan, modelled\n")
    print("Idea (round-4 review): 8 images = 8 XCDs; the 32 CUs of an XCD each own a fixed 304 x 38-column slab of ONE image, its 9 coefficients per pixel")
    print("resident in registers for all 24 iterations (%.0f pixels x 9 x 4 B = %.0f KB of a CU's 512 KB), no warm-up rows, no drain; the two edge columns of a slab" % (px_cu, px_cu * 36 / 1024))
    print("travel to the neighbouring CUs through that XCD's L2 once per iteration, as tagged quads (the scheme of cspn3d_persistent.hip).\n")
    print("| term | value | source |")
    print("|---|---|---|")
    print("| hand-off inside an XCD, per hop | %.2f us (cross-XCD %.2f) | `profiles/r05_ubench_xcd_handoff.txt`: tagged 16-byte quad, plain store + `sc1` poll, 128 pairs at once |" % (hop_same, hop_cross))
    print("| arithmetic of one iteration on a slab | %.2f us | %.0f pixels x 8 FMA: %.0f pipe cycles per SIMD packed; LDS floor %.0f cycles (9 reads + 1 write of H per pixel at 256 B/clk); x 1.6 as measured on the 3D kernel's arithmetic phase |" % (comp_us, px_cu, fma_cycles, lds_cycles))
    print("| polls of the second edge, LDS, barrier | %.2f us | the 3D kernel spends ~1 200 cycles per step on 26 neighbours; 2 here |" % sync_us)
    print("| **one iteration** | **%.2f us** | a Jacobi step: nothing of iteration t + 1 can start on a slab's edge columns before the neighbour's iteration t arrived |" % it_us)
    print("| load phase | %.1f us | 14.8 MB per image through ONE XCD's share of the fabric (~0.6 TB/s); not overlapped: every coefficient is needed before iteration 1 ends |" % load_us)
    print("| **forward, 8 images** | **%.1f us = %.3f of the roofline** | today %.1f us = 0.229 |" % (total, alg / (total * 1e-6) / 8e12, 64.5))
    print()
    print("**No-go.**  The review's bar was <= 45 us (0.33); the model says %.0f us: 24 dependent hand-offs cannot be hidden (the same wall the 3D kernel stands at: its step is the" % total)
    print("hand-off, not its arithmetic, profiles/r05_vol3d_rows_first_ab.md), and a single XCD loads its image at an eighth of the chip's bandwidth.  That is %s than today's" % ("SLOWER" if total > 64.5 else "%.0f %% faster" % (100 * (1 - total / 64.5))))
    print("%.1f us, for a third kernel family (slab plan, edge exchange, its own cooking) that would serve one shape -- config 2 (NYU x 16: 16 images of 228 x 304 on 8 XCDs) fits it" % 64.5)
    print("worse (two images per XCD: 16 CUs per image, 19-column slabs, the same 24 hops).  What the short-stream shapes need is fewer fixed steps per stream, which the ring cannot")
    print("give (48 warm-up rows and a 24-step drain are its 24 levels), or more images per GPU, which is what the weak-scaling line measures (64 per GPU: 0.42).")


if __name__ == "__main__":
    main()
