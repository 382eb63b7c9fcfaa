"""tools/tsw3_trace.py -- where does a step of the round-3 loop spend its cycles?  (run on the GPU box)

    TSW_CFG="dict(trace=True)" bash tools/build_abl3.sh trace ""
    CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_trace.so python tools/tsw3_trace.py [out.json]
Every wave records six s_memtime stamps per step (shader cycles):
  t0 step start | t2 boundary rows arrived | t1 (DMA steps) this wave's LDS-DMA of three steps ago has landed
  t3 chain finished, everything issued | t4 own LDS traffic drained | t5 barrier released
Reports mean cycles per phase by kind of step of a WAVE (step number mod 3: 0 DMA issue, 1 raw reads, 2 arithmetic + writes;
with / without a slot event), and per STEP of the workgroup: its duration and how long the last wave to arrive at the barrier
was busy -- BASELINE config 3 (KITTI 304x1216 x 64, 24 iterations)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd  # noqa: E402

WG_BYTES, NW, REC = 1024 * 8 * 32, 8, 8   # per workgroup; waves; dwords per (wave, step)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tsw3_trace.json"
    lib = cspn_amd.load()
    B, H, W = 64, 304, 1216
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    n_wg = 256
    buf = torch.zeros(n_wg * WG_BYTES // 4, dtype=torch.int32, device="cuda")
    rc = lib.cspn_debug_tsw3_set_trace(ctypes.c_void_p(buf.data_ptr()))
    assert rc == 0, rc
    for _ in range(200):   # clocks
        cspn_amd.cspn2d_forward(g, h, None, 24, "8sum", "fused")
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cspn_amd.cspn2d_forward(g, h, None, 24, "8sum", "fused")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    raw = buf.cpu().numpy().view(np.uint32).reshape(n_wg, 1024, NW, REC)
    phases = ["top wait", "dma wait", "chain+issue", "drain", "barrier"]
    order = [0, 2, 1, 3, 4, 5]   # stamps in time order
    kinds, steps = {}, {0: [], 1: [], 2: []}
    busy_last = {0: [], 1: [], 2: []}
    for wg in range(0, n_wg, 5):
        r = raw[wg]
        nsteps = int((r[:, 0, 0] != 0).sum())
        if nsteps < 50:
            continue
        t = r[20:nsteps - 30].astype(np.int64)            # steady state: skip ring fill / drain
        c = t[:, :, 6]
        ts = t[:, :, order]
        d = (ts[:, :, 1:] - ts[:, :, :-1]) & 0xffffffff      # five phases
        for ph in (0, 1, 2):
            for ev in (0, 1):
                sel = (c % 3 == ph) & ((c < 4) == bool(ev))
                if sel.any():
                    kinds.setdefault("ph%d%s" % (ph, "+event" if ev else ""), []).append(d[sel].mean(0))
        # per step of the workgroup: start = latest barrier release of the step before, end = latest release of this step
        rel = t[:, :, 5]
        base = rel[:-1].min(1)
        dur = ((rel[1:].max(1) - rel[:-1].max(1)) & 0xffffffff)
        arrive = (t[1:, :, 4] - t[1:, :, 0]) & 0xffffffff   # busy time of each wave up to its barrier arrival
        ph_of = (c[1:, 0] % 3)
        for ph in (0, 1, 2):
            steps[ph].append(dur[ph_of == ph].mean())
            busy_last[ph].append(arrive[ph_of == ph].max(1).mean())
    res = {"workload": "KITTI 304x1216 x 64, 24 iterations", "forward_ms_instrumented": round(ms, 4), "phases": phases, "kinds": {},
           "step_cycles_by_phase": {str(k): round(float(np.mean(v)), 1) for k, v in steps.items()},
           "busiest_wave_cycles_by_phase": {str(k): round(float(np.mean(v)), 1) for k, v in busy_last.items()}}
    # one workgroup, 24 consecutive steady-state steps, every wave: counter, stamps relative to the step's first wave start
    r = raw[40]
    ns = int((r[:, 0, 0] != 0).sum())
    t = r[100:124].astype(np.int64)
    detail = []
    for st in range(t.shape[0]):
        t0 = t[st, :, 0].min()
        detail.append([[int(t[st, w, 6])] + [int((t[st, w, k] - t0) & 0xffffffff) for k in order] for w in range(NW)])
    res["detail_wg40_steps100_123"] = detail
    for k in sorted(kinds):
        m = np.mean(kinds[k], 0)
        res["kinds"][k] = {"cycles": [round(float(x), 1) for x in m], "total": round(float(m.sum()), 1)}
    print(json.dumps(res, indent=1))
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
