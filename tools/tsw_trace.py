"""tools/tsw_trace.py -- where does a step of the fused 2D loop spend its cycles?  (run on the GPU box)

Needs a timing-instrumented single-variant build of the library:
    bash tools/r04/build_trace.sh
    CSPN_AMD_LIB=$PWD/cspn_amd/abl/trace/libcspn_amd.so python tools/tsw_trace.py [out.json]
Every wave records six s_memtime stamps per step (shader cycles; the instrumented loop is ~4 % slower than the product loop):
  t0 step start | t1 cooking done (= t0 in steps without cooking) | t2 boundary rows + ring reads arrived
  t3 chain finished, everything issued | t4 own LDS writes landed | t5 barrier released
Reports mean cycles per phase for the four kinds of step of a wave: plain, event (retire + inject), cooking,
cooking + event, and the share of wall time of each kind -- BASELINE config 3 (KITTI 304x1216 x 64, 24 iterations)."""
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
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tsw_trace.json"
    lib = cspn_amd.load()
    B, H, W = 64, 304, 1216
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    n_wg = 256
    buf = torch.zeros(n_wg * WG_BYTES // 4, dtype=torch.int32, device="cuda")
    rc = lib.cspn_debug_tsw_set_trace(ctypes.c_void_p(buf.data_ptr()))
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
    phases = ["cook", "tail+LDS wait", "chain+issue", "write drain", "barrier"]
    kinds = {"plain": [], "event": [], "cook": [], "cook+event": []}
    step_cycles = []
    for wg in range(0, n_wg, 5):
        r = raw[wg]
        nsteps = int((r[:, 0, 0] != 0).sum())
        if nsteps < 50:
            continue
        t = r[20:nsteps - 30].astype(np.int64)            # steady state: skip ring fill / drain
        c = t[:, :, 6]
        d = (t[:, :, 1:6] - t[:, :, 0:5]) & 0xffffffff      # five phases
        full = (t[1:, :, 0] - t[:-1, :, 0]) & 0xffffffff    # step start to next step start (includes the trace flush)
        step_cycles.append(full.mean())
        for name, sel in (("plain", (c % 3 != 2) & (c >= 4)), ("event", (c % 3 != 2) & (c < 4)),
                          ("cook", (c % 3 == 2) & (c >= 4)), ("cook+event", (c % 3 == 2) & (c < 4))):
            if sel.any():
                kinds[name].append(d[sel].mean(0))
    # per STEP of a workgroup (one barrier per step): duration = latest barrier release to latest barrier release; how long the
    # last wave to arrive had been busy; by the kind of step (counter % 3 == 2: every wave cooks)
    dur_by, busy_by, nsteps_all = {"cook": [], "other": []}, {"cook": [], "other": []}, []
    for wg in range(0, n_wg, 5):
        r = raw[wg]
        nsteps = int((r[:, 0, 0] != 0).sum())
        nsteps_all.append(nsteps)
        if nsteps < 50:
            continue
        t = r[20:nsteps - 30].astype(np.int64)
        rel = t[:, :, 5]
        dur = (rel[1:].max(1) - rel[:-1].max(1)) & 0xffffffff
        arrive = ((t[1:, :, 4] - t[1:, :, 0]) & 0xffffffff).max(1)
        is_cook = (t[1:, 0, 6] % 3) == 2
        for name, sel in (("cook", is_cook), ("other", ~is_cook)):
            dur_by[name].append(dur[sel].mean())
            busy_by[name].append(arrive[sel].mean())
    res = {"workload": "KITTI 304x1216 x 64, 24 iterations", "forward_ms_instrumented": round(ms, 4),
           "steps_recorded_per_workgroup": [int(min(nsteps_all)), int(max(nsteps_all))],
           "mean_cycles_per_step_incl_flush": round(float(np.mean(step_cycles)), 1),
           "step_cycles": {k: round(float(np.mean(v)), 1) for k, v in dur_by.items()},
           "busiest_wave_cycles": {k: round(float(np.mean(v)), 1) for k, v in busy_by.items()},
           "phases": phases, "kinds": {}}
    for k, v in kinds.items():
        m = np.mean(v, 0)
        res["kinds"][k] = {"cycles": [round(float(x), 1) for x in m], "total": round(float(m.sum()), 1)}
    print(json.dumps(res, indent=1))
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
