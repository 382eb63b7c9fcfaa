"""tools/r06/tsw4_trace.py -- where does a step of the 12 x 3 ring spend its cycles?  (run on the GPU box)
    tools/r06/build_abl4.sh trace "" "dict(trace=True)"
    CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_t4_trace.so python tools/r06/tsw4_trace.py [out.json]
Every wave records four s_memtime stamps per step (shader cycles): t0 step start (= barrier released) | t1 boundary rows / raw
reads arrived | t2 chain + role work issued | t3 own LDS writes (and the DMA waited for) landed -> barrier.
Reports, per ring counter c: mean cycles of the three phases and of the wait at the barrier, how often the wave with that counter is
the LAST to arrive at the barrier, and the step duration by step parity."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402

NW, REC, MAXSTEPS = 12, 8, 1024
WG_BYTES = MAXSTEPS * NW * REC * 4


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tsw4_trace.json"
    lib = cspn_amd.load()
    B, H, W = 64, 304, 1216
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    n_wg = 256
    buf = torch.zeros(n_wg * WG_BYTES // 4, dtype=torch.int32, device="cuda")
    rc = lib.cspn_debug_tsw4_set_trace(ctypes.c_void_p(buf.data_ptr()))
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
    raw = buf.cpu().numpy().view(np.uint32).reshape(n_wg, MAXSTEPS, NW, REC)
    per_c = {c: {"top": [], "chain": [], "drain": [], "barrier": [], "last": 0, "n": 0} for c in range(24)}
    dur = {0: [], 1: []}
    nsteps_all = []
    for wg in range(0, n_wg, 5):
        r = raw[wg].astype(np.int64)
        nsteps = int((r[:, 0, 0] != 0).sum())
        nsteps_all.append(nsteps)
        if nsteps < 100:
            continue
        t = r[30:nsteps - 40]              # steady state: skip ring fill / drain
        c = t[:, :, 4]
        t0, t1, t2, t3 = (t[:, :, k] for k in range(4))
        nxt = t0[1:]
        step = (nxt.max(1) - t0[:-1].max(1)) & 0xffffffff
        par = c[:-1, 0] & 1
        for p in (0, 1):
            dur[p] += list(step[par == p])
        last = np.argmax((t3[:-1] - t0[:-1].min(1, keepdims=True)) & 0xffffffff, axis=1)
        for s in range(t.shape[0] - 1):
            for w in range(NW):
                d = per_c[int(c[s, w])]
                d["top"].append((t1[s, w] - t0[s, w]) & 0xffffffff)
                d["chain"].append((t2[s, w] - t1[s, w]) & 0xffffffff)
                d["drain"].append((t3[s, w] - t2[s, w]) & 0xffffffff)
                d["barrier"].append((nxt[s, w] - t3[s, w]) & 0xffffffff)
                d["n"] += 1
            per_c[int(c[s, last[s]])]["last"] += 1
    res = {"workload": "KITTI 304x1216 x 64, 24 iterations", "forward_ms_instrumented": round(ms, 4),
           "steps_recorded_per_workgroup": [int(min(nsteps_all)), int(max(nsteps_all))],
           "step_cycles_even": round(float(np.mean(dur[0])), 1), "step_cycles_odd": round(float(np.mean(dur[1])), 1),
           "per_counter": {}}
    print("forward (instrumented) %.4f ms; step cycles even %.0f odd %.0f" % (ms, res["step_cycles_even"], res["step_cycles_odd"]))
    print(" c    top  chain  drain  barrier  busy   last-to-arrive share")
    for c in range(24):
        d = per_c[c]
        if not d["n"]:
            continue
        row = {k: round(float(np.mean(d[k])), 1) for k in ("top", "chain", "drain", "barrier")}
        row["busy"] = round(row["top"] + row["chain"] + row["drain"], 1)
        row["last_share"] = round(d["last"] * 12.0 / d["n"], 3)   # of the steps in which a wave holds this counter
        res["per_counter"][c] = row
        print("%2d %6.0f %6.0f %6.0f %8.0f %6.0f   %.3f" % (c, row["top"], row["chain"], row["drain"], row["barrier"], row["busy"], row["last_share"]))
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
