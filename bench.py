#!/usr/bin/env python
"""bench.py -- CSPN propagation throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (all n_iter propagation iterations, one C-ABI call)
over this rank's batch of synthetic affinity + depth tensors, already resident in HBM.
Workload (config.workload): BASELINE.json configs[2] -- 2D CSPN 3x3, 24 iterations, KITTI
304x1216 -- at 64 images PER GPU (the full config-3 batch fits one MI355X: 1.0 GB of 288 GB);
the batch shards embarrassingly, so N>1 is weak scaling with no data-path collective (--scaling strong --global-batch 64 shards
ONE batch over the ranks instead: BASELINE config 3 as written, 64 / N images per GPU).  The only
collective is the one-time RCCL broadcast of a backbone-sized weight buffer (outside the timed
region, reported as broadcast_ms).  Data: image i of the global batch is seeded by 1000 + i on the
CPU.  Before the counted warm-up the launch runs untimed for --prewarm-s seconds
(shader clocks settle; reported as prewarm_s); after the timed region the oracle checks a sample
of `out` (parity_checked).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the boxes export it,
# a shell that does not gets it here, before the HIP runtime is loaded
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# ---- the printed line stays below the driver's 8 KB tail: prose lives in profiles/bench_legend.json (this table, written by
# `python bench.py --write-legend`, checked by tests/test_bench_line.py), the line carries "@code" references and numbers ----
LEGEND = {
    "oracle": "oracle/cspn_oracle.c: C restatement of reference cspn_pytorch/models/cspn.py:42-172, pinned to golden vectors produced by the unmodified "
              "reference (tests/golden/cspn2d_golden.npz); run AFTER the timed region on a sample of what the timed launches left in `out`",
    "oracle_bwd": "oracle/backward.py, pinned to the reference's autograd gradients (tests/golden/cspn2d_grad_golden.npz)",
    "data2d": "synthetic: randn affinity, uniform depth * scale; image i of the global batch from torch.Generator().manual_seed(1000 + i) on the CPU "
              "(SURVEY 8d), copied to HBM before the timed region",
    "data3d": "synthetic: uniform gates normalised over the 26 channels, uniform feature volume; generated on device",
    "k_tsw4": "cspn2d_tsw4_kernel: gfx950 assembly main loop, round 6: a ring of 12 waves x 3 rows at 168 VGPRs (3 waves per SIMD), rows by LDS-DMA; ONE "
              "launch per forward, every workgroup builds the row stream of its piece of the linear plan in LDS, nothing else runs in the timed region",
    "k_tsw": "cspn2d_tsw_kernel: gfx950 assembly main loop, the 8 waves x 4 rows ring of rounds 1-5; ONE launch per forward (same plan)",
    "k_tsw_short": "cspn2d_tsw_kernel: a short first pass of n_iter % 24 iterations (+ n_iter // 24 passes of 24) of the 8 x 4 ring",
    "k_padded": "normalize2d_pitch_kernel / pad_rows_kernel -> the fused path on rows padded to a multiple of 4 columns -> unpad_rows_kernel (whole forward)",
    "k_fused_cxx": "cspn2d_fused_kernel (compiler-generated ring kernel, one launch per forward)",
    "k_stepwise": "fold2d_kernel + n_iter x step2d_kernel (whole forward)",
    "k_prenorm": "the dispatcher's ring kernel, norm 3 (CSPN_NORM_PRENORM): the guidance planes are the reference's gate_wb (cspn.py:85-144), cooking reduced to "
                 "sigma = sum w, c' = (1 - sigma) H0; same 40 B/pixel",
    "k_bwd": "one cspn2d_backward_f32 call: cspn2d_tsw_kernel history sweep + adjoint sweep + bwd_final_mx_kernel, back to back inside the event pair "
             "(per call, not per kernel)",
    "k_p3": "cspn3d_persistent_kernel: one launch per forward, gates read once and resident in registers for all steps; algorithmic bytes = 26 gates + value in, "
            "value out = 112 B/voxel ONCE per forward (SURVEY 8d)",
    "k_step3d": "step3d_direct_kernel: one launch per iteration (112 B/voxel each); whole_forward_frac prices all iterations against a single pass over the inputs",
    "field_traffic": "`traffic` / `traffic_key`: HBM-side bytes per launch from separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; the entry `traffic_key` of "
               "profiles/pmc_traffic.json names the profile it comes from and the gfx950 corrections applied",
    "unpinned3d": "the Paddle op's source is not in the reference tree: 3D parity is UNPINNED; the two HIP paths are compared on every voxel and with "
                  "oracle/cspn_oracle.c on one full 32x160x608 volume",
    "cpu_port": "kind port = oracle/cspn_oracle.c (the C restatement of reference cspn.py, OpenMP over images, one image per thread) on the host's threads: NOT the "
                "reference's own code path; /root/reference does not exist on the GPU box",
    "cpu_refops": "the reference's OWN path on the CPU: tools/torch_ops_baseline.py = the torch op sequence reference cspn.py:42-83 launches (eight padded copies + "
                  "cat, product, 1x1x1 Conv3d channel sum, per iteration), pinned to the unmodified reference's golden vectors (tests/test_oracle.py); ONE image per "
                  "forward on torch-CPU in a child process at 16 / 64 / physical-core thread counts; `value` = the fastest that ran, `physical_cores` = "
                  "torch.set_num_threads(number of physical cores)",
    "cfg1": "BASELINE config 1: 2D CSPN 3x3, 12 iters, 1x1x228x304 (the reference's CPU-runnable plumbing case)",
    "cfg2": "BASELINE config 2: 2D CSPN 3x3, 24 iters, NYUv2 228x304, batch 16 on one GPU",
    "cfg3": "BASELINE config 3: 2D CSPN 3x3, 24 iters, KITTI 304x1216 (batch 64 sharded over 8 GPUs as written: 8 per GPU; the headline runs 64 per GPU)",
    "cfg4": "BASELINE config 4: 2D CSPN 3x3 + sparse-depth replacement (500-point mask), 24 iters, KITTI 304x1216, batch 32 on one GPU",
    "cfg5": "BASELINE config 5: 3D CSPN 3x3x3, 12 iters, 32x160x608 cost volume, batch 4 on one GPU; gates pre-normalised by the caller (Paddle contract)",
    "head": "the producer of BASELINE config 3's inputs (SURVEY 8f-2): both guidance heads of the reference backbone (torch_resnet_cspn_nyu.py:187-206, 372-373: Unpool + "
            "3x3 conv 64 -> 8 and 64 -> 1) on a [64,64,152,608] feature map; algorithmic work = the 9 non-zero products per input pixel, input channel and output "
            "plane (2 x 81 x C FLOP per input pixel); head_plus_forward_ms = this head + the headline forward on its outputs, one stream, one event pair; "
            "torch_heads_ms = the reference's op sequence for the heads (conv_transpose2d + two conv2d, MIOpen) on the same GPU; gradient_ms = "
            "cspn_guidance_head_backward_f32 on the same shape (dL/dx alone, dL/dW alone; parity: tests/test_head.py)",
    "k_head": "head_raw_kernel (cspn_head.hip): packed fp32 FMAs (no fp32 MFMA gain on gfx950: v_mfma_f32 and v_pk_fma_f32 share the 157.3 TFLOP/s peak), feature "
              "rows by LDS-DMA 8 channels ahead, weights as scalar operands",
    "oracle_head": "oracle/oracle.py guidance_head_oracle (numpy, fp64 accumulation), pinned to the unmodified reference's modules (tests/golden/head_golden.npz)",
    "field_training_mode": "backward leg, `training_mode`: cspn2d_forward_history_f32 (the forward that also keeps H_4 .. H_20 and the folded planes: what "
                           "cspn_amd.Affinity_Propagate runs when an input requires grad) and cspn2d_backward_history_f32 (adjoint sweep + final pass from that history), "
                           "device ms per call by HIP events; `same_gradients`: bitwise equal to the recomputing call's",
    "bwd": "cspn2d_backward_f32: gradient of BASELINE config 3's forward w.r.t. guidance and blur_depth (reference train.py:196-198), 76 B/pixel algorithmic",
}
BACKBONE_PARAMS = 256_078_272  # resnet50-CSPN fp32 parameter count (probed)

WORKLOADS = {
    # name: (H, W, n_iter, sparse, depth scale, description)
    "kitti": (304, 1216, 24, False, 80.0, "@cfg3"),
    "kitti_sparse": (304, 1216, 24, True, 80.0, "@cfg4"),
    "nyu": (228, 304, 24, True, 10.0, "@cfg2"),
    "plumbing": (228, 304, 12, False, 10.0, "@cfg1"),
    # off the fast path (round-4 review, weak 8): measured for the record, not part of the driver's line
    "kitti_n12": (304, 1216, 12, False, 80.0, "KITTI 304x1216, 12 iterations (BASELINE config 1's count; the reference's own defaults are 24, cspn_paddle/demo.py:92): one short pass of the assembly loop (round 5: a row is stored when it completes level 12)"),
    "kitti_w1218": (304, 1218, 24, False, 80.0, "304x1218 (W % 4 != 0), 24 iterations: rows padded to 1220 columns in the workspace, fused path on those (round 5; before: fold + one launch per iteration)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="kitti", choices=sorted(WORKLOADS) + ["vol3d"])
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch-per-gpu images on every rank.  strong: --global-batch images (BASELINE config 3: "
                         "batch 64 sharded over the GPUs of the node, the scatter of reference cspn_pytorch/eval.py:115-118) cut into "
                         "contiguous per-rank chunks, rank r owns images [r G / N, (r + 1) G / N)")
    ap.add_argument("--global-batch", type=int, default=64)
    ap.add_argument("--algo", default="auto", choices=["auto", "stepwise", "fused", "fused_cxx", "fused_padded"])
    ap.add_argument("--norm-type", default="8sum", choices=["8sum", "8sum_abs"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--write-legend", action="store_true", help="write profiles/bench_legend.json (the prose behind the line's @codes) and exit; no GPU needed")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N = 1, default workload: do not also time BASELINE configs 4, 2, 3-as-written's per-GPU share, 5 and the 2D "
                         "backward after the headline's timed region (reported under \"configs\")")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="seconds of untimed launches before the counted warm-up (lets the shader clock settle: the first "
                         "few dozen launches of a process run ~20 %% slower); outside the timed region, reported as prewarm_s")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--pmc-calib", action="store_true",
                    help="profiling runs: one torch elementwise kernel over the depth batch before the loop (known byte count, "
                         "calibrates FETCH_SIZE / WRITE_SIZE in the same rocprofv3 trace)")
    ap.add_argument("--no-broadcast", action="store_true")
    ap.add_argument("--event-every", type=int, default=1,
                    help="bracket every N-th timed launch with a HIP event pair (1: every launch).  Event records between kernels cost "
                         "inter-kernel gap (wall time per step), not kernel time; the reported device time is the mean over the bracketed launches")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="N > 1, weak scaling: do not also time BASELINE config 3 as written (--global-batch images sharded over the ranks)")
    ap.add_argument("--plan-mode", type=int, default=0, choices=[0, 1, 2, 3, 8, 9, 10, 16, 17, 18],
                    help="A/B measurements only (through the hook library, not the ABI): 1 = the linear plan without XCD-aware placement, "
                         "2 = the band-group plan of rounds 1-3, 3 = the round-3 loop (experiment builds); 0 = the product (the ABI call)")
    ap.add_argument("--layout", default="planar", choices=["planar", "prenorm", "sited8"],
                    help="prenorm (SURVEY 8f-2, DESIGN.md 3.6): the guidance is normalised ONCE, outside the timed region, by "
                         "cspn2d_normalize_f32 (what a producer head with a fused epilogue would emit: the reference's gate_wb) and the timed "
                         "step is the forward with norm CSPN_NORM_PRENORM; parity is still checked against the oracle on the RAW guidance.  "
                         "sited8: the closed round-2 experiment (experiment builds only, hook library)")
    return ap.parse_args()


def synth(B, H, W, scale, sparse, device, first=0, seed0=1000):
    """per-image seeding -- image i (GLOBAL index, first..first+B-1) is generated on the CPU from
    torch.Generator().manual_seed(seed0 + i) and copied to the device, so every sharding of the batch sees the same
    bits (tests/helpers.config_inputs is the same recipe)."""
    g = torch.empty(B, 8, H, W)
    h = torch.empty(B, 1, H, W)
    s = torch.empty(B, 1, H, W) if sparse else None
    for i in range(B):
        gen = torch.Generator().manual_seed(seed0 + first + i)
        g[i] = torch.randn(8, H, W, generator=gen)
        h[i] = torch.rand(1, H, W, generator=gen) * scale
        if sparse:
            m = (torch.rand(1, H, W, generator=gen) < 500.0 / (H * W)).float()
            s[i] = m * (torch.rand(1, H, W, generator=gen) * scale + 0.1)
    if str(device) == "cpu":
        return g, h, s
    return g.to(device), h.to(device), (s.to(device) if sparse else None)


def parity_check(out, g, h, s, n_iter, norm, rtol=1e-4):
    """after the timed region: the oracle (CPU port of reference cspn.py, test infrastructure) on a sample of this rank's
    batch -- the first image, the two in the middle (at 64 images: 31, which the plan cuts in the middle between two workgroups,
    and 32, which it does not) and the last one (whose bands the last four CUs share) -- against what the timed launches left in
    `out`."""
    from oracle import cspn2d_oracle
    B = out.shape[0]
    idx = sorted({0, max(0, B // 2 - 1), B // 2, B - 1})
    ref = cspn2d_oracle(g[idx].cpu(), h[idx].cpu(), None if s is None else s[idx].cpu(), n_iter, norm)
    ref = torch.from_numpy(ref) if not isinstance(ref, torch.Tensor) else ref
    got = out[idx].cpu()
    if not torch.equal(torch.isfinite(got), torch.isfinite(ref)):
        return {"ok": False, "images": idx, "error": "non-finite pattern differs"}
    fin = torch.isfinite(ref)
    scale = float(ref[fin].abs().max()) if fin.any() else 1.0
    d = (got[fin] - ref[fin]).abs()
    err = float(d.max() / scale) if fin.any() else 0.0
    elem_ok = bool((d <= 1e-6 * scale + rtol * ref[fin].abs()).all()) if fin.any() else True
    return {"ok": bool(err <= rtol and elem_ok), "images": idx, "max_rel_err": err, "rtol": rtol,
            "against": "@oracle"}


_REFOPS_CHILD = r"""
import json, os, sys, time
import torch
sys.path.insert(0, sys.argv[1])
import bench
from tools.torch_ops_baseline import affinity_propagate_torch_ops
H, W, n_iter, sparse, scale, norm, cores, budget = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1", float(sys.argv[6]), sys.argv[7], int(sys.argv[8]), float(sys.argv[9])
g, h, s = bench.synth(1, H, W, scale, sparse, "cpu", first=0, seed0=4242)
t_start = time.perf_counter()
for nthr in sorted({min(cores, 16), min(cores, 64), cores}):
    torch.set_num_threads(nthr)
    affinity_propagate_torch_ops(g, h, s, 1, norm)            # thread pool, primitive caches
    t0 = time.perf_counter()
    affinity_propagate_torch_ops(g, h, s, 2, norm)            # normalisation + 2 iterations: what a full run will cost
    t2 = time.perf_counter() - t0
    projected = t2 * (n_iter + 2.0) / 4.0
    left = budget - (time.perf_counter() - t_start)
    if projected > left:
        print(json.dumps({"threads": nthr, "skipped": "projects to %.1f s, %.1f s left" % (projected, left),
                          "projected_value": round(H * W * n_iter / 1e6 / projected, 2)}), flush=True)
        continue
    reps, t_total = 0, 0.0
    while reps < 1 or (t_total < 1.5 and reps < 8):
        t0 = time.perf_counter()
        affinity_propagate_torch_ops(g, h, s, n_iter, norm)
        t_total += time.perf_counter() - t0
        reps += 1
    print(json.dumps({"threads": torch.get_num_threads(), "value": round(H * W * n_iter * reps / 1e6 / t_total, 2), "reps": reps}), flush=True)
"""


def reference_op_sequence_cpu(H, W, n_iter, sparse, scale, norm, cores, budget_s=24.0):
    """tools/torch_ops_baseline.py -- the op sequence reference cspn.py:42-83 launches, pinned to the unmodified reference's golden
    vectors -- on torch-CPU, ONE image per forward, at 16 / 64 / all host threads.  Runs in a CHILD process (its own OpenMP runtime:
    in this process the oracle's 256 spinning OpenMP threads made torch-CPU 100x slower, profiles/r05_perf_notes.md) and is
    bounded: a thread count whose full forward projects beyond the leg's budget is reported as skipped with the projection."""
    import subprocess
    try:
        env = dict(os.environ, OMP_WAIT_POLICY="PASSIVE", HIP_VISIBLE_DEVICES="")
        p = subprocess.run([sys.executable, "-c", _REFOPS_CHILD, ROOT, str(H), str(W), str(n_iter), "1" if sparse else "0", str(scale), norm,
                            str(cores), str(budget_s)], capture_output=True, text=True, timeout=budget_s + 30, env=env)
        legs = [json.loads(line) for line in p.stdout.splitlines() if line.startswith("{")]
        if not legs:
            return {"error": "no result (rc %d): %s" % (p.returncode, p.stderr[-300:])}
    except subprocess.TimeoutExpired as ex:
        legs = [json.loads(line) for line in (ex.stdout or b"").decode("utf-8", "replace").splitlines() if line.startswith("{")]
        if not legs:
            return {"error": "timed out after %.0f s without a result" % (budget_s + 30)}
    except Exception as ex:   # noqa: BLE001 -- reported in the line, never hidden
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    done = [x for x in legs if "value" in x]
    allc = [x for x in legs if x["threads"] == cores]   # (`cores` = the host's PHYSICAL cores: see cpu_baseline)
    best = max(done, key=lambda x: x["value"]) if done else None
    out = {"unit": "Mpix*iters/s", "kind": "reference_op_sequence", "by_threads": legs,
           "what": "@cpu_refops", "image": [H, W], "n_iter": n_iter, "torch": torch.__version__.split("+")[0]}
    if best:
        out["value"], out["cores"] = best["value"], best["threads"]
    if allc:
        out["physical_cores"] = allc[0]
    return out


def physical_cores():
    """physical cores of the host (SMT siblings counted once): the thread count at which the reference's torch-CPU path is timed as "all cores"
    (round-5 review: 256 torch threads on 128 cores oversubscribe the ~30 small ops per iteration)"""
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except Exception:   # noqa: BLE001
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(H, W, n_iter, sparse, scale, norm):
    """Two CPU legs on this host's cores, each on a bounded sample of the same workload:
    (1) kind "port": oracle/cspn_oracle.c (C restatement of reference cspn.py:42-172, OpenMP over images, one image per thread) on
        ALL host threads;
    (2) "reference_op_sequence": tools/torch_ops_baseline.py -- the op sequence reference cspn.py:42-83 launches (eight padded
        copies + cat, product, 1x1x1 Conv3d channel sum, per iteration), pinned to the unmodified reference's golden vectors by
        tests/test_oracle.py -- on torch-CPU with all host threads.  /root/reference itself does not exist on the GPU box."""
    from oracle import cspn2d_oracle, oracle_threads, set_oracle_threads
    cores = os.cpu_count() or 1
    set_oracle_threads(cores)
    threads = oracle_threads()
    nimg = threads  # one image per thread per repetition
    g, h, s = synth(min(nimg, 64), H, W, scale, sparse, "cpu", first=0, seed0=4242)
    if nimg > 64:   # (the sample's bits do not matter beyond 64 distinct images: tile them)
        reps_t = (nimg + 63) // 64
        g, h = g.repeat(reps_t, 1, 1, 1)[:nimg].contiguous(), h.repeat(reps_t, 1, 1, 1)[:nimg].contiguous()
        s = s.repeat(reps_t, 1, 1, 1)[:nimg].contiguous() if s is not None else None
    cspn2d_oracle(g[:1], h[:1], None if s is None else s[:1], 1, norm)  # build/load + warm
    reps, t_total = 0, 0.0
    while reps < 3 or (t_total < 4.0 and reps < 20):
        t0 = time.perf_counter()
        cspn2d_oracle(g, h, s, n_iter, norm)
        t_total += time.perf_counter() - t0
        reps += 1
    mpix_iters = nimg * H * W * n_iter * reps / 1e6
    res = {
        "value": round(mpix_iters / t_total, 2),
        "unit": "Mpix*iters/s",
        "cores": threads,
        "kind": "port",
        "what": "@cpu_port",
        "sample": "%d images %dx%d x %d iters x %d reps; host threads %d" % (nimg, H, W, n_iter, reps, cores),
    }
    # the same port on 64 threads (rounds 1-4 capped it there: on this 256-thread host the memory-bound port is FASTER on 64)
    try:
        if threads > 64:
            set_oracle_threads(64)
            t0 = time.perf_counter()
            cspn2d_oracle(g[:64], h[:64], None if s is None else s[:64], n_iter, norm)
            dt = time.perf_counter() - t0
            res["port_on_64_threads"] = {"value": round(64 * H * W * n_iter / 1e6 / dt, 2), "unit": "Mpix*iters/s", "cores": 64,
                                         "sample": "64 images, 1 rep"}
            set_oracle_threads(cores)
    except Exception as ex:   # noqa: BLE001
        res["port_on_64_threads"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    res["reference_op_sequence"] = reference_op_sequence_cpu(H, W, n_iter, sparse, scale, norm, physical_cores())
    return res


def measure_vol3d(a, lib, _lib, dev, dist, world, rank, shared_gpu, B, algo3, steps, warmup, prewarm_s, cpu_base=False):
    """BASELINE config 5: 3x3x3 propagation, 12 iterations, 32x160x608 volume, batch B per GPU, gates normalised by the
    caller and used as given (the fluid.layers.affinity_propagate contract, reference cspn_paddle/demo.py:41-52).
    algo3 0 / 2: the persistent kernel (gates resident in registers across all steps: one pass over the 104 B/voxel
    gate tensor per forward); 1: one step3d_direct_kernel launch per iteration.  -> the result object (rank 0) or None"""
    D, H, W, n_iter = 32, 160, 608, 12
    gen = torch.Generator(device=dev).manual_seed(5000 + rank)
    g = torch.rand(B, 26, D, H, W, generator=gen, device=dev)
    g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, generator=gen, device=dev)
    out = torch.empty_like(h)
    norm = _lib.NORM_TYPES["none"]
    ws_bytes = lib.cspn3d_workspace_bytes_ex(B, D, H, W, n_iter, norm, 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(lib.cspn3d_forward_f32_algo(g.data_ptr(), h.data_ptr(), None, out.data_ptr(), B, D, H, W, n_iter, norm, algo3,
                                               ws.data_ptr(), ws_bytes, stream.cuda_stream), "cspn3d_forward_f32_algo")

    if dist is not None:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        warm_now, pre_now = 0, 0.0
    else:
        warm_now, pre_now = warmup, prewarm_s
    elapsed, dev_ms = timed_leg(step, stream, steps, warm_now, pre_now)
    if dist is not None:
        dist.barrier()
    dev_ms_avg = sum(dev_ms) / len(dev_ms)
    # parity of what the timed launches left in `out`: the other 3D path on every voxel + the CPU oracle on ONE FULL volume
    # (volume 1: the chunks of the persistent kernel are cut across the row of volumes, so it has a neighbour on both sides)
    parity = None
    if not a.no_parity_check:
        other = torch.empty_like(out)
        _lib.check(lib.cspn3d_forward_f32_algo(g.data_ptr(), h.data_ptr(), None, other.data_ptr(), B, D, H, W, n_iter, norm,
                                               1 if algo3 != 1 else 0, ws.data_ptr(), ws_bytes, stream.cuda_stream), "other 3D path")
        torch.cuda.synchronize()
        err = float((out - other).abs().max() / other.abs().max())
        from oracle import cspn3d_oracle, set_oracle_threads
        set_oracle_threads(os.cpu_count() or 1)
        vi = 1 if B > 1 else 0
        ref = torch.from_numpy(cspn3d_oracle(g[vi:vi + 1].cpu(), h[vi:vi + 1].cpu(), None, n_iter, "none"))
        got = out[vi:vi + 1].cpu()
        eo = float((got - ref).abs().max() / ref.abs().max())
        elem = bool(((got - ref).abs() <= 1e-6 * float(ref.abs().max()) + 1e-4 * ref.abs()).all())
        del other
        parity = {"ok": bool(err <= 1e-5 and eo <= 1e-4 and elem and torch.isfinite(out).all()), "max_rel_diff_between_3d_paths": err,
                  "oracle_full_volume": {"volume": vi, "voxels": D * H * W, "max_rel_err": eo, "rtol": 1e-4},
                  "pinned": False, "note": "@unpinned3d"}
    if dist is not None:
        t = torch.tensor([elapsed, dev_ms_avg], device="cpu" if shared_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms_avg = float(t[0]), float(t[1])
    if rank != 0:
        return None
    vox = B * D * H * W
    persistent = algo3 != 1
    fwd_frac = vox * 112 / (dev_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS
    traffic, source = pmc_traffic("vol3d_B%d_%s" % (B, "persistent" if persistent else "stepwise"))
    if persistent:
        roof = {"bound": "hbm", "kernel": "@k_p3",
                "achieved": round(vox * 112 / (dev_ms_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(fwd_frac, 4), "traffic": traffic, "algorithmic_bytes_per_launch": vox * 112,
                "device_ms_per_launch": round(dev_ms_avg, 4), "device_ms_min": round(dev_ms[0], 4), "whole_forward_frac": round(fwd_frac, 4),
                }
    else:
        launch_ms = dev_ms_avg / n_iter
        roof = {"bound": "hbm", "kernel": "@k_step3d",
                "achieved": round(vox * 112 / (launch_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(vox * 112 / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": vox * 112, "device_ms_per_launch": round(launch_ms, 4),
                "whole_forward_frac": round(fwd_frac, 4)}
    if source:
        roof["traffic_key"] = source
    res = {
        "metric": "CSPN iterations/sec (Mvox*iters/s), 3x3x3x12", "value": round(world * vox * n_iter * steps / 1e6 / elapsed, 1),
        "unit": "Mvox*iters/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "@data3d",
        "parity_checked": parity,
        "config": {"workload": "@cfg5", "batch_per_gpu": B,
                   "B_per_gpu": B, "D": D, "H": H, "W": W, "n_iter": n_iter, "norm_type": "none",
                   "algo": "persistent" if persistent else "stepwise",
                   "parallelism": "batch-sharded x%d, no data-path collective" % world},
        "roofline": roof,
    }
    if cpu_base:
        from oracle import cspn3d_oracle, oracle_threads, set_oracle_threads
        set_oracle_threads(os.cpu_count() or 1)
        gc, hc = g[:, :, :, :40].cpu(), h[:, :, :, :40].cpu()   # bounded sample: a 32x40x608 slab of every volume
        cspn3d_oracle(gc[:1], hc[:1], None, 1, "none")
        t0 = time.perf_counter()
        cspn3d_oracle(gc, hc, None, n_iter, "none")
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(gc.shape[0] * D * 40 * W * n_iter / 1e6 / dt, 2), "unit": "Mvox*iters/s",
                               "cores": oracle_threads(), "kind": "port",
                               "what": "@cpu_port", "sample": "%d slabs 32x40x608" % gc.shape[0]}
    return res


def run_vol3d(a, lib, _lib, dev, dist, world, rank, shared_gpu):
    """--workload vol3d: BASELINE config 5 as the line's workload"""
    B = 4 if a.batch_per_gpu == 64 else a.batch_per_gpu
    algo3 = {"auto": 0, "stepwise": 1, "fused": 2, "fused_cxx": 2}.get(a.algo)
    if algo3 is None:
        raise SystemExit("--workload vol3d takes --algo auto | stepwise | fused (not %s: that is a 2D path)" % a.algo)
    res = measure_vol3d(a, lib, _lib, dev, dist, world, rank, shared_gpu, B, algo3, min(a.steps, 60), min(a.warmup, 20), a.prewarm_s,
                        cpu_base=(world == 1 and not a.no_cpu_baseline))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(key):
    """HBM bytes per launch from the PMC passes kept in profiles/pmc_traffic.json (separate rocprofv3 --pmc runs; the file says which
    profile each entry comes from), or (None, None)"""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(pmc)).get(key)
        if t:
            return t["hbm_bytes_per_launch"], key
    except Exception:   # noqa: BLE001
        pass
    return None, None


def roofline2d(m):
    """the `roofline` object of one timed 2D forward leg (40 / 44 B per pixel once per forward)"""
    W, n_iter, algo_name = m["W"], m["n_iter"], m["algo_name"]
    traffic, source = pmc_traffic("%s_B%d_%s%s" % (m["workload"], m["B"], algo_name, "_" + m["layout"] if m.get("layout") not in (None, "planar") else ""))
    r = {
        "bound": "hbm",
        "kernel": ("@k_tsw4" if ring2d(m["B"], m["H"], W, m["sparse"]) == 12 else "@k_tsw")
                  if algo_name == "fused" and W >= 256 and W % 4 == 0 and n_iter == 24
                  else "@k_tsw_short" if algo_name == "fused" and W >= 256 and W % 4 == 0
                  else "@k_padded" if algo_name == "fused_padded"
                  else "@k_fused_cxx" if algo_name.startswith("fused")
                  else "@k_stepwise",
        "achieved": round(m["achieved"], 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(m["achieved"] / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": m["alg_bytes"],
        "device_ms_per_launch": round(m["dev_ms_avg"], 4),
        "device_ms_min": round(m["dev_ms_min"], 4),
    }
    if source:
        r["traffic_key"] = source
    return r


LINE_LIMIT = 7500   # the driver keeps an 8 KB tail of stdout


def emit_line(res):
    """the ONE JSON line: compact separators, "@code" strings resolved by profiles/bench_legend.json, and -- should a future leg push it over
    the limit -- optional detail dropped key by key (recorded under "dropped") rather than a line the driver would cut"""
    res = dict(res)
    res["legend"] = "profiles/bench_legend.json"
    dump = lambda r: json.dumps(r, separators=(",", ":"))
    line = dump(res)
    dropped = []
    optional = [("cpu_baseline", "port_on_64_threads"), ("cpu_baseline", "reference_op_sequence", "by_threads"), ("notes",), ("prewarm_launches",),
                ("configs", "*", "parity_checked", "images"), ("configs", "*", "roofline", "device_ms_min"), ("configs", "*", "roofline", "peak"),
                ("configs", "*", "roofline", "unit"), ("configs", "*", "warmup"), ("configs", "*", "steps")]
    for path in optional:
        if len(line) <= LINE_LIMIT:
            break
        def drop(o, p):
            if not isinstance(o, dict):
                return
            if len(p) == 1:
                o.pop(p[0], None)
            elif p[0] == "*":
                for v in o.values():
                    drop(v, p[1:])
            else:
                drop(o.get(p[0]), p[1:])
        drop(res, path)
        dropped.append("/".join(path))
        res["dropped"] = dropped
        line = dump(res)
    return line


def ring2d(B, H, W, sparse):
    """which ring the dispatcher runs a full first pass of this shape on: 12 (cspn2d_tsw4.hip) or 8 (cspn2d_tsw.hip); 0: unknown (no hook library)"""
    try:
        from cspn_amd import _lib
        return int(_lib.load_hooks().cspn_debug_fused2d_ring(B, H, W, 1 if sparse else 0))
    except Exception:   # noqa: BLE001
        return 0


def timed_leg(step, stream, steps, warmup, prewarm_s):
    """prewarm (untimed, fixed wall time), `warmup` untimed calls, then `steps` calls each bracketed by a HIP event pair on the
    launch stream, the whole region bracketed by synchronize.  -> (wall seconds of the timed region, sorted per-call device ms)"""
    if prewarm_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < prewarm_s:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        step()
        e1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, sorted(e0.elapsed_time(e1) for e0, e1 in evs)


def leg_backward2d(lib, _lib, dev, g, h, s, n_iter, norm_name, steps, warmup, prewarm_s):
    """cspn2d_backward_f32 (what reference cspn_pytorch/train.py:196-198 back-propagates through) on the headline's batch: one call =
    the recomputing backward (history forward sweep + adjoint sweep + final pass).  Algorithmic bytes of a gradient that touches
    every tensor once: guidance 32 + blur 4 + grad_out 4 in, grad_guidance 32 + grad_blur 4 out = 76 B/pixel (+ 4 with a mask).
    Parity: oracle/backward.py (numpy restatement, pinned to gradients of the unmodified reference's autograd) on image 0."""
    B, _, H, W = g.shape
    gen = torch.Generator().manual_seed(77)
    go = torch.randn(B, 1, H, W, generator=gen).to(dev)
    gg, gh = torch.empty_like(g), torch.empty_like(h)
    norm = _lib.NORM_TYPES[norm_name]
    ws_bytes = lib.cspn2d_backward_workspace_bytes(B, H, W, n_iter)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(lib.cspn2d_backward_f32(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, go.data_ptr(),
                                           gg.data_ptr(), gh.data_ptr(), B, H, W, n_iter, norm, ws.data_ptr(), ws_bytes,
                                           stream.cuda_stream), "cspn2d_backward_f32")

    elapsed, dev_ms = timed_leg(step, stream, steps, warmup, prewarm_s)
    ms = sum(dev_ms) / len(dev_ms)
    # training mode (what cspn_amd.Affinity_Propagate does when an input requires grad): the forward keeps its checkpoints + folded planes (13 planes),
    # the backward starts from them -- no history sweep inside the backward call
    train = None
    try:
        hb = lib.cspn2d_history_bytes(B, H, W, n_iter)
        if hb:
            hist = torch.empty(hb, dtype=torch.uint8, device=dev)
            out = torch.empty_like(h)
            wsf_bytes = lib.cspn2d_workspace_bytes(B, H, W, n_iter)
            wsf = torch.empty(max(wsf_bytes, 1), dtype=torch.uint8, device=dev)
            wsh_bytes = lib.cspn2d_backward_history_workspace_bytes(B, H, W, n_iter)
            wsh = torch.empty(max(wsh_bytes, 1), dtype=torch.uint8, device=dev)
            gg2, gh2 = torch.empty_like(g), torch.empty_like(h)
            sp = s.data_ptr() if s is not None else None

            def fwd_hist():
                _lib.check(lib.cspn2d_forward_history_f32(g.data_ptr(), h.data_ptr(), sp, out.data_ptr(), hist.data_ptr(), hb, B, H, W, n_iter, norm,
                                                          wsf.data_ptr(), wsf_bytes, stream.cuda_stream), "cspn2d_forward_history_f32")

            def bwd_hist():
                _lib.check(lib.cspn2d_backward_history_f32(g.data_ptr(), h.data_ptr(), sp, go.data_ptr(), hist.data_ptr(), hb, gg2.data_ptr(), gh2.data_ptr(),
                                                           B, H, W, n_iter, norm, wsh.data_ptr(), wsh_bytes, stream.cuda_stream), "cspn2d_backward_history_f32")
            _, f_ms = timed_leg(fwd_hist, stream, steps, min(warmup, 5), 0.0)
            _, b_ms = timed_leg(bwd_hist, stream, steps, min(warmup, 5), 0.0)
            train = {"forward_keeping_history_ms": round(sum(f_ms) / len(f_ms), 4), "backward_from_history_ms": round(sum(b_ms) / len(b_ms), 4),
                     "same_gradients": bool(torch.equal(gg2, gg) and torch.equal(gh2, gh))}
            del hist, out, wsf, wsh, gg2, gh2
    except Exception as ex:   # noqa: BLE001
        train = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:120])}
    from oracle.backward import cspn2d_backward_oracle
    _, rg, rh = cspn2d_backward_oracle(g[:1].cpu().numpy(), h[:1].cpu().numpy(), None if s is None else s[:1].cpu().numpy(),
                                       go[:1].cpu().numpy(), n_iter, norm_name)
    rg, rh = torch.from_numpy(rg), torch.from_numpy(rh)
    eg = float((gg[:1].cpu() - rg).abs().max() / rg.abs().max())
    eh = float((gh[:1].cpu() - rh).abs().max() / rh.abs().max())
    alg = B * H * W * (80 if s is not None else 76)
    traffic, source = pmc_traffic("backward2d_kitti_B%d" % B)
    return {
        "workload": "@bwd", "shape": [B, H, W], "n_iter": n_iter,
        "value": round(B * H * W * n_iter * steps / 1e6 / elapsed, 1), "unit": "Mpix*iters/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "parity_checked": {"ok": bool(eg <= 2e-4 and eh <= 2e-4), "images": [0], "max_err_over_max_grad": {"guidance": eg, "blur": eh},
                           "tol": 2e-4, "against": "@oracle_bwd"},
        "roofline": {"bound": "hbm", "kernel": "@k_bwd",
                     "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_key": source,
                     "algorithmic_bytes_per_launch": alg, "device_ms_per_launch": round(ms, 4), "device_ms_min": round(dev_ms[0], 4)},
        "training_mode": train,
    }


F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector = fp32 matrix peak


def leg_head(lib, _lib, dev, B, n_iter, norm_name, steps, warmup, prewarm_s):
    """cspn_guidance_head_f32 (raw guidance + blur) on the feature map behind BASELINE config 3's batch, alone and followed by the forward"""
    import torch.nn.functional as F
    import cspn_amd
    from cspn_amd.train_utils import guidance_heads
    C, h, w = 64, 152, 608
    gen = torch.Generator().manual_seed(4242)
    x = torch.randn(B, C, h, w, generator=gen).to(dev)
    w6 = (torch.randn(8, C, 3, 3, generator=gen) / (3.0 * C ** 0.5)).to(dev)
    w5 = (torch.randn(1, C, 3, 3, generator=gen) / (3.0 * C ** 0.5) + 0.02).to(dev)
    stream = torch.cuda.current_stream(dev)
    keep = {}

    def step():
        keep["g"], keep["b"] = guidance_heads(x, w6, w5)

    elapsed, dev_ms = timed_leg(step, stream, steps, warmup, prewarm_s)
    ms = sum(dev_ms) / len(dev_ms)
    flop = 2.0 * 81 * C * B * h * w
    from oracle import guidance_head_oracle
    rg, rb = guidance_head_oracle(x[:1].cpu().numpy(), w6.cpu().numpy(), w5.cpu().numpy())
    rg, rb = torch.from_numpy(rg), torch.from_numpy(rb)
    eg = float((keep["g"][:1].cpu() - rg).abs().max() / rg.abs().max())
    eb = float((keep["b"][:1].cpu() - rb).abs().max() / rb.abs().max())
    traffic, source = pmc_traffic("head_kitti_B%d" % B)

    def e2e():
        g_, b_ = guidance_heads(x, w6, w5)
        keep["o"] = cspn_amd.cspn2d_forward(g_, b_, None, n_iter, norm_name)

    _, e2e_ms = timed_leg(e2e, stream, steps, min(warmup, 5), 0.0)
    up = torch.zeros(C, 1, 2, 2, device=dev)
    up[:, :, 0, 0] = 1

    def torch_heads():
        U = F.conv_transpose2d(x, up, stride=2, groups=C)
        keep["tg"], keep["tb"] = F.conv2d(U, w6, padding=1), F.conv2d(U, w5, padding=1)

    res = {
        "workload": "@head", "shape": [B, C, h, w], "value": round(B * 4 * h * w * steps / 1e6 / elapsed, 1), "unit": "Mpix/s (output pixels)",
        "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "parity_checked": {"ok": bool(eg <= 1e-5 and eb <= 1e-5), "images": [0], "max_err_over_max": {"guidance": eg, "blur": eb}, "tol": 1e-5,
                           "against": "@oracle_head"},
        "roofline": {"bound": "mfma", "kernel": "@k_head", "achieved": round(flop / (ms * 1e-3) / 1e12, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flop / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_key": source,
                     "algorithmic_flop_per_launch": flop, "hbm_bytes_unique": 4 * B * (C * h * w + 9 * 4 * h * w),
                     "device_ms_per_launch": round(ms, 4), "device_ms_min": round(dev_ms[0], 4)},
        "head_plus_forward_ms": round(sum(e2e_ms) / len(e2e_ms), 4),
    }
    try:   # the heads' gradient (cspn_guidance_head_backward_f32): dL/dx by packed FMAs, dL/dW on the fp32 matrix cores; 61.3 GFLOP each
        from cspn_amd.train_utils import guidance_heads_backward
        gg_, gb_ = keep["g"].clone(), keep["b"].clone()          # (any dL/dout serves the timing: the heads' own outputs)
        _, gx_ms = timed_leg(lambda: guidance_heads_backward(x, w6, w5, gg_, gb_, need_w=False), stream, min(steps, 10), 2, 0.0)
        _, gw_ms = timed_leg(lambda: guidance_heads_backward(x, w6, w5, gg_, gb_, need_x=False), stream, min(steps, 10), 2, 0.0)
        res["gradient_ms"] = {"grad_x": round(sum(gx_ms) / len(gx_ms), 4), "grad_w": round(sum(gw_ms) / len(gw_ms), 4)}
        del gg_, gb_
    except Exception as ex:   # noqa: BLE001
        res["gradient_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:120])
    try:
        _, t_ms = timed_leg(torch_heads, stream, min(steps, 5), 2, 0.0)
        res["torch_heads_ms"] = round(sum(t_ms) / len(t_ms), 3)
        res["raw_vs_torch_max_rel"] = float((keep["g"] - keep["tg"]).abs().max() / keep["tg"].abs().max())
    except Exception as ex:   # noqa: BLE001
        res["torch_heads_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:120])
    return res


def extra_configs(a, lib, _lib, dev, headline, notes):
    """N = 1: the other BASELINE configs and the training path, each timed like the headline (own prewarm, counted warm-up, K steps
    bracketed by synchronize, per-launch device time from HIP events on the launch stream, parity against the oracle AFTER the
    timed region), so that the ONE line the driver runs carries a driver-timed roofline fraction for every config."""
    out = {}
    steps, warmup = a.steps, a.warmup
    pre = min(a.prewarm_s, 0.5)

    def fwd2d(key, workload, batch):
        try:
            m = measure2d(a, lib, _lib, dev, None, 1, 0, False, "weak", steps, warmup, pre, notes, parity=not a.no_parity_check,
                          workload=workload, batch=batch)
            out[key] = {"workload": m["desc"], "batch": m["B"], "value": round(m["value"], 1), "unit": "Mpix*iters/s",
                        "steps": steps, "warmup": warmup, "ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
                        "parity_checked": m["parity"], "roofline": roofline2d(m)}
        except Exception as ex:   # noqa: BLE001 -- a failing leg is reported, it must not cost the headline
            out[key] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        torch.cuda.empty_cache()

    # the training path on the headline's own batch (its tensors are still resident)
    g, h, s = headline["tensors"]
    try:
        out["backward2d_kitti_B%d" % g.shape[0]] = leg_backward2d(lib, _lib, dev, g, h, s, headline["n_iter"], a.norm_type, steps, warmup, pre)
    except Exception as ex:   # noqa: BLE001
        out["backward2d_kitti_B%d" % g.shape[0]] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    del g, h, s
    headline["tensors"] = None
    torch.cuda.empty_cache()
    # the same batch with the normalisation done by the producer (timed forward = norm PRENORM on the reference's gate_wb;
    # the stand-alone producer epilogue, cspn2d_normalize_f32, is timed on its own and reported beside it, NOT inside the timed step)
    try:
        m = measure2d(a, lib, _lib, dev, None, 1, 0, False, "weak", steps, warmup, pre, notes, parity=not a.no_parity_check,
                      workload="kitti", batch=headline["B"], layout="prenorm")
        r = roofline2d(m)
        r["kernel"] = "@k_prenorm"
        out["prenorm_kitti_B%d" % m["B"]] = {
            "workload": m["desc"], "batch": m["B"], "norm": "prenorm",
            "value": round(m["value"], 1), "unit": "Mpix*iters/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(m["elapsed"] / steps * 1e3, 4), "parity_checked": m["parity"], "roofline": r,
            "producer_epilogue_standalone_ms": round(m["normalize_ms"], 4),
            "vs_headline_device_ms": round(m["dev_ms_avg"] / headline["dev_ms_avg"], 4)}
    except Exception as ex:   # noqa: BLE001
        out["prenorm_kitti_B%d" % headline["B"]] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    torch.cuda.empty_cache()
    try:
        out["head_kitti_B%d" % headline["B"]] = leg_head(lib, _lib, dev, headline["B"], headline["n_iter"], a.norm_type, steps, warmup, pre)
    except Exception as ex:   # noqa: BLE001
        out["head_kitti_B%d" % headline["B"]] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    torch.cuda.empty_cache()
    fwd2d("config4_kitti_sparse_B32", "kitti_sparse", 32)
    fwd2d("config2_nyu_B16", "nyu", 16)
    fwd2d("config3_as_written_share_B8", "kitti", 8)
    # config 1 is the reference's own CPU-runnable case: the engine on that shape (12 iterations: one short pass of the assembly loop)
    # with the reference's op sequence on the host cores beside it
    fwd2d("config1_plumbing_B1", "plumbing", 1)
    if "error" not in out["config1_plumbing_B1"] and not a.no_cpu_baseline:
        out["config1_plumbing_B1"]["cpu_reference_op_sequence"] = reference_op_sequence_cpu(228, 304, 12, False, 10.0, a.norm_type, physical_cores(), budget_s=8.0)
    try:
        out["config5_vol3d_B4"] = measure_vol3d(a, lib, _lib, dev, None, 1, 0, False, 4, 2, min(steps, 60), min(warmup, 20), pre)
    except Exception as ex:   # noqa: BLE001
        out["config5_vol3d_B4"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    torch.cuda.empty_cache()
    return out


_COLLECTIVES_BROKEN = [False]   # set by the first failing collective: every later one is skipped (the process group is in an undefined
                                # state, and a rank that carried on alone would leave the others blocked until the RCCL timeout)


def _reduce(dist, vals, op, dev, shared_gpu, notes):
    """all_reduce a few float64 values; a failing collective must not cost the throughput line: rank 0 then reports its own
    numbers, says so (`notes`, "reduced": false), and no further collective is attempted"""
    if dist is None or _COLLECTIVES_BROKEN[0]:
        return list(vals)
    try:
        t = torch.tensor(list(vals), device="cpu" if shared_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return [float(x) for x in t]
    except Exception as ex:   # noqa: BLE001 -- reported, never swallowed silently
        _COLLECTIVES_BROKEN[0] = True
        notes.append("all_reduce failed (%s: %s): rank-0 values reported, later collectives skipped" % (type(ex).__name__, str(ex)[:200]))
        return list(vals)


def _barrier(dist, notes):
    if dist is None or _COLLECTIVES_BROKEN[0]:
        return
    try:
        dist.barrier()
    except Exception as ex:   # noqa: BLE001
        _COLLECTIVES_BROKEN[0] = True
        notes.append("barrier failed (%s: %s): later collectives skipped" % (type(ex).__name__, str(ex)[:200]))


def measure2d(a, lib, _lib, dev, dist, world, rank, shared_gpu, scaling, steps, warmup, prewarm_s, notes, parity=True,
              workload=None, batch=None, keep_tensors=False, layout=None):
    """One timed leg of the 2D hot path: `steps` forwards over this rank's batch, bracketed by barrier + synchronize on both sides
    (max over ranks), per-launch device time from HIP events on the launch stream.  scaling 'weak': --batch-per-gpu images on
    every rank; 'strong': --global-batch images sharded (BASELINE config 3 as written)."""
    workload = workload or a.workload
    H, W, n_iter, sparse, scale, desc = WORKLOADS[workload]
    if scaling == "strong":
        from cspn_amd.dist import shard_range
        first, last = shard_range(a.global_batch, rank, world)
        B = last - first
        if B <= 0:
            raise SystemExit("--scaling strong: --global-batch %d leaves rank %d of %d without an image" % (a.global_batch, rank, world))
    else:
        B = batch or a.batch_per_gpu
        first = rank * B
    g, h, s = synth(B, H, W, scale, sparse, dev, first=first)
    algo_id = _lib.ALGOS[a.algo] or lib.cspn2d_auto_algo(B, H, W, n_iter)
    algo_name = {1: "stepwise", 2: "fused", 3: "fused_cxx", 4: "fused_padded"}[algo_id]
    norm = _lib.NORM_TYPES[a.norm_type]
    ws_bytes = lib.cspn2d_workspace_bytes(B, H, W, n_iter)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    out = torch.empty_like(h)
    stream = torch.cuda.current_stream(dev)

    g8, g_in = None, g
    layout = layout or a.layout
    if layout == "sited8":
        import cspn_amd
        g8 = cspn_amd.guidance_to_sited8(g, a.norm_type)
        torch.cuda.synchronize()
    normalize_ms = None
    if layout == "prenorm":
        import cspn_amd
        g_in = cspn_amd.cspn2d_normalize(g, a.norm_type)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            _lib.check(lib.cspn2d_normalize_f32(g.data_ptr(), g_in.data_ptr(), B, H, W, norm, stream.cuda_stream), "cspn2d_normalize_f32")
        e1.record(stream)
        torch.cuda.synchronize()
        normalize_ms = e0.elapsed_time(e1) / 5
        norm = _lib.NORM_TYPES["prenorm"]
    hooks = _lib.load_hooks() if (a.plan_mode or g8 is not None) else None

    def step():
        if hooks is not None and g8 is None:
            rc = hooks.cspn_debug_forward2d_plan(g_in.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, out.data_ptr(),
                                                 B, H, W, n_iter, norm, a.plan_mode, ws.data_ptr(), stream.cuda_stream)
        elif g8 is not None:
            rc = hooks.cspn_debug_forward_sited8(g8.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, out.data_ptr(),
                                                 B, H, W, n_iter, norm, stream.cuda_stream)
        else:
            rc = lib.cspn2d_forward_f32_algo(g_in.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None,
                                             out.data_ptr(), B, H, W, n_iter, norm, algo_id, ws.data_ptr(), ws_bytes,
                                             stream.cuda_stream)
        _lib.check(rc, "cspn2d_forward")

    if a.pmc_calib:
        _calib = h * 1.0   # reads and writes B*H*W*4 bytes
        del _calib
    # clock pre-warm: untimed full-work launches for a fixed wall time (outside the timed region)
    prewarm_done, prewarm_launches = 0.0, 0
    if prewarm_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < prewarm_s:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            prewarm_launches += 50
        prewarm_done = time.perf_counter() - t0
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    _barrier(dist, notes)
    torch.cuda.synchronize()
    # per-launch device time: HIP events on the stream the kernels are launched on
    every = max(1, int(a.event_every))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % every == 0 else None for i in range(steps)]
    t0 = time.perf_counter()
    for ev in evs:
        if ev is not None:
            ev[0].record(stream)
        step()
        if ev is not None:
            ev[1].record(stream)
    torch.cuda.synchronize()
    _barrier(dist, notes)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = sorted(ev[0].elapsed_time(ev[1]) for ev in evs if ev is not None)
    dev_ms_avg = sum(dev_ms) / len(dev_ms)
    par = None
    if parity:
        par = parity_check(out, g, h, s, n_iter, a.norm_type)
        if dist is not None:
            par["all_ranks_ok"] = bool(_reduce(dist, [1.0 if par["ok"] else 0.0], dist.ReduceOp.MIN, dev, shared_gpu, notes)[0] == 1.0)
    elapsed, dev_ms_avg = _reduce(dist, [elapsed, dev_ms_avg], dist.ReduceOp.MAX if dist is not None else None, dev, shared_gpu, notes)
    total_images = B * world
    if dist is not None and scaling == "strong":
        total_images = int(_reduce(dist, [float(B)], dist.ReduceOp.SUM, dev, shared_gpu, notes)[0])
        if total_images == B and world > 1:   # (the reduction failed: the shards are a.global_batch in total by construction)
            total_images = a.global_batch
    bytes_per_px = 44 if sparse else 40  # guidance 32 + blur 4 (+ sparse 4) + out 4
    alg_bytes = B * H * W * bytes_per_px  # per launch (one forward = all n_iter iterations), per GPU
    return {"tensors": (g, h, s) if keep_tensors else None, "workload": workload, "layout": layout, "normalize_ms": normalize_ms,
            "B": B, "H": H, "W": W, "n_iter": n_iter, "sparse": sparse, "scale": scale, "desc": desc, "algo_name": algo_name,
            "elapsed": elapsed, "dev_ms_avg": dev_ms_avg, "dev_ms_min": dev_ms[0], "parity": par, "total_images": total_images,
            "value": total_images * H * W * n_iter * steps / 1e6 / elapsed, "alg_bytes": alg_bytes,
            "achieved": alg_bytes / (dev_ms_avg * 1e-3) / 1e9, "prewarm_s": prewarm_done, "prewarm_launches": prewarm_launches}


def main():
    a = parse()
    if a.write_legend:
        with open(os.path.join(ROOT, "profiles", "bench_legend.json"), "w") as f:
            json.dump(LEGEND, f, indent=1, sort_keys=True)
            f.write("\n")
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    ndev = torch.cuda.device_count()
    shared_gpu = world > ndev  # more ranks than GPUs: only for exercising the launch path on a 1-GPU box
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import cspn_amd
    from cspn_amd import _lib
    lib = cspn_amd.load()

    dist = None
    notes = []
    broadcast_ms, broadcast_error, backend = None, None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared_gpu else "nccl"   # RCCL refuses two ranks on one device; gloo keeps the control flow identical
        # every rank says where it is BEFORE the first collective: if the rendezvous or RCCL hangs or dies, this is in the log
        sys.stderr.write("[bench.py rank %d/%d] device %d of %d: %s, backend %s (%s), broadcast %s bytes, HSA_ENABLE_IPC_MODE_LEGACY=%s\n" % (
            rank, world, dev_index, ndev, torch.cuda.get_device_name(dev_index), backend, "RCCL over xGMI" if backend == "nccl" else "CPU",
            "none" if (a.no_broadcast or shared_gpu) else str(BACKBONE_PARAMS * 4), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
        sys.stderr.flush()
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if not a.no_broadcast and not shared_gpu:
            # the one collective of the path (weights of the affinity backbone, once).  It is not part of the timed region: if it
            # fails, the throughput line is still printed, with "broadcast_ms": null and the error
            try:
                from cspn_amd.dist import broadcast_flat_
                buf = torch.empty(BACKBONE_PARAMS, dtype=torch.float32, device=dev).normal_()
                broadcast_flat_([buf[:1024]])  # communicator warm-up
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                dist.broadcast(buf, src=0)
                torch.cuda.synchronize()
                broadcast_ms = (time.perf_counter() - t0) * 1e3
                del buf
            except Exception as ex:   # noqa: BLE001
                broadcast_error = "%s: %s" % (type(ex).__name__, str(ex)[:300])
                sys.stderr.write("[bench.py rank %d] broadcast failed: %s\n" % (rank, broadcast_error))

    if a.workload == "vol3d":
        return run_vol3d(a, lib, _lib, dev, dist, world, rank, shared_gpu)
    extras_wanted = (world == 1 and not a.no_extra_configs and a.workload == "kitti" and a.scaling == "weak" and a.layout == "planar"
                     and not a.plan_mode and a.algo == "auto")
    m = measure2d(a, lib, _lib, dev, dist, world, rank, shared_gpu, a.scaling, a.steps, a.warmup, a.prewarm_s, notes,
                  parity=not a.no_parity_check, keep_tensors=extras_wanted)
    # N = 1, the driver's command: after the headline's timed region the other BASELINE configs and the backward, each timed the same way
    configs = extra_configs(a, lib, _lib, dev, m, notes) if extras_wanted else None
    # N > 1, weak scaling (the driver's command): BASELINE config 3 as written is the STRONG shape (batch 64 sharded over the GPUs),
    # so the same run also times that and reports it under "strong" (same K / W, same barriers, outside the first timed region)
    strong = None
    if world > 1 and a.scaling == "weak" and not a.no_strong_leg:
        try:
            if _COLLECTIVES_BROKEN[0]:
                raise SystemExit("a collective failed in the first leg")
            strong = measure2d(a, lib, _lib, dev, dist, world, rank, shared_gpu, "strong", a.steps, a.warmup, min(a.prewarm_s, 0.3), notes,
                               parity=not a.no_parity_check)
        except SystemExit as ex:
            notes.append("strong leg skipped: %s" % ex)
    if rank == 0:
        B, H, W, n_iter, sparse, scale, desc, algo_name = (m[k] for k in ("B", "H", "W", "n_iter", "sparse", "scale", "desc", "algo_name"))
        total_images, elapsed, dev_ms_avg = m["total_images"], m["elapsed"], m["dev_ms_avg"]
        res = {
            "metric": "CSPN iterations/sec (Mpix*iters/s), 3x3x24 at KITTI res" if a.workload.startswith("kitti")
                      else "CSPN iterations/sec (Mpix*iters/s)",
            "value": round(m["value"], 1),
            "unit": "Mpix*iters/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "@data2d",
            "prewarm_s": round(m["prewarm_s"], 3), "prewarm_launches": m["prewarm_launches"],
            "parity_checked": m["parity"],
            "config": {
                "workload": ("%s, batch %d per GPU" % ("BASELINE config 3: 2D CSPN 3x3, 24 iters, KITTI 304x1216" if desc == "@cfg3" else desc, B)) if a.scaling == "weak"
                            else "%s, global batch %d sharded over %d GPU(s) (%d images on rank 0)" % (desc, total_images, world, B),
                "B_per_gpu": B, "global_batch": total_images, "H": H, "W": W, "n_iter": n_iter, "norm_type": a.norm_type, "sparse": sparse,
                "algo": algo_name, "guidance_layout": a.layout, **({"plan_mode": a.plan_mode} if a.plan_mode else {}),
                "parallelism": "batch-sharded x%d, no data-path collective" % world
                               + (" (ranks share %d GPU(s): launch-path test only)" % ndev if shared_gpu else ""),
            },
            "roofline": roofline2d(m),
        }
        if world > 1:
            res["backend"] = backend
            res["devices"] = {"visible": ndev, "name": torch.cuda.get_device_name(dev_index)}
            res["broadcast_ms"] = round(broadcast_ms, 3) if broadcast_ms is not None else None
            res["broadcast_bytes"] = BACKBONE_PARAMS * 4 if broadcast_ms is not None else 0
            if broadcast_error is not None:
                res["broadcast_error"] = broadcast_error
        if strong is not None:
            res["strong"] = {
                "workload": desc, "sharded_over": world,
                "value": round(strong["value"], 1), "unit": "Mpix*iters/s", "ms_per_step": round(strong["elapsed"] / a.steps * 1e3, 4),
                "B_per_gpu": strong["B"], "global_batch": strong["total_images"],
                "roofline_frac_per_gpu": round(strong["achieved"] / HBM_PEAK_GBS, 4), "device_ms_per_launch": round(strong["dev_ms_avg"], 4),
                "parity_checked": strong["parity"]}
        if configs is not None:
            res["configs"] = configs
        if _COLLECTIVES_BROKEN[0]:
            res["reduced"] = False   # a collective failed: value / ms_per_step are rank 0's, not the max over ranks
        if notes:
            res["notes"] = notes
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(H, W, n_iter, sparse, scale, a.norm_type)
        print(emit_line(res), flush=True)
    if dist is not None:
        _barrier(dist, notes)
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        sys.stderr.write("[bench.py rank %s] failed:\n%s" % (os.environ.get("RANK", "0"), traceback.format_exc()))
        sys.stderr.flush()
        raise
