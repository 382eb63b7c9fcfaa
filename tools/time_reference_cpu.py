"""tools/time_reference_cpu.py -- the UNMODIFIED reference module (/root/reference/cspn_pytorch/models/cspn.py, torch CPU, with the
harness-side `.cuda()` shim of oracle/ref_harness.py) timed in the AUTHORING container on BASELINE configs 1 and 3 (per image):
the "reference CPU path" number BASELINE.json's north_star names.  /root/reference does not exist on the GPU box, so this cannot
run there; bench.py's cpu_baseline leg times the C port on the GPU box's host cores instead.
usage: python tools/time_reference_cpu.py > profiles/r03_reference_cpu.md"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness  # noqa: E402


def run(B, H, W, n_iter, reps):
    g = torch.randn(B, 8, H, W)
    h = torch.rand(B, 1, H, W) * 80
    ref_harness.reference_forward(g, h, None, n_iter, "8sum")
    t0 = time.perf_counter()
    for _ in range(reps):
        ref_harness.reference_forward(g, h, None, n_iter, "8sum")
    dt = (time.perf_counter() - t0) / reps
    return dt, B * H * W * n_iter / dt / 1e6


def main():
    assert ref_harness.available()
    threads = torch.get_num_threads()
    print("# r03 -- the reference's own CPU path (unmodified cspn.py:42-83 under torch %s, CPU), authoring container\n" % torch.__version__)
    print("Host: %d cores visible (`os.cpu_count()`), torch intra-op threads %d; `/root/reference` is not shipped to the GPU box, so this" % (os.cpu_count(), threads))
    print("number comes from here and NOT from the MI355X host (bench.py's `cpu_baseline` times the C port there).\n")
    print("| workload | s / forward | Mpix*iters/s | threads |\n|---|---|---|---|")
    for name, B, H, W, n, reps in (("BASELINE config 1: 1 x 228x304, 12 iters", 1, 228, 304, 12, 10),
                                   ("KITTI 304x1216, 24 iters, 1 image", 1, 304, 1216, 24, 3),
                                   ("KITTI 304x1216, 24 iters, 4 images", 4, 304, 1216, 24, 2)):
        dt, rate = run(B, H, W, n, reps)
        print("| %s | %.4f | %.1f | %d |" % (name, dt, rate, threads))
    print("\nFor scale: the HIP engine on one MI355X does 1.93 M Mpix*iters/s on KITTI x 64 (BENCH_r03), the C port of the same path")
    print("on 64 host threads of the GPU box ~1.5 k Mpix*iters/s (`cpu_baseline` of the same bench line).")


if __name__ == "__main__":
    main()
