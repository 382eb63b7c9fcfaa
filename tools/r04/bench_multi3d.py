"""tools/r04/bench_multi3d.py -- C channels on shared gates: one persistent launch (gates loaded once per chunk) against C
single-channel launches; config 5's volume (4 x 32x160x608) and its cross-section, 12 steps."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    res = []
    for (B, D, H, W) in ((4, 32, 160, 608), (1, 32, 160, 152)):
        for C in (1, 2, 3, 4):
            gen = torch.Generator(device="cuda").manual_seed(1)
            g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda")
            g /= g.sum(1, keepdim=True)
            x = torch.rand(B, C, D, H, W, generator=gen, device="cuda")
            xs = [x[:, c:c + 1].contiguous() for c in range(C)]
            t_multi = timeit(lambda: cspn_amd.cspn3d_forward_multi(g, x, 12))
            t_loop = timeit(lambda: [cspn_amd.cspn3d_forward(g, xc, None, 12, "none", algo="persistent") for xc in xs])
            cspn_amd.cspn3d_check_status()
            res.append({"shape": [B, C, D, H, W], "ms_one_launch": round(t_multi, 4), "ms_per_channel_loop": round(t_loop, 4),
                        "speedup": round(t_loop / t_multi, 3)})
            print(res[-1], flush=True)
    json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04c/multi3d.json", "w"), indent=1)


if __name__ == "__main__":
    main()
