"""GPU: the ONE line the driver runs (`python bench.py --gpus 1 --steps K --warmup W`) carries the contract's fields, a roofline and a parity verdict for
the headline AND for every other BASELINE config, the backward and the pre-normalised contract (round-4 review item 1).  Short K / W here: the numbers are
not judged, the shape of the line is."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

LEGS = ("backward2d_kitti_B64", "prenorm_kitti_B64", "head_kitti_B64", "config4_kitti_sparse_B32", "config2_nyu_B16", "config3_as_written_share_B8", "config1_plumbing_B1",
        "config5_vol3d_B4")


def test_driver_line_carries_every_config():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--prewarm-s", "0.05",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "config 3" in d["config"]["workload"] and d["config"]["B_per_gpu"] == 64 and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 64 * 304 * 1216 * 40 and r["traffic"] and r["traffic"] >= r["algorithmic_bytes_per_launch"]
    assert d["parity_checked"]["ok"] is True
    # value follows from the timed region: images x pixels x iterations x steps / elapsed
    assert abs(d["value"] - 64 * 304 * 1216 * 24 / 1e6 / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]
    assert set(d["configs"]) == set(LEGS), sorted(d["configs"])
    for name in LEGS:
        leg = d["configs"][name]
        assert "error" not in leg, (name, leg.get("error"))
        assert leg["parity_checked"]["ok"] is True, (name, leg["parity_checked"])
        assert leg["steps"] == 3 and leg["ms_per_step"] > 0 and leg["roofline"]["device_ms_per_launch"] > 0
        assert 0 < leg["roofline"]["frac"] < 1 and leg["roofline"]["bound"] == ("mfma" if name.startswith("head_") else "hbm")
    assert d["configs"]["config5_vol3d_B4"]["parity_checked"]["oracle_full_volume"]["voxels"] == 32 * 160 * 608
    assert d["configs"]["config4_kitti_sparse_B32"]["roofline"]["algorithmic_bytes_per_launch"] == 32 * 304 * 1216 * 44
    assert d["configs"]["backward2d_kitti_B64"]["roofline"]["algorithmic_bytes_per_launch"] == 64 * 304 * 1216 * 76
    head = d["configs"]["head_kitti_B64"]   # round 6: the producer of config 3's inputs, priced against the fp32 peak
    assert head["roofline"]["peak"] == 157.3 and head["roofline"]["unit"] == "TFLOP/s" and head["roofline"]["algorithmic_flop_per_launch"] == 2.0 * 81 * 64 * 64 * 152 * 608
    assert head["head_plus_forward_ms"] > head["roofline"]["device_ms_per_launch"]
