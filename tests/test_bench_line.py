"""bench.py's ONE JSON line must fit the driver's 8 KB stdout tail (round-5 review: the legs behind the headline were cut off): prose lives in
profiles/bench_legend.json, the line carries "@code" references and numbers, and emit_line() drops optional detail rather than overflow."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_legend_file_is_current_and_covers_every_code_the_source_uses():
    on_disk = json.load(open(os.path.join(ROOT, "profiles", "bench_legend.json")))
    assert on_disk == bench.LEGEND, "profiles/bench_legend.json is stale: python bench.py --write-legend"
    src = open(os.path.join(ROOT, "bench.py")).read()
    used = set(re.findall(r'"@([a-z0-9_]+)"', src)) - {"code"}
    assert used and used <= set(bench.LEGEND), sorted(used - set(bench.LEGEND))
    assert all(len(v) > 20 for v in bench.LEGEND.values())


def test_traffic_keys_resolve():
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("kitti_B64_fused", "kitti_sparse_B32_fused", "nyu_B16_fused", "kitti_B8_fused", "backward2d_kitti_B64", "vol3d_B4_persistent"):
        assert key in pmc and pmc[key]["hbm_bytes_per_launch"] > 0 and len(pmc[key]["source"]) > 40
        assert bench.pmc_traffic(key) == (pmc[key]["hbm_bytes_per_launch"], key)


def _fake_leg(i):
    return {"workload": "@cfg%d" % (i % 5 + 1), "batch": 64, "value": 2042930.0 + i, "unit": "Mpix*iters/s", "steps": 20, "warmup": 5, "ms_per_step": 0.2779,
            "parity_checked": {"ok": True, "images": [0, 31, 32, 63], "max_rel_err": 3.140307569537981e-07, "rtol": 0.0001, "against": "@oracle"},
            "roofline": {"bound": "hbm", "kernel": "@k_tsw4", "achieved": 3490.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.4363, "traffic": 1015298355,
                         "algorithmic_bytes_per_launch": 946339840, "device_ms_per_launch": 0.2711, "device_ms_min": 0.2672, "traffic_key": "kitti_B64_fused"}}


def test_emit_line_stays_below_the_limit_and_says_what_it_dropped():
    res = dict(_fake_leg(0), metric="CSPN iterations/sec (Mpix*iters/s), 3x3x24 at KITTI res", n_gpus=1, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32", data="@data2d", configs={"leg%d" % i: _fake_leg(i) for i in range(7)},
               cpu_baseline={"value": 906.9, "unit": "Mpix*iters/s", "cores": 256, "kind": "port", "what": "@cpu_port",
                             "port_on_64_threads": {"value": 1600.0, "cores": 64}, "reference_op_sequence": {"what": "@cpu_refops", "value": 157.0, "cores": 16,
                                                                                                             "by_threads": [{"threads": t, "value": 100.0} for t in (16, 64, 128)]}})
    line = bench.emit_line(res)
    assert len(line) <= bench.LINE_LIMIT and json.loads(line)["legend"] == "profiles/bench_legend.json"
    assert "dropped" not in json.loads(line)          # seven legs + both CPU baselines fit as they are
    res["notes"] = ["x" * 400] * 12                    # something unforeseen: optional detail goes, the legs' numbers stay
    line = bench.emit_line(res)
    d = json.loads(line)
    assert len(line) <= bench.LINE_LIMIT and d["dropped"] and all("frac" in leg["roofline"] for leg in d["configs"].values())


def test_committed_driver_line_of_this_round_fits_and_carries_every_leg():
    path = os.path.join(ROOT, "profiles", "r06_bench_driver.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("no driver-command line recorded yet this round")
    raw = open(path).read().strip()
    assert len(raw) <= bench.LINE_LIMIT, len(raw)
    d = json.loads(raw)
    legs = d["configs"]
    for key in ("backward2d_kitti_B64", "prenorm_kitti_B64", "config4_kitti_sparse_B32", "config2_nyu_B16", "config3_as_written_share_B8",
                "config1_plumbing_B1", "config5_vol3d_B4"):
        assert key in legs and "error" not in legs[key], key
        if key != "config1_plumbing_B1":
            assert legs[key]["roofline"]["frac"] > 0
    assert d["roofline"]["frac"] > 0.4 and d["cpu_baseline"]["kind"] == "port" and "reference_op_sequence" in d["cpu_baseline"]
    codes = set(re.findall(r'"@([a-z0-9_]+)"', raw))
    assert codes <= set(bench.LEGEND)
