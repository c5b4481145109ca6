"""The JSON line bench.py prints is a contract with the driver (task statement, ④): check the last committed run
(profiles/r01e_bench.json, produced on an MI355X by tools/profile_round.sh + bench.py) field by field."""
import glob
import json
import os

from conftest import ROOT


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    d = _latest()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["steps"] > 0 and d["warmup"] >= 0 and d["data"].startswith("synthetic")
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01          # value = streams / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "segm_lite_v681" in d["config"]["workload"] and "batch=256" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    p = c["parity_sample"]
    assert p["mask_iou_min"] >= 0.999 and p["composite_max_abs_diff"] <= 1
