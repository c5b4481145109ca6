"""The JSON line bench.py prints is a contract with the driver (task statement, ④).  On a GPU box the test RUNS bench.py (short: 3 steps,
a 2-second CPU baseline sample, no extra configurations) and validates the fresh line field by field; without a GPU only the argument
handling and the multi-process plumbing can run (tests/test_dist_gloo.py covers the latter)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_bench_refuses_to_run_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_a_fresh_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--ramp-seconds", "0", "--no-extra-configs", "--cpu-seconds", "2",
                        "--profile-iters", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line"
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["data"].startswith("synthetic")
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01          # value = streams / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "segm_lite_v681" in d["config"]["workload"] and "batch=256" in d["config"]["workload"]
    r_ = d["roofline"]
    assert r_["bound"] in ("hbm", "mfma") and r_["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-3 and (r_["traffic"] is None or r_["traffic"] > 0)
    assert r_["traffic"] is None or ("traffic_source" in r_ and isinstance(r_["traffic_stale"], bool))   # counters come from a committed profile: the line says so, and whether the kernels changed since
    assert r_["traffic"] is None or 0 < r_["frac_counted_traffic"] <= 1.0             # the counted-bytes figure next to the algorithmic one
    assert d["host"]["model"] and d["host"]["logical_cpus"] >= 1                       # CPU model and core count stated
    h = d["host_io"]                                                                   # the with-H2D/D2H variant is in the default line
    assert h["value"] > 0 and h["steps"] >= 4 and h["value"] <= d["value"] * 1.05
    for t in d["top_launches"]:
        assert t["GBps"] <= 8000.0, "%s: %s GB/s is above the HBM peak — its byte model is wrong" % (t["name"], t["GBps"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert [l["threads"] for l in c["legs"]][:2] == [1, 2] and all(l["value"] > 0 for l in c["legs"])
    assert c["value"] == max(l["value"] for l in c["legs"]) and c["cores"] in [l["threads"] for l in c["legs"]] and c["host"]["model"]
    p = c["parity_sample"]
    assert p["mask_iou_min"] >= 0.999 and p["composite_max_abs_diff"] <= 1
    fb = d["full_batch_twin_streams"]                                                  # every stream of the batch was compared with its scene twin on the GPU
    assert fb["streams"] == 256 and fb["groups_compared_with_group_0"] == 15 and fb["all_identical"] is True


def test_roofline_denominators_name_the_pipe_the_kernel_issues_on():
    """DeepLab's fused expand + depthwise issues v_mfma_f32_16x16x32_f16 (3-term split) — priced against the f32 matrix peak it looked
    mfma-bound at 0.56; against its own pipe (2500 / 3 TFLOP/s useful, ridge 104 FLOP/B) a 38 FLOP/B launch is HBM-bound."""
    sys.path.insert(0, ROOT)
    import bench
    s = {"name": "conv#50+dw#51", "avg_ms": 1.059, "bytes": 2.46e9, "flops": 93.8e9, "GBps": 2.46e9 / 1.059e-3 / 1e9}
    r = bench.roofline_of(s, {}, "deeplabv3_257_mv_gpu.tflite")
    assert r["bound"] == "hbm" and abs(r["frac"] - 0.29) < 0.01 and r["ridge_flop_per_byte"] > 100 and r["flops_frac_of_pipe"] < 0.15
    assert bench.pipe_of("deeplabv3_257_mv_gpu.tflite", "conv#0+dw#1+conv#2")[1] == bench.FP32_PEAK_TFLOPS       # the stem kernel is f32 MFMA
    assert bench.pipe_of("segm_lite_v681.tflite", "seg_head")[1] == bench.FP32_PEAK_TFLOPS
    os.environ["BSX_F16_GEMM"] = "off"
    try:
        assert bench.pipe_of("deeplabv3_257_mv_gpu.tflite", "conv#60")[1] == bench.FP32_PEAK_TFLOPS
    finally:
        os.environ.pop("BSX_F16_GEMM")
    # an HBM-bound byte kernel with a stale / fresh PMC stamp
    pmc = {"round": "rXX", "csrc_digest": "0" * 16, "kernels": {"mask_tile_k<true>": {"FETCH_SIZE_KiB": 130430, "WRITE_SIZE_KiB": 304700, "kernel": "mask_tile_k<true>"}}}
    m = {"name": "mask_blend", "avg_ms": 0.1389, "bytes": 789577728.0, "flops": 0.0, "GBps": 789577728.0 / 0.1389e-3 / 1e9}
    r = bench.roofline_of(m, pmc, "segm_lite_v681.tflite")
    assert r["traffic_stale"] is True and 0.5 < r["frac_counted_traffic"] < r["frac"]
    pmc["csrc_digest"] = bench.csrc_digest()
    assert bench.roofline_of(m, pmc, "segm_lite_v681.tflite")["traffic_stale"] is False


def test_a_shared_background_never_prices_a_line_above_the_hbm_peak():
    """one background image shared by all streams is cache-resident: its reads are algorithmic bytes but not HBM bytes.  The stand-alone blend moved
    786 MB of algorithmic bytes in 86.6 us on the round-4 box = 9.1 TB/s "of 8": the line then reports its HBM side (7 of the 10 B/px) and keeps the other figure beside it."""
    sys.path.insert(0, ROOT)
    import bench
    b = {"name": "blend (standalone)", "avg_ms": 0.0866, "bytes": 786432000.0, "flops": 0.0, "GBps": 786432000.0 / 0.0866e-3 / 1e9, "shared_bytes": 256 * 3.0 * 640 * 480}
    r = bench.roofline_of(b, {}, "segm_lite_v681.tflite")
    assert r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved_incl_shared_background"] > 8000.0
    assert r["algorithmic_bytes_per_launch"] == int(786432000 * 0.7) and r["frac_hbm_side"] == r["frac"]
    m = {"name": "mask_blend", "avg_ms": 0.1062, "bytes": 670564352.0, "flops": 0.0, "GBps": 670564352.0 / 0.1062e-3 / 1e9, "shared_bytes": 256 * 3.0 * 640 * 480 * 0.91}
    r = bench.roofline_of(m, {}, "segm_lite_v681.tflite")
    assert r["frac"] < 1.0 and r["frac_hbm_side"] < r["frac"] and "achieved_incl_shared_background" not in r      # below the peak: both figures, the algorithmic one leads


def test_committed_pmc_file_is_stamped():
    pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert "csrc_digest" in pj and len(pj["csrc_digest"]) == 16
