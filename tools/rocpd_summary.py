"""Summarise a rocprofv3 (rocpd SQLite) result into the text files kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_kernel_stats.md
    python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch/bench_results.db gpurun_out/pmc_write/bench_results.db

Kernel stats = what `rocprofv3 --kernel-trace --stats` tabulates (calls, total, average,
share) — this ROCm build writes them into the database instead of CSV files.
PMC mode prints per-kernel average counter values (per dispatch).
"""
import sqlite3
import sys


def short(name):
    name = name.replace("bsx::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx in rows:
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (short(n), c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))


def pmc(paths):
    print("| kernel | counter | dispatches | avg value per dispatch | max |")
    print("|---|---|---:|---:|---:|")
    for path in paths:
        db = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, count(*), avg(v), max(v) from (select kernel_name, counter_name, dispatch_id, sum(value) v "
             "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name order by avg(v) desc")
        for n, cn, c, v, mx in db.execute(q).fetchall():
            print("| %s | %s | %d | %.1f | %.1f |" % (short(n), cn, c, v, mx))


def pmc_json(tag, paths):
    """FETCH_SIZE / WRITE_SIZE (KiB, average per dispatch) per kernel → the JSON bench.py reads for roofline.traffic."""
    import json
    kernels = {}
    for path in paths:
        db = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, avg(v) from (select kernel_name, counter_name, dispatch_id, sum(value) v "
             "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
        for n, cn, v in db.execute(q).fetchall():
            if cn in ("FETCH_SIZE", "WRITE_SIZE") and "bsx::" in n:
                kernels.setdefault(short(n), {})[cn + "_KiB"] = round(v, 1)
    print(json.dumps({"round": tag, "workload": {"batch": 256, "width": 640, "height": 480, "model": "segm_lite_v681.tflite"},
                      "note": "avg per dispatch; traffic bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section",
                      "kernels": kernels}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--pmc-json":
        pmc_json(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "--pmc":
        pmc(sys.argv[2:])
    else:
        kernel_stats(sys.argv[1])
