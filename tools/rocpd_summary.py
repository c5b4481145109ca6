"""Summarise a rocprofv3 (rocpd SQLite) result into the text files kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_kernel_stats.md
    python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch/bench_results.db gpurun_out/pmc_write/bench_results.db

Kernel stats = what `rocprofv3 --kernel-trace --stats` tabulates (calls, total, average,
share) — this ROCm build writes them into the database instead of CSV files.
PMC mode prints per-kernel average counter values (per dispatch).
"""
import sqlite3
import sys


_DEMANGLED = {}


def demangle(name):
    """rocprofv3 leaves long template instantiations mangled in the counter tables: c++filt them (cached)."""
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLED:
        import subprocess
        try:
            sym = name[:-3] if name.endswith(".kd") else name          # kernel-descriptor symbol: the function's name + ".kd"
            _DEMANGLED[name] = subprocess.run(["c++filt", sym], capture_output=True, text=True, timeout=10).stdout.strip() or name
        except Exception:
            _DEMANGLED[name] = name
    return _DEMANGLED[name]


def demangle_ours(name):
    """`_ZN3bsx12_GLOBAL__N_1<len><kernel>I<Li<int>E | Lb<0|1>E ...>E...` → `kernel<3, 1, false>`: binutils' c++filt gives up on signatures that
    contain _Float16 (DF16_), which is every split-f16 kernel of kernels_nn.hip."""
    import re
    m = re.match(r"_ZN3bsx12_GLOBAL__N_1(\d+)", name)
    if not m:
        return None
    n = int(m.group(1))
    rest = name[m.end():]
    kern, rest = rest[:n], rest[n:]
    if not rest.startswith("I"):
        return kern
    args, rest = [], rest[1:]
    while rest and not rest.startswith("E"):
        a = re.match(r"L([ib])(n?\d+)E", rest)
        if not a:
            return kern + "<...>"
        v = a.group(2).replace("n", "-")
        args.append(("true" if v != "0" else "false") if a.group(1) == "b" else v)
        rest = rest[a.end():]
    return kern + "<" + ", ".join(args) + ">"


def is_ours(name):
    return "bsx::" in name or name.startswith("_ZN3bsx") or name.startswith("bsx_")      # bsx_mid: the hipRTC-compiled specialised program


def short(name):
    own = demangle_ours(name)
    if own:
        return own[:60]
    name = demangle(name).replace("bsx::(anonymous namespace)::", "").replace("void ", "")
    if name.endswith(".kd"):
        name = name[:-3]
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
            if cn in ("FETCH_SIZE", "WRITE_SIZE") and is_ours(n):
                kernels.setdefault(short(n), {})[cn + "_KiB"] = round(v, 1)
    print(json.dumps({"round": tag, "workload": {"batch": 256, "width": 640, "height": 480, "model": "segm_lite_v681.tflite"},
                      "note": "avg per dispatch; traffic bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section",
                      "kernels": kernels}, indent=1))


def step_sequence(path, counter, first=("prep_fused_k", "prep_resize_k")):
    """Dispatches of ONE bench step in launch order: the kernels between two consecutive dispatches of the step's first kernel
    (the last complete step of the run) → [(kernel, counter value)]."""
    db = sqlite3.connect(path)
    q = ("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id, kernel_name order by dispatch_id")
    rows = [(short(n), v) for _, n, v in db.execute(q, (counter,)).fetchall() if is_ours(n)]
    starts = [i for i, (n, _) in enumerate(rows) if n.startswith(first)]      # the step's first kernel: the fused prep (or, with BSX_PREP_SPLIT=1, the resize)
    if len(starts) < 6:
        return []
    seq = rows[starts[3]:starts[4]]        # the second timed step of `bench.py --warmup 2` (the per-launch profile loop comes after the timed steps)
    while seq and seq[-1][0].startswith("resize_bgr_k"):      # --bg-ring: the NEXT step's background resize precedes its prep_resize_k
        seq.pop()
    return seq


def pmc_json2(tag, desc, fetch_db, write_db):
    """One workload entry of profiles/pmc_latest.json: per-kernel averages AND the per-launch sequence of one step (so that layers
    which share a kernel instantiation, e.g. DeepLab's GEMMs, still get their own traffic figure)."""
    import json
    kernels = {}
    for path in (fetch_db, write_db):
        db = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, avg(v) from (select kernel_name, counter_name, dispatch_id, sum(value) v "
             "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
        for n, cn, v in db.execute(q).fetchall():
            if cn in ("FETCH_SIZE", "WRITE_SIZE") and is_ours(n):
                kernels.setdefault(short(n), {})[cn + "_KiB"] = round(v, 1)
    f, w = step_sequence(fetch_db, "FETCH_SIZE"), step_sequence(write_db, "WRITE_SIZE")
    seq = []
    if f and len(f) == len(w) and all(a[0] == b[0] for a, b in zip(f, w)):
        seq = [{"kernel": a[0], "FETCH_SIZE_KiB": round(a[1], 1), "WRITE_SIZE_KiB": round(b[1], 1)} for a, b in zip(f, w)]
    wl = json.loads(desc)
    print(json.dumps({"round": tag, "workload": wl, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; traffic bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                      "(FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section)", "kernels": kernels, "step_launches": seq}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--pmc-json2":
        pmc_json2(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "--pmc-json":
        pmc_json(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "--pmc":
        pmc(sys.argv[2:])
    else:
        kernel_stats(sys.argv[1])
