#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration per access width (run ON the GPU box through gpurun): tools/microbench_fetch.hip under two --pmc passes
# → gpurun_out/<tag>_fetch_calibration.md  (known bytes next to what the counters report; factor = known / reported)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
cd $R && [ -x tools/microbench_fetch ] || hipcc --offload-arch=gfx950 -O3 -o tools/microbench_fetch tools/microbench_fetch.hip
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/microbench_fetch > $R/gpurun_out/${TAG}_fetch_plain.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/cal_fetch -o cal -- $R/tools/microbench_fetch > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/cal_write -o cal -- $R/tools/microbench_fetch > /dev/null 2>&1
cd $R
TAG=$TAG python - <<'PY' > gpurun_out/${TAG}_fetch_calibration.md
import re, sqlite3, glob, os
known = {}
for l in open("gpurun_out/%s_fetch_plain.txt" % os.environ["TAG"]):
    m = re.match(r"(\S+)\s+(\d+) B/lane\s+known KiB\s+([\d.]+)\s+([\d.]+) ms\s+([\d.]+) GB/s", l)
    if m:
        known[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
def table(dbdir, counter):
    out = {}
    for db in glob.glob(dbdir + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        q = ("select kernel_name, avg(v), count(*) from (select kernel_name, dispatch_id, sum(value) v from counters_collection where counter_name = ? "
             "group by kernel_name, dispatch_id) group by kernel_name")
        for n, v, k in c.execute(q, (counter,)).fetchall():
            n = re.sub(r"\(.*", "", n).replace(".kd", "")
            out[n] = (v, k)
    return out
f, w = table("gpurun_out/cal_fetch", "FETCH_SIZE"), table("gpurun_out/cal_write", "WRITE_SIZE")
print("FETCH_SIZE / WRITE_SIZE calibration by access width (tools/microbench_fetch.hip; 1.5 GiB streamed once per kernel, 2 dispatches averaged)\n")
print("| kernel | B/lane | known KiB | GB/s (plain run) | FETCH_SIZE KiB | known / FETCH | WRITE_SIZE KiB | known / WRITE |")
print("|---|---:|---:|---:|---:|---:|---:|---:|")
for name, (bl, kib, gbs) in known.items():
    fv = next((v for n, (v, _) in f.items() if name in n), None)
    wv = next((v for n, (v, _) in w.items() if name in n), None)
    rd = name.startswith("read")
    print("| %s | %d | %.0f | %.0f | %s | %s | %s | %s |" % (name, bl, kib, gbs, "%.0f" % fv if fv is not None else "-", ("%.3f" % (kib / fv)) if (fv and rd) else "-",
                                                          "%.0f" % wv if wv is not None else "-", ("%.3f" % (kib / wv)) if (wv and not rd) else "-"))
PY
rm -rf gpurun_out/cal_fetch gpurun_out/cal_write
cat gpurun_out/${TAG}_fetch_calibration.md
