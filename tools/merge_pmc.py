"""Merge the per-workload PMC summaries of one profiling round (tools/profile_config.sh → gpurun_out/pmc_<tag>_<name>.json) into
profiles/pmc_latest.json — the committed file bench.py reads `roofline.traffic` from (load_pmc).

    python tools/merge_pmc.py r03z lite mlkit_hd deeplab full_hd
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    tag, names = sys.argv[1], sys.argv[2:]
    wl = []
    for n in names:
        p = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s.json" % (tag, n))
        e = json.load(open(p))
        if not e.get("kernels"):
            raise SystemExit("no counters in " + p)
        wl.append(e)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of `bench.py --steps 5 --warmup 2` per workload, round %s; "
                   "traffic bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section); produced by "
                   "tools/profile_config.sh + tools/merge_pmc.py" % tag, "workloads": wl,
           # what ties the passes to the kernels: bench.py compares this with the tree it runs from and prints traffic_stale when they differ
           "csrc_digest": bench.csrc_digest(), "csrc_digest_note": "sha256[:16] over backscrub_amd/csrc/*.{hip,cpp,hpp} (bench.csrc_digest) of the tree the passes ran on"}
    for dst in (os.path.join(ROOT, "profiles", "pmc_latest.json"), os.path.join(ROOT, "gpurun_out", "pmc_latest.json")):
        with open(dst, "w") as f:
            json.dump(out, f, indent=1)
    print("merged", len(wl), "workloads")


if __name__ == "__main__":
    main()
