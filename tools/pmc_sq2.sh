#!/bin/bash
# Wave-state + instruction-mix counters for every kernel of one bench step (two passes, --kernel-trace only; see the gpurun rules).
# usage (on the GPU box): bash tools/pmc_sq2.sh <tag> [bench args]  → gpurun_out/pmc_sq2_<tag>.md
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-side-probes --profile-iters 1 --ramp-seconds 0 $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace \
  -d $R/gpurun_out/pmc_sq2a_$TAG -o b -- $B > $R/gpurun_out/pmc_sq2_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace \
  -d $R/gpurun_out/pmc_sq2b_$TAG -o b -- $B >> $R/gpurun_out/pmc_sq2_$TAG.log 2>&1
python $R/tools/rocpd_summary.py --pmc $R/gpurun_out/pmc_sq2a_$TAG/b_results.db $R/gpurun_out/pmc_sq2b_$TAG/b_results.db | grep -v "at::\|elementwise\|fill_k" > $R/gpurun_out/pmc_sq2_$TAG.md
python - <<PY >> $R/gpurun_out/pmc_sq2_$TAG.md
import sqlite3
db = sqlite3.connect("$R/gpurun_out/pmc_sq2a_$TAG/b_results.db")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("\nkernels columns:", cols)
want = [c for c in cols if any(k in c.lower() for k in ("name", "lds", "vgpr", "sgpr", "scratch", "workgroup", "grid", "duration"))]
for r in db.execute("select %s from kernels group by name" % ",".join(want)): print(dict(zip(want, r)))
PY
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_[A-Z_]*OCCUP[A-Z_]*\|SQ_LEVEL[A-Z_]*" | sort -u >> $R/gpurun_out/pmc_sq2_$TAG.md
