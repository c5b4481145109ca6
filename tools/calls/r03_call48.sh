#!/bin/bash
# kernel stats + HBM PMC of the bgblur step, two-call vs one-pass (verdict item 7: one fewer full-frame pass, measured)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
for form in two one; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bgblur_$form -o p -- python $R/tools/profile_bgblur.py $form 12 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf_bgblur_$form -o p -- python $R/tools/profile_bgblur.py $form 4 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw_bgblur_$form -o p -- python $R/tools/profile_bgblur.py $form 4 > /dev/null 2>&1
done
cd $R
for form in two one; do
  echo "== $form =="; python tools/rocpd_summary.py gpurun_out/prof_bgblur_$form/p_results.db | grep -v "at::\|rocclr" | head -14 | tee gpurun_out/r03am_bgblur_${form}_kernel_stats.md
  python tools/rocpd_summary.py --pmc gpurun_out/pmcf_bgblur_$form/p_results.db gpurun_out/pmcw_bgblur_$form/p_results.db | grep -v "at::\|rocclr" | head -14 | tee gpurun_out/r03am_bgblur_${form}_pmc_hbm.md
done
