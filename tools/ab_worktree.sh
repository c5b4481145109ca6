#!/bin/bash
# Same-box A/B of two builds through ONE gpurun call (the pool's boxes differ by up to ±10 %, so numbers from two calls do not compare).
#   bash tools/ab_worktree.sh [commit]      (here, on CPU)  — checks <commit> (default HEAD) out as a git worktree under _ab_old/ and builds it there:
#                                            libbsx.so + the hipRTC kernel cache; _ab_old/ is excluded locally (.git/info/exclude) but NOT gpurun-ignored, so it travels with the snapshot
#   then, in the gpurun script:            run() { cd $1; python bench.py --no-extra-configs --no-cpu-baseline ... ; }
#                                            run $ROOT/_ab_old; run $ROOT; run $ROOT/_ab_old; run $ROOT      (alternate: clocks drift inside a call too)
#   when done:                             git worktree remove --force _ab_old
# Used from profiles/r03ai on (tools/calls/r03_call42.sh … r03_call49.sh).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=${1:-HEAD}
cd "$R"
H=$(git rev-parse "$C")
grep -qx "_ab_old/" .git/info/exclude 2>/dev/null || echo "_ab_old/" >> .git/info/exclude
if [ -d _ab_old ]; then (cd _ab_old && git checkout -q --detach "$H"); else git worktree add -f _ab_old "$H" -q; fi
[ -e _ab_old/models ] || ln -s ../models _ab_old/models
mkdir -p _ab_old/tests/golden; [ -e _ab_old/tests/golden/models ] || { [ -d tests/golden/models ] && ln -s ../../../tests/golden/models _ab_old/tests/golden/models; }
(cd _ab_old && python -c "import __graft_entry__ as g; g.build()" > /tmp/ab_build.log 2>&1) || { tail -20 /tmp/ab_build.log; exit 1; }
echo "_ab_old = $H built ($(ls _ab_old/backscrub_amd/kcache | wc -l) cached RTC kernels)"
