#!/bin/bash
# gauss_blur_k: delayed coefficient sets (no alignbyte), padded hbuf columns, 4-pixel staging for every radius and for y-border tiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gaussian or own_blur or blur_own" 2>&1 | tail -4
sed -n '/^cat > \/tmp\/gt.py/,/^PY$/p' tools/calls/r03_call38.sh | sed '1d;$d' > /tmp/gt.py
TAG="words+stage4 " timeout 200 python /tmp/gt.py 2>&1 | tail -2
TAG="bytes+stage4 " BSX_GAUSS_BYTE_STORE=1 timeout 200 python /tmp/gt.py 2>&1 | tail -2
TAG="words+bytestg" BSX_GAUSS_BYTE_STAGE=1 timeout 200 python /tmp/gt.py 2>&1 | tail -2
