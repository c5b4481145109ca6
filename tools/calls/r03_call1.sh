#!/bin/bash
# round 3, GPU call 1: diagnostics for the frame program + full GPU test suite + HBM-traffic counters of configs[2..4]
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out



timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03a_pytest.txt 2>&1
tail -5 gpurun_out/r03a_pytest.txt
timeout 500 bash tools/profile_config.sh r03a mlkit_hd '{"batch":256,"width":1280,"height":720,"model":"selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite"}' --model mlkit --batch 256 --width 1280 --height 720
timeout 700 bash tools/profile_config.sh r03a deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring
timeout 700 bash tools/profile_config.sh r03a full_hd '{"batch":1024,"width":1280,"height":720,"model":"segm_full_v679.tflite"}' --model full --batch 1024 --width 1280 --height 720
timeout 400 bash tools/profile_config.sh r03a lite '{"batch":256,"width":640,"height":480,"model":"segm_lite_v681.tflite"}'
ls gpurun_out | head -50
