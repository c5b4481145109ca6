#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for b in 32 64 128 256 1024; do
  timeout 300 python bench.py --model deeplab --batch $b --no-extra-configs --no-cpu-baseline --steps 10 --warmup 3 --ramp-seconds 1 --dump-launches gpurun_out/r03c_dl_launches_$b.txt > gpurun_out/r03c_dl_$b.json 2>/dev/null
done
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03c_pytest.txt 2>&1
tail -3 gpurun_out/r03c_pytest.txt
timeout 500 bash tools/profile_config.sh r03c mlkit_hd '{"batch":256,"width":1280,"height":720,"model":"selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite"}' --model mlkit --batch 256 --width 1280 --height 720
timeout 700 bash tools/profile_config.sh r03c deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring
timeout 700 bash tools/profile_config.sh r03c full_hd '{"batch":1024,"width":1280,"height":720,"model":"segm_full_v679.tflite"}' --model full --batch 1024 --width 1280 --height 720
timeout 400 bash tools/profile_config.sh r03c lite '{"batch":256,"width":640,"height":480,"model":"segm_lite_v681.tflite"}'
