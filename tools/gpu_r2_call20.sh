#!/bin/bash
mkdir -p gpurun_out/r2f
python tools/experiments/ab_rollout.py box-close-v3 peg-unplug-side-v3 stick-pull-v3 hammer-v3 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 900 python tools/measure_caps_gpu.py 4096 1500 > gpurun_out/r2f/caps.log 2>&1; tail -40 gpurun_out/r2f/caps.log
python - <<'PY'
import json
m = json.load(open("gpurun_out/model_caps_measured.json"))
json.dump({k: {"maxcon": v["maxcon"], "maxefc": v["maxefc"]} for k, v in m.items()}, open("metaworld_amd/data/model_caps.json", "w"), indent=1)
PY
python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest_gpu.log
tail -15 gpurun_out/r2f/pytest_gpu.log
python bench.py > gpurun_out/r2f/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2f/bench_default.log
tail -3 gpurun_out/r2f/bench_default.log | cut -c1-3500
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-precision > gpurun_out/r2f/bench_short.log 2>&1
tail -1 gpurun_out/r2f/bench_short.log | cut -c1-400
