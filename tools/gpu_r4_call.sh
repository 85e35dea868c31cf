set -u
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -n 4 gpurun_out/c1/pytest_gpu.txt
bash tools/gpu_ab_ref.sh 2 2>&1 | tail -n 3
MW_LIB=libmwgpu_v_colltiming.so timeout 300 python tools/experiments/coll_timing.py 100 > gpurun_out/c1/coll_timing.txt 2>&1; head -n 12 gpurun_out/c1/coll_timing.txt | cut -c1-400
MW_LIB=libmwgpu_v_timing.so timeout 300 python tools/experiments/tail_probe.py 16 > gpurun_out/c1/tail_probe.txt 2>&1; tail -n 6 gpurun_out/c1/tail_probe.txt | cut -c1-300
