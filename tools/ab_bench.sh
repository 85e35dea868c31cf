#!/bin/bash
# A/B of library variants INSIDE ONE gpurun call (boxes of the pool differ by up to 25 % on this kernel, so numbers of different calls
# cannot be compared): the variants are benchmarked interleaved, ROUNDS times each, and the values are printed per variant.
#   build:  tools/ab_bench.sh build NAME [extra hipcc flags...]      -> metaworld_amd/libmwgpu_v_NAME.so from the working tree
#   run:    gpurun -- bash tools/ab_bench.sh run [ROUNDS=2] libmwgpu.so libmwgpu_v_NAME.so ...   (bench args via AB_ARGS)
set -u
cd "$(dirname "$0")/.."
cmd=${1:-run}; shift || true
if [ "$cmd" = build ]; then
  name=$1; shift
  ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form "$@" -o metaworld_amd/libmwgpu_v_$name.so metaworld_amd/csrc/mwgpu.hip && ls -la metaworld_amd/libmwgpu_v_$name.so
  exit $?
fi
rounds=2
case "${1:-}" in ''|*[!0-9]*) ;; *) rounds=$1; shift;; esac
O=gpurun_out/ab_$(date +%H%M%S); mkdir -p $O
args=${AB_ARGS:---no-cpu-baseline --no-extra-precision --steps 300}
for r in $(seq $rounds); do
  for v in "$@"; do
    MW_LIB=$v timeout 300 python bench.py $args >> $O/$v.txt 2>&1
  done
done
for v in "$@"; do
  echo "$v: $(grep -h -o '"value": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | tr '\n' ' ')  flags: $(grep -h -o '"flags": [0-9]*' $O/$v.txt | cut -d' ' -f2 | sort -u | tr '\n' ' ')"
done
