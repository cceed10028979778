#!/bin/bash
# per-launch HIP-event profiles of one workload under several environments, in ONE gpurun call:
#   tools/launch_ab.sh <tag> "<bench args>" "<env 0>" "<env 1>" ...   -> gpurun_out/launches_<tag>_v<i>.csv
tag=$1; args=$2; shift 2
mkdir -p gpurun_out
i=0
for v in "$@"; do
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-line $args --launch-csv gpurun_out/launches_${tag}_v${i}.csv \
      > gpurun_out/launches_${tag}_v${i}.json 2> gpurun_out/launches_${tag}_v${i}.err || { echo "variant $i failed"; tail -5 gpurun_out/launches_${tag}_v${i}.err; exit 1; }
  echo "variant $i [$v] done"
  i=$((i+1))
done
