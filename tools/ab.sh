#!/bin/bash
# A/B runs of the headline bench inside ONE gpurun call (boxes of the pool differ by up to 5 %):
#   tools/ab.sh <tag> <rounds> "<env of variant 0>" "<env of variant 1>" ...
# prints ms/step of every run; variants are interleaved round by round
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out
n=$#
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    out=gpurun_out/ab_${tag}_v${i}_r${r}.json
    env $v python bench.py --steps ${AB_STEPS:-100} --warmup 20 --no-cpu-baseline --no-bf16-line --profile-steps 0 ${AB_ARGS} > $out 2> gpurun_out/ab_${tag}_v${i}_r${r}.err || { echo "variant $i failed"; tail -5 gpurun_out/ab_${tag}_v${i}_r${r}.err; exit 1; }
    python - "$out" "$i" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"variant {sys.argv[2]} [{sys.argv[3]}]: {d['ms_per_step']:.4f} ms/step  {d['value']:.1f} {d['unit']}", flush=True)
PY
    i=$((i+1))
  done
done
