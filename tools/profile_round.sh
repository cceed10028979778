#!/bin/bash
# Everything under profiles/<TAG>_* comes from this script, run on the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r3'           # the headline workload (float32 + bf16 companion)
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r3 extra'     # cnn3, unet1024, resnet1024, maskrcnn: stats + PMC passes
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r3 extra "maskrcnn"'     # ... of the named workloads only
# then `python tools/collect_profiles.py r3` here folds gpurun_out/<TAG>/ into profiles/.
# Bench lines of every workload / arithmetic, rocprofv3 kernel statistics (serial and with the weight-gradient
# overlap), and the PMC passes (HBM traffic, matrix-pipe busy cycles) -- counters in their own runs, --kernel-trace
# only, the program itself directly after `--`.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3}
PART=${2:-main}
WLS=${3:-cnn3 unet1024 resnet1024 maskrcnn}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
T="timeout -k 10"
P="--no-cpu-baseline --profile-steps 0 --steps 10 --warmup 3"
Q="--no-cpu-baseline --profile-steps 0 --steps 5 --warmup 2"
pmc3() {   # $1 = output tag, rest = bench arguments: FETCH / WRITE / matrix-pipe passes of the serial command
  local tag=$1; shift
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$tag" -- $B $Q "$@" > "$O/pmc_fetch_$tag.log" 2>&1
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$tag" -- $B $Q "$@" > "$O/pmc_write_$tag.log" 2>&1
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_mfma_$tag" -- $B $Q "$@" > "$O/pmc_mfma_$tag.log" 2>&1
  # the per-dispatch counter tables are tens of megabytes each and gpurun_out/ travels back only up to 64 MiB: summarise
  # here (HBM bytes per launch with the guide's gfx950 corrections; matrix-pipe busy share), keep the summaries only
  python3 $R/tools/summarize_pmc.py "$O/pmc_fetch_$tag" "$O/pmc_write_$tag" "$O/traffic_$tag.json" > /dev/null
  python3 $R/tools/summarize_mfma.py "$O/pmc_mfma_$tag" "$O/mfma_util_$tag.json" > /dev/null
  rm -rf "$O/pmc_fetch_$tag" "$O/pmc_write_$tag" "$O/pmc_mfma_$tag"
  echo "pmc $tag done"
}
if [ "$PART" = main ]; then
  $T 400 $B > "$O/bench.json" 2> "$O/bench.err"                      # the driver's command: float32 line + its bf16 companion
  $T 300 $B --dtype bf16 --no-cpu-baseline > "$O/bench_bf16.json" 2>> "$O/bench.err"
  $T 300 $B --dtype f32mfma --no-cpu-baseline > "$O/bench_f32mfma.json" 2>> "$O/bench.err"
  $T 300 $B --workload resnet --no-cpu-baseline > "$O/bench_resnet.json" 2>> "$O/bench.err"
  echo "bench lines done"
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_serial" -- $B $P --dtype f32 > "$O/stats_serial.log" 2>&1
  $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_overlap" -- $B $P --dtype f32 > "$O/stats_overlap.log" 2>&1
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_serial_bf16" -- $B $P --dtype bf16 > "$O/stats_serial_bf16.log" 2>&1
  $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_overlap_bf16" -- $B $P --dtype bf16 > "$O/stats_overlap_bf16.log" 2>&1
  echo "kernel statistics done"
  pmc3 f32 --dtype f32
  pmc3 bf16 --dtype bf16
else
  for wl in $WLS; do
    $T 400 $B --workload $wl > "$O/bench_$wl.json" 2>> "$O/bench.err"
    RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_serial_$wl" -- $B $P --workload $wl > "$O/stats_serial_$wl.log" 2>&1
    pmc3 $wl --workload $wl
  done
fi
# keep what travels back small: the per-dispatch traces are large (the statistics tables next to them stay)
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
du -sh "$O"
