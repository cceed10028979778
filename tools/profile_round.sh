#!/bin/bash
# Everything under profiles/<TAG>_* comes from this script, run on the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r2'
# then `python tools/collect_profiles.py r2` here folds gpurun_out/<TAG>/ into profiles/.
# Bench lines of every workload / arithmetic, rocprofv3 kernel statistics (serial and with the weight-gradient
# overlap), and the PMC passes (HBM traffic, matrix-pipe busy cycles) of the default and the bf16 command -- counters
# in their own runs, --kernel-trace only.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
T="timeout -k 10"
$T 400 $B > "$O/bench.json" 2> "$O/bench.err"                      # the driver's command: bf16 line + its float32 companion
$T 300 $B --dtype f32 --no-cpu-baseline > "$O/bench_f32.json" 2>> "$O/bench.err"
$T 300 $B --dtype f32mfma --no-cpu-baseline > "$O/bench_f32mfma.json" 2>> "$O/bench.err"
$T 300 $B --workload cnn3 > "$O/bench_cnn3.json" 2>> "$O/bench.err"
$T 400 $B --workload unet1024 > "$O/bench_unet1024.json" 2>> "$O/bench.err"
$T 400 $B --workload resnet1024 > "$O/bench_resnet1024.json" 2>> "$O/bench.err"
$T 300 $B --workload resnet --no-cpu-baseline > "$O/bench_resnet.json" 2>> "$O/bench.err"
echo "bench lines done"
P="--no-cpu-baseline --profile-steps 0 --steps 10 --warmup 3"
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_serial" -- $B $P --dtype f32 > "$O/stats_serial.log" 2>&1
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_overlap" -- $B $P --dtype f32 > "$O/stats_overlap.log" 2>&1
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_serial_bf16" -- $B $P --dtype bf16 > "$O/stats_serial_bf16.log" 2>&1
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_overlap_bf16" -- $B $P --dtype bf16 > "$O/stats_overlap_bf16.log" 2>&1
echo "kernel statistics done"
Q="--no-cpu-baseline --profile-steps 0 --steps 5 --warmup 2"
for dt in f32 bf16; do
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$dt" -- $B $Q --dtype $dt > "$O/pmc_fetch_$dt.log" 2>&1
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$dt" -- $B $Q --dtype $dt > "$O/pmc_write_$dt.log" 2>&1
  RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_mfma_$dt" -- $B $Q --dtype $dt > "$O/pmc_mfma_$dt.log" 2>&1
  echo "pmc $dt done"
done
# keep what travels back small: the per-dispatch traces are large
find "$O" -name "*kernel_trace.csv" -size +20M -delete
du -sh "$O"
