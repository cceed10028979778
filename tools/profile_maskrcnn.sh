set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4m
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
T="timeout -k 10"
$T 400 $B --workload maskrcnn > $O/bench_maskrcnn.json 2> $O/bench.err
$T 300 $B --workload resnet1024 --no-cpu-baseline > $O/bench_resnet1024.json 2>> $O/bench.err
$T 300 $B --workload cnn3 --no-cpu-baseline > $O/bench_cnn3.json 2>> $O/bench.err
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial_maskrcnn -- $B --no-cpu-baseline --profile-steps 0 --steps 10 --warmup 3 --workload maskrcnn > $O/stats_serial_maskrcnn.log 2>&1
Q="--no-cpu-baseline --profile-steps 0 --steps 5 --warmup 2 --workload maskrcnn"
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_maskrcnn -- $B $Q > $O/pmc_fetch.log 2>&1
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_maskrcnn -- $B $Q > $O/pmc_write.log 2>&1
RFI_NO_OVERLAP=1 $T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma_maskrcnn -- $B $Q > $O/pmc_mfma.log 2>&1
python3 $R/tools/summarize_pmc.py $O/pmc_fetch_maskrcnn $O/pmc_write_maskrcnn $O/traffic_maskrcnn.json > /dev/null
python3 $R/tools/summarize_mfma.py $O/pmc_mfma_maskrcnn $O/mfma_util_maskrcnn.json > /dev/null
rm -rf $O/pmc_fetch_maskrcnn $O/pmc_write_maskrcnn $O/pmc_mfma_maskrcnn
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
