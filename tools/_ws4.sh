set -e
O=gpurun_out/r3_ws4.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3" >> $O 2>&1
for s in "64 128 128 32 32" "64 64 64 64 64" "64 32 32 128 128" "64 16 16 256 256" "64 8 8 512 512" "64 16 16 512 256" "64 128 128 64 32" "64 128 128 32 64"; do
  echo "== $s diag" >> $O
  RFI_HIP_LIB=build/librfi_diag.so timeout -k 10 120 python tools/bench_conv.py $s fwd 1 7 2>&1 | grep stamps | tail -1 >> $O
  echo "== $s ws impl7" >> $O
  timeout -k 10 120 python tools/bench_conv.py $s fwd 20 7 >> $O 2>&1
  echo "== $s old impl4" >> $O
  timeout -k 10 120 python tools/bench_conv.py $s fwd 20 4 >> $O 2>&1
done
cat $O
