set -e -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_bf16.py tests/test_gpu_unet.py tests/test_gpu_resnet_unet.py tests/test_gpu_reproducible.py -x -q 2>&1 | tail -6
for s in "64 128 128 32 32" "64 64 64 64 64" "64 32 32 128 128" "64 8 8 512 512" "64 128 128 64 32"; do
  echo "== $s wgrad ws (impl 4) / split (RFI_NO_WGRAD_WS)"
  timeout -k 10 120 python tools/bench_conv.py $s wgrad 20 4
  RFI_NO_WGRAD_WS=1 timeout -k 10 120 python tools/bench_conv.py $s wgrad 20 4
done
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('f32', j['value'], j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
RFI_NO_WGRAD_WS=1 timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('f32 no wgrad_ws', j['value'], j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
timeout -k 10 400 python bench.py --workload resnet1024 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('resnet1024', j['value'], j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
