set -e
O=gpurun_out/r3_ws5.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_bench_config.py -x -q >> $O 2>&1 || { tail -40 $O; exit 1; }
tail -3 $O
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --launch-csv gpurun_out/r3_launches_f32.csv > gpurun_out/r3_b_f32_ws.json 2>> $O
RFI_NO_WS=1 timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r3_b_f32_nows.json 2>> $O
python - <<'PY'
import json
for f in ("gpurun_out/r3_b_f32_ws.json","gpurun_out/r3_b_f32_nows.json"):
    j=json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["roofline"]["tflops"], j["roofline"]["frac"], {k:v["ms_per_step"] for k,v in j["families"].items()})
PY
