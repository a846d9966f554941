mkdir -p gpurun_out/r3 && cd tests && timeout -k 10 300 python -u -m pytest test_gpu_x3_training.py -x -q -s --timeout 120 --timeout-method thread > ../gpurun_out/r3/x3_t.log 2>&1; grep -E "^bf16x3|^fp32|passed|failed|Error|assert" ../gpurun_out/r3/x3_t.log | head -20; cd .. && timeout -k 10 200 python bench.py --train-precision bf16x3 --no-cpu-baseline > gpurun_out/r3/bench_x3.json 2> gpurun_out/r3/bench_x3.err; tail -3 gpurun_out/r3/bench_x3.err; python - <<EOF2
import json
d=json.loads(open("gpurun_out/r3/bench_x3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), round(v["tflops"],1))
for k,v in (d.get("kernels_serialized") or {}).items(): print("ser", k, round(v["ms_per_step"],3), round(v["tflops"],1))
EOF2
