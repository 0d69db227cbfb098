timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_t3.log 2>&1; tail -2 gpurun_out/r2_gpu_t3.log
for V in 1 0; do
  P2_CELL_TABLE3=$V timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_t3_$V.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2_t3_$V.log") if l.startswith("{")][-1])
print("TABLE3=$V", d["value"], {k:v["ms_per_step"] for k,v in d["single_pass"].items() if isinstance(v,dict)}, d["in_flight"]["cell"]["ms_per_step"], d["roofline"]["forward"]["ms"], d["roofline"]["backward"]["ms"], d["roofline"]["frac"])
PY
done
