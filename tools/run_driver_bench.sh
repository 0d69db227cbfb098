for H in 0 1; do
  if [ $H = 1 ]; then export P2_NO_HELD_CUS=1; fi
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_driver$H.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_driver$H.log") if l.startswith("{")][-1])
print("NO_HELD=$H", d["value"], {k:v["ms_per_step"] for k,v in d["single_pass"].items() if isinstance(v,dict)}, d["in_flight"]["cell"]["ms_per_step"], d["in_flight"]["operator_api"]["ms_per_step"], d["roofline"]["forward"]["ms"], d["roofline"]["backward"]["ms"])
PY
done
