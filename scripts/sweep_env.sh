#!/bin/bash
# usage: scripts/sweep_env.sh "VAR1=a VAR2=b" "VAR1=c" ...   -- one short bench.py run per environment, one summary line each
for envs in "$@"; do
  out=$(env $envs python bench.py ${BENCH_ARGS:---steps 20 --warmup 3} --cpu-iters 0 --fit-iters 0 2>/dev/null | tail -1)
  python - "$envs" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    r = d["roofline"]
    print("%-40s it/s %7.2f  ms/step %6.3f  %s avg %7.2f us  frac %.3f  by_kernel %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"],
        {k: v for k, v in list(r["by_kernel_ms_per_step"].items())[:5]}))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, sys.argv[2][-300:])
PY
done
