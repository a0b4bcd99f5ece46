# A/B helper: bench config 3 three times, print value (and the host timeline of the last run)
B="python bench.py --steps 400 --warmup 20 --cpu-iters 0 --fit-iters 0 --no-other-configs --no-kernel-timing --long-seconds 0"
for i in 1 2 3; do $B 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; done
MYFM_AMD_HOST_TIMELINE=1 $B 2>&1 | grep "host timeline" | tail -1
