#!/bin/bash
# Same-box A/B of bench.py under environment switches: scripts/experiments/ab_env.sh OUT "VAR=val ..." "VAR=val ..." ... (each argument one arm; "-" = no switch)
out=$1; shift
: > "$out"
for round in 1 2; do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
    line=$(env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$round [$arm] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d.get("stage_ms_per_step",{}).get("nms"))')" >> "$out"
  done
done
