#!/bin/bash
out=$1; shift
: > "$out"
for arm in "$@"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  line=$(env $envs python bench.py --steps 20 --warmup 5 --ingest-host --no-cpu-baseline 2>/dev/null | tail -1)
  echo "[$arm] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],2), round(d["pcie"]["achieved"],1), d["stage_ms_per_step"]["wall"])')" >> "$out"
done
