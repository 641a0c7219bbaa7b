#!/bin/bash
# Same-box A/B of bench.py variants: scripts/ab_bench.sh OUTDIR "label1:ENV=.. ENV=.." "label2:..."   (each run twice)
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    [ "$envs" = "$spec" ] && envs=""
    env $envs python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $AB_ARGS > "$out/$label.$rep.json" 2> "$out/$label.$rep.err" || echo "FAILED $label"
    python - "$out/$label.$rep.json" "$label.$rep" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(f"{sys.argv[2]:28s} value {d['value']:8.1f}  sustained {d.get('sustained_1000_updates_per_sec') or 0:8.1f}  step_sum {r.get('step_sum_us')}  small {r.get('small_launches_us')}", flush=True)
if sys.argv[2].endswith(".2"):
    print("   ", {k.split(':')[0]: v for k, v in (r.get('step_launches_us') or {}).items()})
PY
  done
done
