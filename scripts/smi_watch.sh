#!/bin/bash
# Sample sclk / power while a bench variant runs:  smi_watch.sh <out-file> <bench args...>
out=$1; shift
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power \(W\)|Average|Socket" | tr '\n' ' '; echo; sleep 0.25; done ) > $out.smi &
W=$!
PORL_BENCH_SUSTAINED=0 python bench.py --steps 20000 --warmup 50 --no-cpu-baseline --no-roofline "$@" > $out.json 2> $out.err
kill $W 2>/dev/null
python -c "import json; d=json.loads(open('$out.json').read().strip().splitlines()[-1]); print('$out', round(d['value'],1))"
sed -n '8,14p' $out.smi
