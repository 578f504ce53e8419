#!/bin/bash
# A/B of an environment switch on one GPU box (boxes differ by +-3 %: only runs of one session compare):
#   tools/ab_env.sh OUT.jsonl "VAR=a" "VAR=b" ...      three alternating rounds of the bench.py line; stderr -> OUT.err
set -eu -o pipefail
OUT=$1; shift
: > "$OUT"
for round in 1 2 3; do
  for kv in "$@"; do
    line=$(env $kv timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 2>>"${OUT%.jsonl}.err" | tail -1)
    echo "{\"env\": \"$kv\", \"round\": $round, \"line\": $line}" >> "$OUT"
    python3 - "$kv" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline", {})
one = d.get("one_step_at_a_time", {}).get("value")
print(f"{sys.argv[1]:40s} {d['value']:9.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  (one step at a time: {one})  conv {r.get('achieved')} TFLOP/s", flush=True)
PY
  done
done
