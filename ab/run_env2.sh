#!/bin/bash
# A/B of whole-step time between settings of one environment variable (others fixed by the caller's environment)
V=$1; shift
for r in 1 2; do
  for v in "$@"; do
    b=$(env $V=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])")
    echo "round $r $V=$v: step_ms $b"
  done
done
