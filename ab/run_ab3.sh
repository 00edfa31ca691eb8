#!/bin/bash
# A/B of whole-step time between library builds (CF_LIB_PATH), three interleaved rounds
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "main" ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=$PWD/ab/$v; fi
    b=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "round $r $v: step_ms $b"
  done
done
