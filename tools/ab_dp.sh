#!/bin/bash
# A/B of library variants on the one-rank data-parallel path: tools/ab_dp.sh build/lib_a.so build/lib_b.so ...
for round in 1 2; do
  for lib in "$@"; do
    CF_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --train-loop-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], {k: v for k, v in d['dp_path_ms_per_step'].items() if k != 'max_step_ms'})"
  done
done
