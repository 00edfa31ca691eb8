#!/bin/bash
# A/B of library variants on one box: tools/ab.sh build/lib_a.so build/lib_b.so ...   (each: ms/step eager, graph replay; two interleaved rounds)
for round in 1 2; do
  for lib in "$@"; do
    CF_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['graph_replay_ms_per_step'], {k: round(v, 1) for k, v in d.get('kernel_us', {}).items()} if 'kernel_us' in d else '')"
  done
done
