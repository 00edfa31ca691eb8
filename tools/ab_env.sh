#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh "A=1" "A=2 B=3" ...   (ms/step eager, graph replay, launches; two interleaved rounds)
for round in 1 2; do
  for e in "$@"; do
    env $e timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d['graph_replay_ms_per_step'], d['config']['launches_per_step'])"
  done
done
