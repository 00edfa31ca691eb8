#!/bin/bash
# A/B of whole-step time (and one HIP-event-timed kernel, $KERNEL) between library builds
K=${KERNEL:-k_wgrad}
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "main" ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=$PWD/ab/$v; fi
    b=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    k=$(python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --roofline-kernel $K 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_us'])")
    echo "round $r $v: step_ms $b   $K us $k"
  done
done
