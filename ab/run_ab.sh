#!/bin/bash
# A/B: interleaved bench rounds of two library builds (CF_LIB_PATH), HIP-event time of the selected kernel
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "main" ]; then unset CF_LIB_PATH; else export CF_LIB_PATH=$PWD/ab/$v; fi
    f=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --roofline-kernel k_reg_fwd --no-graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_us'], d['ms_per_step'])")
    b=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_us'], d['ms_per_step'])")
    echo "round $r $v: fwd_us/ms $f   bwd_us/ms $b"
  done
done
