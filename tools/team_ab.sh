#!/bin/bash
# builds the library with each flag set and times the team forward
cd $GRAFT_REPO_ROOT
for fl in "-DCF_TQ_SLEEP=2" "-DCF_TQ_SLEEP=0" "-DCF_TQ_SLEEP=6" "-DCF_TQ_PRE=2" "-DCF_TQ_PRE=4"; do
  CF_HIPCC_FLAGS="$fl" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for i in 1 2; do
    echo -n "$fl: "
    CF_HIPCC_FLAGS="$fl" CF_REG_TEAM=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --roofline-kernel k_reg_fwd 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
