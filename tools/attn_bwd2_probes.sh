#!/bin/bash
# Timing probes of k_attn_bwd2 (stress shape): the kernel with one ingredient removed at a time (WRONG results, time only) + its issue / stall counters.
#   tools/attn_bwd2_probes.sh <out file>      (run on the GPU box)
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/attn_bwd2_probes.txt}
cd $GRAFT_REPO_ROOT
CF_AB_CMD="python3 tools/stress_bench.py --reps 3" CF_AB_COLS=140 tools/flags_ab.sh - "" "-DCF_AB2_PROBE=1" "-DCF_AB2_PROBE=2" "-DCF_AB2_PROBE=4" "-DCF_AB2_PROBE=8" "-DCF_AB2_PROBE=16" "-DCF_AB2_PROBE=32" "-DCF_AB2_PROBE=64" "-DCF_AB2_PROBE=15" "-DCF_AB2_SCHED=0" > $OUT 2>&1
cat $OUT
