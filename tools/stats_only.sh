#!/bin/bash
# rocprofv3 kernel statistics of the benchmark step only.   tools/stats_only.sh <tag>   (run on the GPU box)
TAG=${1:-stats}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dp-path --train-loop-steps 0 > $OUT/stats_bench.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
steps = max(int(r["Calls"]) for r in rows if "k_reg8_bwd" in r["Name"] or "k_reg_bwd" in r["Name"])      # one launch per step
tot = 0
for r in rows:
    if "cf::" in r["Name"]:
        per_step = float(r["TotalDurationNs"]) / steps / 1000
        tot += per_step
        print("%8.1f us/step  %4d calls  avg %7.1f us  %s" % (per_step, int(r["Calls"]), float(r["AverageNs"]) / 1000, r["Name"][:90]))
print("sum %.1f us/step" % tot)
PY
