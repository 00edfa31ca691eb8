#!/bin/bash
# Aggregate training throughput of J independent jobs sharing ONE GPU (the sweep's --jobs-per-gpu): J bench.py processes at once.
#   tools/two_jobs_one_gpu.sh [J ...]      (default: 1 2 3)
cd $GRAFT_REPO_ROOT
B="bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc"
for J in ${@:-1 2 3}; do
  pids=""
  for j in $(seq $J); do python3 $B > /tmp/j$j.txt 2>/dev/null & pids="$pids $!"; done
  wait $pids
  python3 - $J <<'PY'
import json, sys
J = int(sys.argv[1])
v = [json.loads(open("/tmp/j%d.txt" % j).read().strip().split("\n")[-1]) for j in range(1, J + 1)]
print("%d job(s) at once: %s genes/s each, %.1f k genes/s together, %s ms per step" % (J, " + ".join("%.1f k" % (x["value"] / 1e3) for x in v), sum(x["value"] for x in v) / 1e3, " / ".join("%.4f" % x["ms_per_step"] for x in v)))
PY
done
