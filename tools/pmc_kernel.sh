#!/bin/bash
# A few SQ counters for the kernels of a bench run, one rocprofv3 pass per counter group (run on the GPU box):
#   tools/pmc_kernel.sh "<bench args>" "<kernel name substring>" COUNTER [COUNTER ...]
set -u
ARGS=$1; shift
KERN=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/p_k
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/p_k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dp-path --train-loop-steps 0 $ARGS > /dev/null 2>&1
  python3 - "$KERN" <<PY
import csv, sys, glob, collections
f = glob.glob("/tmp/p_k/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r["Kernel_Name"]:
        tot[(r["Kernel_Name"][:48], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:48], r["Counter_Name"])] += 1
for k in sorted(tot): print("%-50s %-28s %14.0f per launch (%d launches)" % (k[0], k[1], tot[k] / n[k], n[k]))
PY
done
