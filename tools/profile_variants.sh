#!/bin/bash
# Kernel breakdown (rocprofv3 --kernel-trace --stats) of a training step at d_emb = 256 and 64 (tools/variant_step_time.py): the shapes that run the
# stand-alone kernels layer by layer.   tools/profile_variants.sh   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for d in 256 64; do
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/variant_step_time.py $d > /dev/null 2>&1
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
echo "== d_emb $d"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
