#!/bin/bash
# rocprofv3 evidence of one round: kernel statistics of the benchmark step, HBM traffic (separate FETCH_SIZE / WRITE_SIZE
# passes, as the gfx950 guide prescribes) and MFMA-pipe occupancy.   tools/profile_round.sh <tag>   (run on the GPU box)
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dp-path --train-loop-steps 0"
rocprofv3 --kernel-trace --stats -d /tmp/p_stats --output-format csv -- python3 $B > $OUT/stats_bench.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch --output-format csv -- python3 $B --no-graph > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write --output-format csv -- python3 $B --no-graph > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/p_write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json $OUT/pmc_hbm_traffic.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_mfma --output-format csv -- python3 $B --no-graph > /dev/null 2>&1
python3 - <<PY
import csv, collections
f = "$(find /tmp/p_mfma -name '*counter_collection.csv' | head -1)"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[r["Kernel_Name"]] += 1
with open("$OUT/mfma_utilisation.csv", "w") as o:
    o.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,GRBM_GUI_ACTIVE_per_launch,mfma_pipe_busy_fraction (busy / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs))\n")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        if "cf::" not in k or not n[k]: continue
        busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k], v["GRBM_GUI_ACTIVE"] / n[k]
        o.write('"%s",%d,%.0f,%.0f,%.4f\n' % (k, n[k], busy, act, busy / (act / 8 * 1024) if act else 0))
PY
ls -la $OUT
