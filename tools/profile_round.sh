#!/bin/bash
# rocprofv3 evidence of one round: kernel statistics of the benchmark step, HBM traffic (separate FETCH_SIZE / WRITE_SIZE
# passes, as the gfx950 guide prescribes) and MFMA-pipe occupancy, plus the bench lines of the variants DESIGN.md quotes.
#   tools/profile_round.sh <tag>   (run on the GPU box; every step under its own `timeout`)
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="$R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats --output-format csv -- python3 $B > $OUT/stats_bench.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch --output-format csv -- python3 $B --eager > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write --output-format csv -- python3 $B --eager > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/p_write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json $OUT/pmc_hbm_traffic.csv "${CF_COMMIT:-unknown}"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_mfma --output-format csv -- python3 $B --eager > /dev/null 2>&1
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
cd $R
# the bench lines DESIGN.md quotes (un-profiled runs)
timeout 400 python3 bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --graph > $OUT/bench_graph.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --roofline-kernel k_reg_fwd > $OUT/bench_k_reg_fwd.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --roofline-kernel k_wgrad > $OUT/bench_k_wgrad.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --roofline-kernel k_trunk_fwd > $OUT/bench_k_trunk_fwd.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --roofline-kernel k_trunk_bwd > $OUT/bench_k_trunk_bwd.json 2> /dev/null
timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --regime realistic > $OUT/bench_realistic.json 2> /dev/null
timeout 200 python3 bench.py --config stress --steps 20 --warmup 3 > $OUT/bench_stress.json 2> /dev/null
CF_TRUNK=0 timeout 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 > $OUT/bench_no_trunk.json 2> /dev/null
timeout 100 python3 tools/trunk_stamps.py > $OUT/trunk_stamps.txt 2> /dev/null
timeout 100 python3 tools/bin_bench.py 2> /dev/null | tail -1 > $OUT/binning.json
timeout 200 python3 tools/stress_bench.py 2> /dev/null | tail -1 > $OUT/stress_attention.json
CF_ATTN_BWD_V1=1 timeout 200 python3 tools/stress_bench.py 2> /dev/null | tail -1 > $OUT/stress_attention_bwd64.json      # the 64-keys-per-pass backward of rounds 4-5, same box
# HBM traffic of the dense attention kernels at the stress shape (separate FETCH_SIZE / WRITE_SIZE passes), both backward kernels; issue / stall counters
( cd /tmp && for v in 0 1; do rm -rf /tmp/pa_f /tmp/pa_w
    CF_ATTN_BWD_V1=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pa_f --output-format csv -- python3 $R/tools/stress_bench.py --reps 2 > /dev/null 2>&1
    CF_ATTN_BWD_V1=$v timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pa_w --output-format csv -- python3 $R/tools/stress_bench.py --reps 2 > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find /tmp/pa_f -name "*counter_collection.csv" | head -1) $(find /tmp/pa_w -name "*counter_collection.csv" | head -1) /tmp/pa.json $OUT/stress_pmc_hbm_traffic_bwdv1_$v.csv "${CF_COMMIT:-unknown}" > /dev/null
  done )
timeout 400 tools/attn_pmc.sh $OUT/attn_pmc.txt > /dev/null 2>&1
timeout 100 python3 tools/reg_stamps.py fwd 2 > $OUT/reg_stamps_fwd.txt 2> /dev/null
timeout 100 python3 tools/reg_stamps.py bwd 2 > $OUT/reg_stamps_bwd.txt 2> /dev/null
# the driver's own command on this box (20 steps behind the disclosed pre-warm)
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> /dev/null
# attention-kernel HBM evidence on the current code (north_star clause) and the data-parallel step's kernel timeline on a one-rank RCCL group
timeout 600 tools/attc_bandwidth.sh $TAG > /dev/null 2>&1
( cd /tmp && for m in eager graph; do rm -rf /tmp/pd; timeout 200 rocprofv3 --kernel-trace -d /tmp/pd --output-format csv -- python3 $R/tools/dp_probe.py $m 100 2> /dev/null | tail -1 > $OUT/dp_timeline_$m.txt; python3 $R/tools/timeline.py $(find /tmp/pd -name "*kernel_trace.csv" | head -1) 60 k_trunk_fwd >> $OUT/dp_timeline_$m.txt; done )
timeout 200 python3 tools/epoch_evidence.py --bench-genes-per-s $(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['train_loop']['value'])" 2> /dev/null || echo 0) > $OUT/epoch_18955.txt 2>&1
ls -la $OUT
