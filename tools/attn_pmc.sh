#!/bin/bash
# Issue / stall counters of the dense attention kernels (stress configuration): one rocprofv3 --pmc pass over tools/stress_bench.py.
#   tools/attn_pmc.sh <out file>      (run on the GPU box)
# Per kernel and launch: matrix-pipe busy cycles, wave cycles split into parked (s_waitcnt / barrier), issue-stalled and issuing, LDS
# conflict cycles, and the effective clock (GRBM_GUI_ACTIVE / duration) -- SQ_* wave counters are in quad-cycles (MI355X_MICROARCH.md).
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/attn_pmc.txt}
case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac      # (the script changes directory: a relative path is relative to the repository)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_attn /tmp/p_attn2
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_attn --output-format csv -- python3 $R/tools/stress_bench.py --reps 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d /tmp/p_attn2 --output-format csv -- python3 $R/tools/stress_bench.py --reps 2 > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, collections, glob, sys
out = open(sys.argv[1], "w")
for d in ("/tmp/p_attn", "/tmp/p_attn2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    ts = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not fs:
        out.write("%s: no counter file\n" % d); continue
    dur = collections.defaultdict(list)
    if ts:
        for r in csv.DictReader(open(ts[0])):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(fs[0])):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]][r["Counter_Name"]] += 1
    for k, v in tot.items():
        if "k_attn" not in k: continue
        out.write("%s   launches %d   avg duration %.1f us\n" % (k[:60], max(n[k].values()), sum(dur[k]) / max(len(dur[k]), 1) / 1e3))
        for c, x in sorted(v.items()):
            out.write("    %-34s %16.0f per launch\n" % (c, x / n[k][c]))
out.close()
print(open(sys.argv[1]).read())
PY
