#!/bin/bash
# HBM bandwidth of the centre-row attention kernels on the CURRENT code (north_star: "HBM utilisation in the attention kernels").  In the
# default step they are phases of k_trunk_fwd / k_trunk_bwd; CF_TRUNK=0 runs the same bodies as launches of their own (k_attc1: one region
# per workgroup, the Embedding stage; k_attc2<., 8>: eight regions per workgroup, the Pairwise stage), which is what can be profiled per kernel.
# Three rocprofv3 passes of the same command (kernel trace for durations; FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as the gfx950
# guide prescribes; hbm = (2 FETCH_SIZE + WRITE_SIZE) x 1024, calibrated on k_adamw).   tools/attc_bandwidth.sh <tag>   (on the GPU box)
set -u
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CF_TRUNK=0
B="$GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --prewarm-s 0 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --eager"
rm -rf /tmp/pa_s /tmp/pa_f /tmp/pa_w
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa_s --output-format csv -- python3 $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pa_f --output-format csv -- python3 $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pa_w --output-format csv -- python3 $B > /dev/null 2>&1
python3 - "$OUT/attc_bandwidth.csv" <<'PY'
import csv, glob, sys, collections
def one(pat): return glob.glob(pat, recursive=True)[0]
stats = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(one("/tmp/pa_s/**/*kernel_stats.csv")))}
def pmc(d, c):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(one(d + "/**/*counter_collection.csv"))):
        if r["Counter_Name"] == c:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}
f, w = pmc("/tmp/pa_f", "FETCH_SIZE"), pmc("/tmp/pa_w", "WRITE_SIZE")
B, S, F, Ls = 64, 8, 7, (20, 80, 400)
# compulsory bytes of one launch over the three resolutions: per region the 7-mark features (L x F x 4), its pad-mask row (L), the saved
# probabilities of two heads (2 x L x 4; written by the forward, read by the backward) and the operand rows in and out (2 heads x (128 + F) x 4, twice)
def alg(regions): return regions * sum(L * F * 4 + L + 2 * L * 4 + 2 * 2 * (128 + F) * 4 for L in Ls)
with open(sys.argv[1], "w") as o:
    o.write("kernel,regions_per_resolution,avg_us,algorithmic_bytes,algorithmic_GBps,frac_of_8TBps,pmc_hbm_bytes,pmc_GBps,pmc_frac_of_8TBps\n")
    for k in sorted(stats):
        if "k_attc" not in k or k not in f or k not in w:
            continue
        regions = B * S if "k_attc2" in k else B
        us, a, h = stats[k], alg(regions), (2 * f[k] + w[k]) * 1024
        o.write('"%s",%d,%.2f,%d,%.0f,%.4f,%d,%.0f,%.4f\n' % (k, regions, us, a, a / us / 1e3, a / us / 1e3 / 8000, h, h / us / 1e3, h / us / 1e3 / 8000))
print(open(sys.argv[1]).read())
PY
