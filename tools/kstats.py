"""Kernel averages of a rocprofv3 --kernel-trace --stats run:  python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= int(sys.argv[2]) if len(sys.argv) > 2 else i >= 8:
        break
    print("%-60s calls %5s  avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
