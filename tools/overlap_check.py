"""Do two kernels of a step overlap in time?  Reads a rocprofv3 --kernel-trace CSV:  python tools/overlap_check.py <kernel_trace.csv> k_trunk_bwd k_reduce"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
a = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if sys.argv[2] in r["Kernel_Name"]]
b = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows if sys.argv[3] in r["Kernel_Name"]]
a.sort(); b.sort()
print(len(a), "launches of", sys.argv[2], "/", len(b), "of", sys.argv[3])
for s, e in a[-3:]:
    near = [x for x in b if x[1] > s - 200000 and x[0] < e + 200000]
    print("%s: %.1f us" % (sys.argv[2], (e - s) / 1e3))
    for bs, be, n in near:
        ov = max(0, min(e, be) - max(s, bs))
        print("    %-40s start %+8.1f us  dur %6.1f us  overlap %6.1f us" % (n, (bs - s) / 1e3, (be - bs) / 1e3, ov / 1e3))
