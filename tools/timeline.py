"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <kernel_trace.csv> [step] [anchor kernel substring]
Prints, for one step in the middle of the run, every kernel with start offset / duration / stream-queue, so that
overlap between the two streams of the step engine can be read off."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_retile"      # the first kernel of a step
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f us  +%7.1f us  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
