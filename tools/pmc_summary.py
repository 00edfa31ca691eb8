"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the gfx950 guide
prescribes):  python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <out.csv>
hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(calibrated on k_adamw, whose traffic is known exactly: 4 reads + 3 writes of the active parameter range)."""
import csv, json, re, sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def short(name):
    m = re.match(r"(?:void )?cf::(k_\w+)", name)
    # the 512-thread Regulation kernels (cf_reg8.h) keep the names bench.py's roofline uses
    return {"k_reg8_fwd": "k_reg_fwd", "k_reg8_bwd": "k_reg_bwd"}.get(m.group(1), m.group(1)) if m else None


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out, rows = {}, []
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    s = short(k)
    if not s or k not in write:
        continue
    f, w = fetch[k][0], write[k][0]
    hbm = (2.0 * f + w) * 1024.0
    rows.append((k, fetch[k][1], round(f, 1), round(w, 1), int(hbm)))
    if s not in out or hbm > out[s]["hbm_bytes_per_launch"]:
        out[s] = {"hbm_bytes_per_launch": hbm, "fetch_kb": f, "write_kb": w, "instantiation": k}
json.dump({"note": __doc__.split("\n\n")[0].replace("\n", " ") if False else
           "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --no-graph`, bsz 64; hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024: "
           "FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (calibrated on k_adamw)",
           "commit": sys.argv[5] if len(sys.argv) > 5 else "unknown", "kernels": out},
          open(sys.argv[3], "w"), indent=1)
with open(sys.argv[4], "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
    for r in rows:
        fh.write('"%s",%d,%s,%s,%d\n' % r)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
