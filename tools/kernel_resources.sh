#!/bin/bash
# Registers, LDS and scratch of every kernel of the library (device-only compile of csrc/cf_api.hip to assembly, metadata grep):
#   tools/kernel_resources.sh [pattern]      e.g.  tools/kernel_resources.sh k_trunk
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${CF_ASM_OUT:-/tmp/cf_api_gfx950.s}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed $CF_HIPCC_FLAGS --cuda-device-only -S "$R/chromoformer_amd/csrc/cf_api.hip" -o "$OUT" || exit 1
python3 - "$OUT" "${1:-}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
print("%-90s %5s %5s %8s %8s %6s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "spill"))
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat and pat not in name:
        continue
    print("%-90s %5s %5s %8s %8s %6s" % (name[:90], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
PY
