#!/bin/bash
# A/B of compile-time switches on BOTH axes the review asks about: step time and HBM-side traffic per step (separate FETCH_SIZE / WRITE_SIZE passes).
#   tools/traffic_ab.sh "<flags 1>" "<flags 2>" ...        (run on the GPU box; rebuilds the plain library on exit)
R=$GRAFT_REPO_ROOT
cd $R
trap 'CF_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1' EXIT
B="$R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --no-extras"
for fl in "$@"; do
  if ! CF_HIPCC_FLAGS="$fl" python -c "import __graft_entry__ as g; g.build()" > /tmp/tab_build.log 2>&1; then echo "[$fl]: BUILD FAILED"; tail -3 /tmp/tab_build.log; continue; fi
  for i in 1 2; do echo -n "[$fl] ms per step: "; CF_HIPCC_FLAGS="$fl" timeout 300 python3 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tf /tmp/tw
    CF_HIPCC_FLAGS="$fl" timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/tf --output-format csv -- python3 $B --eager > /dev/null 2>&1
    CF_HIPCC_FLAGS="$fl" timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/tw --output-format csv -- python3 $B --eager > /dev/null 2>&1
    echo -n "[$fl] MB per launch: "; python3 $R/tools/pmc_summary.py $(find /tmp/tf -name "*counter_collection.csv" | head -1) $(find /tmp/tw -name "*counter_collection.csv" | head -1) /tmp/t.json /tmp/t.csv ab )
done
