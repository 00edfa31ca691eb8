#!/bin/bash
# A/B of compile-time switches: builds the library on the GPU box with each flag set and prints step / kernel time.  A failed build is
# reported and skipped (never timed under the previous variant's label); on exit the library is rebuilt without flags, so no
# experimental variant stays behind as the shipped one (chromoformer_amd/_lib.py also refuses a library whose flags do not match).
#   tools/flags_ab.sh <roofline kernel> "<flags 1>" "<flags 2>" ...
#   CF_AB_CMD="python3 tools/stress_bench.py" tools/flags_ab.sh - "<flags 1>" ...     (another command: its last output line is printed)
cd $GRAFT_REPO_ROOT
trap 'CF_HIPCC_FLAGS="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1' EXIT
K=$1; shift
for fl in "$@"; do
  if ! CF_HIPCC_FLAGS="$fl" python -c "import __graft_entry__ as g; g.build()" > /tmp/flags_ab_build.log 2>&1; then
    echo "[$fl]: BUILD FAILED"; tail -5 /tmp/flags_ab_build.log; continue
  fi
  if [ -n "${CF_AB_CMD:-}" ]; then
    for i in 1 2; do echo -n "[$fl]: "; CF_HIPCC_FLAGS="$fl" timeout 300 $CF_AB_CMD 2>/dev/null | tail -1 | cut -c1-${CF_AB_COLS:-260}; done
    continue
  fi
  for i in 1 2; do
    echo -n "[$fl]: "
    CF_HIPCC_FLAGS="$fl" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc --roofline-kernel $K 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
