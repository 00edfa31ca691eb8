#!/bin/bash
# Run-to-run determinism of the data-parallel step, two gloo ranks sharing ONE device (tools/dp_feed_determinism.py), six runs replayed and six eager: every
# saved tensor of every run must equal the first run's.   tools/dp_soak.sh   (run on the GPU box)
cd $GRAFT_REPO_ROOT
export CF_SHARE_DEVICE=1
for i in 1 2 3 4 5 6; do
  for g in 1 0; do
    DP_GRAPH=$g python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + i * 2 + g)) tools/dp_feed_determinism.py /tmp/dp_${g}_$i.pt 40 > /tmp/dp_${g}_$i.log 2>&1 || { echo "run $i graph=$g failed"; tail -5 /tmp/dp_${g}_$i.log; }
  done
done
python3 - <<'PY'
import torch, glob
for g in (1, 0):
    runs = [torch.load(f, weights_only=False) for f in sorted(glob.glob("/tmp/dp_%d_*.pt" % g))]
    ref = runs[0]
    bad = 0
    for r in runs[1:]:
        same = all(torch.equal(torch.as_tensor(ref[k]), torch.as_tensor(r[k])) if not isinstance(ref[k], (list, tuple)) else ref[k] == r[k] for k in ref)
        bad += not same
    print("dp two ranks on one device, %s: %d runs of 40 steps, %d differ from the first; keys %s" % ("graph" if g else "eager", len(runs), bad, list(ref)[:6]))
PY
