"""`python -m chromoformer_amd.sweep` -- the cross-validation sweep (reference: chromoformer/Snakefile, which pins
cell lines to two GPUs with Snakemake resources).  The (cell line, fold) trainings are independent: they run as
REPLICAS, one `chromoformer_amd.train` process per GPU at a time, no collective -- 11 cell lines x 4 folds over the 8
GPUs of a node is 44 jobs, 5-6 per GPU.

    python -m chromoformer_amd.sweep --meta-template data/{eid}/train.csv --npy-dir-template data/{eid}/npy \\
        --config configs/default.yaml --eids E003 E004 ... --folds 1 2 3 4 --gpus 8 [--regression] [--conf 1]

Checkpoints go to `ckpts/{eid}/{exp_id}-{eid}-conf{conf}-fold{fold}.pt` (Snakefile:25,50,58).  A job counts as finished
only when train.py has written its `<checkpoint>.done` marker after the last epoch (train.py rewrites the checkpoint
every epoch, so the file alone proves nothing -- Snakemake deletes incomplete outputs, this is the equivalent); finished
jobs are skipped, so an interrupted sweep resumes and crashed or killed jobs are run again.  A job that exits non-zero
has its marker and checkpoint removed.  `--fold` values are the Snakefile's 1..4; train.py's own fold
argument is 0-based modulo 4 (train.py:99-105), so fold f is passed as f % 4 -- the same validation quarter.

`--jobs-per-gpu J` (default 1): J trainings share a GPU at a time.  One training step is a chain of five launches of <= 192 long-running
workgroups on 256 CUs: a second, independent job fills the CUs and the launch tails the first leaves idle -- measured on one
MI355X, two jobs at once run at 79.2 k + 79.9 k = 159 k genes/s against 127 k for one (profiles/r05j_two_jobs.txt: +25 % for the sweep as a whole;
every job takes 1.6x as long).  The jobs are separate processes with separate stores and weights: replicas, nothing shared but the device."""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time


def plan(eids, folds, exp_id, conf, out_dir="ckpts"):
    """-> [(eid, fold, checkpoint path)] in the Snakefile's expand() order."""
    return [(e, f, os.path.join(out_dir, e, "%s-%s-conf%s-fold%s.pt" % (exp_id, e, conf, f))) for e in eids for f in folds]


def command(eid, fold, ckpt, args):
    cmd = [sys.executable, "-m", "chromoformer_amd.train", "-o", ckpt, "-c", args.config, "--exp-id", args.exp_id,
           "-m", args.meta_template.format(eid=eid), "-d", args.npy_dir_template.format(eid=eid), "--fold", str(int(fold) % 4)]
    if args.binsizes:
        cmd += ["--binsizes"] + [str(b) for b in args.binsizes]
    if args.regression:
        cmd.append("--regression")
    return cmd


def run(jobs, args, launch=subprocess.Popen):
    """Greedy list scheduling: a GPU takes the next pending job as soon as its previous one exits.
    -> {checkpoint: return code}"""
    finished = lambda ckpt: os.path.exists(ckpt) and os.path.exists(ckpt + ".done")
    pending = [j for j in jobs if not finished(j[2])]
    done = {j[2]: 0 for j in jobs if finished(j[2])}
    running = {}                                    # slot -> (process, checkpoint); slot s runs on GPU s % gpus
    per_gpu = max(1, int(getattr(args, "jobs_per_gpu", 1) or 1))
    while pending or running:
        for slot in range(args.gpus * per_gpu):     # (slots 0 .. gpus - 1 first: every GPU gets a job before any gets a second one)
            if slot not in running and pending:
                gpu = slot % args.gpus
                eid, fold, ckpt = pending.pop(0)
                os.makedirs(os.path.dirname(ckpt) or ".", exist_ok=True)
                env = dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu))
                log = open(ckpt + ".log", "w")
                running[slot] = (launch(command(eid, fold, ckpt, args), env=env, stdout=log, stderr=subprocess.STDOUT), ckpt)
                print("[sweep] gpu %d <- %s fold %s" % (gpu, eid, fold), flush=True)
        for gpu, (p, ckpt) in list(running.items()):
            rc = p.poll()
            if rc is not None:
                if rc == 0 and not finished(ckpt):
                    rc = -1                         # exited cleanly without reaching its last epoch
                if rc != 0:
                    for path in (ckpt, ckpt + ".done", ckpt + ".tmp"):
                        if os.path.exists(path):
                            os.remove(path)
                done[ckpt] = rc
                del running[gpu]
                print("[sweep] gpu %d done rc=%d %s" % (gpu % args.gpus, rc, ckpt), flush=True)
        if running:
            time.sleep(args.poll)
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--meta-template", required=True, help="metadata csv path with an {eid} placeholder")
    ap.add_argument("--npy-dir-template", required=True, help="signal directory with an {eid} placeholder")
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--exp-id", default="chromoformer-reproduction")
    ap.add_argument("--conf", default="1")
    ap.add_argument("--eids", nargs="+", default=["E003", "E004", "E005", "E006", "E007", "E016", "E066", "E087", "E114", "E116", "E118"])
    ap.add_argument("--folds", nargs="+", default=["1", "2", "3", "4"])
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--jobs-per-gpu", type=int, default=1, help="trainings that share a GPU at a time (2: +25 %% sweep throughput measured, see the module docstring)")
    ap.add_argument("--out-dir", default="ckpts")
    ap.add_argument("--binsizes", nargs="+", default=None)
    ap.add_argument("--regression", action="store_true")
    ap.add_argument("--poll", type=float, default=1.0)
    args = ap.parse_args(argv)
    done = run(plan(args.eids, args.folds, args.exp_id, args.conf, args.out_dir), args)
    bad = {k: v for k, v in done.items() if v != 0}
    print("[sweep] %d jobs, %d failed" % (len(done), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
