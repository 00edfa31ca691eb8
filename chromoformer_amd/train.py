"""`python -m chromoformer_amd.train` -- drop-in for the reference's training entrypoint.

Same command line (-o -c --exp-id -m -d --fold [--binsizes ...] [--regression] [--use-wandb]),
same config.yaml schema and same .pt checkpoint layout as
/root/reference/chromoformer/train.py:26-37, 45-68, 322-343.  What changed underneath:

  * the model runs on the HIP path (chromoformer_amd.net) and a step is one library sequence
    (forward, loss, backward, [RCCL all-reduce], AdamW) replayed from a hipGraph;
  * genes are binned once, on the GPU (cf_bin_regions), into a store that stays resident in HBM, instead of per
    step in 8 loader processes; batches are gathered on the device;
  * launched under torch.distributed.run it trains data-parallel, one process per GPU
    (`bsz` is then the per-GPU batch and the gradients are averaged over ranks);
  * `--binsizes` given on the command line are parsed as ints (the reference crashes on them).

Behaviour that is kept on purpose: `for epoch in range(1, num_epoch)` (num_epoch - 1 epochs),
the model constructor reseeding torch with 42, the checkpoint written BEFORE the scheduler step,
metrics scaled by 100, validation loss computed on the host over the concatenated outputs, and the
DataLoader's consumption of the global torch RNG (so the shuffled batch order is the reference's).
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import pandas as pd
import torch
import torch.nn as nn
import yaml

from .data import ChromoformerDataset, GeneStore, shard_indices, static_epoch_batches, static_shard
from .engine import EpochFeed, Trainer
from .net import ChromoformerClassifier, ChromoformerRegressor
from .util import seed_everything


def _wandb(enabled):
    try:
        import wandb
        return wandb
    except ImportError:
        if enabled:
            raise SystemExit("--use-wandb given but wandb is not installed")
        stub = types.SimpleNamespace(init=lambda *a, **k: None, log=lambda *a, **k: None,
                                     config=types.SimpleNamespace(update=lambda *a, **k: None),
                                     summary=types.SimpleNamespace(update=lambda *a, **k: None))
        return stub


def _draw_loader_seed():
    """One int64 draw from the global torch RNG, as DataLoader / RandomSampler do."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def epoch_permutation(n):
    """Index order of DataLoader(shuffle=True): the iterator draws its base seed, then the
    RandomSampler seeds a private generator from the global RNG and takes a randperm."""
    _draw_loader_seed()
    g = torch.Generator()
    g.manual_seed(_draw_loader_seed())
    return torch.randperm(n, generator=g).tolist()


def optimizer_state_dict(model, lr, initial_lr=None, m_host=None, v_host=None):
    """torch.optim.AdamW.state_dict() layout: entries only for parameters that ever had a gradient,
    `step` as a 0-dim float32 tensor (train.py:325, 336).  m_host / v_host: host copies of the flat moment buffers (one
    device-to-host copy each instead of one per tensor); default: copied here."""
    named = list(model.named_parameters())
    ref = torch.optim.AdamW([nn.Parameter(torch.zeros(1)) for _ in named], lr=float(lr))
    sd = ref.state_dict()
    sd["param_groups"][0]["lr"] = float(lr)
    # the reference's optimizer has a StepLR attached (train.py:157-158), which adds this key to the group: a consumer that
    # loads the state and rebuilds the scheduler with last_epoch >= 0 needs it
    sd["param_groups"][0]["initial_lr"] = float(initial_lr if initial_lr is not None else lr)
    state = {}
    if model._step > 0:
        if m_host is None:
            m_host, v_host = model._mflat.detach().cpu(), model._vflat.detach().cpu()
        for i, e in enumerate(model._table):
            if not e["trainable"]:
                continue
            sl = slice(e["offset"], e["offset"] + e["numel"])
            state[i] = {"step": torch.tensor(float(model._step)),
                        "exp_avg": m_host[sl].view(e["shape"]).clone(),
                        "exp_avg_sq": v_host[sl].view(e["shape"]).clone()}
    sd["state"] = state
    return sd


class CheckpointWriter:
    """The per-epoch checkpoint (train.py:321-341) off the critical path.  A Roadmap-sized epoch is 0.13 s of GPU work; building
    the reference's checkpoint the obvious way (one .cpu() per tensor: 370 + 668 copies, a fresh torch.optim.AdamW for its
    state layout, then torch.save of 64 MB) took 0.9 s.  Here the epoch loop pays three device-to-host copies of the flat
    parameter / moment buffers into pinned memory (~3 ms); the dictionaries in the reference's layout -- state_dict order and
    (legacy-aware) key names, AdamW's state / param_groups -- are built ONCE as views of those pinned buffers, so a checkpoint
    is torch.save + rename on a worker thread under the next epoch, with next to no Python work that would compete with the
    step loop for the interpreter.  (The tensors of a loaded checkpoint therefore share three storages; values, shapes, dtypes,
    keys and order are the reference's.)  One save is in flight at a time -- submit() waits for the previous one before it
    overwrites the buffers --, wait() re-raises a failed save, and the run writes its `.done` marker only behind the last wait()."""

    def __init__(self, model):
        import threading
        self._threading = threading
        self.model = model
        n = model._flat.numel()
        self._host = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(3)]
        self._thread, self._err = None, None
        self._go = threading.Event()
        p_h, m_h, v_h = self._host
        table = {e["name"]: e for e in model._table}
        keys, named = list(model.state_dict().keys()), [n_ for n_, _ in model.named_parameters()]
        if len(keys) != len(named):
            raise RuntimeError("state_dict keys and parameters disagree")
        view = lambda buf, e: buf[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])
        self._net = type(model.state_dict())((k, view(p_h, table[n_])) for k, n_ in zip(keys, named))
        # torch.optim.AdamW.state_dict() layout (train.py:325, 336): entries only for parameters that ever had a gradient, `step` a
        # 0-dim float32 tensor per parameter; param_groups from a real AdamW over as many parameters (defaults, ids), plus the
        # `initial_lr` the reference's attached StepLR adds
        self._opt = torch.optim.AdamW([nn.Parameter(torch.zeros(1)) for _ in named], lr=1.0).state_dict()
        self._steps = []
        self._state = {}
        for i, e in enumerate(model._table):
            if e["trainable"]:
                st = torch.zeros((), dtype=torch.float32)
                self._steps.append(st)
                self._state[i] = {"step": st, "exp_avg": view(m_h, e), "exp_avg_sq": view(v_h, e)}

    def go(self):
        """The step loop has queued its epoch (train_epoch's `after_issue`): the pending save may start.  Until then the worker
        stays off the interpreter -- a thread that pickles while the main thread issues the first steps of an epoch into an empty
        stream costs the GPU ~13 ms of idling per epoch (5 ms interpreter switch intervals; measured 0.137 vs 0.124 s per
        222-step epoch)."""
        self._go.set()

    def wait(self):
        if self._thread is not None:
            self._go.set()
            self._thread.join()
            self._thread = None
        if self._err is not None:
            err, self._err = self._err, None
            raise err

    def submit(self, path, ckpt, lr, initial_lr):
        """ckpt: the epoch's dictionary with net / optimizer still None (filled here, in the reference's key order)."""
        self.wait()                                   # the pinned buffers are the previous save's until it has finished
        m = self.model
        for dst, src in zip(self._host, (m._flat, m._mflat, m._vflat)):
            dst.copy_(src.detach(), non_blocking=True)
        torch.cuda.current_stream(m._device).synchronize()
        for st in self._steps:
            st.fill_(float(m._step))
        self._opt["param_groups"][0]["lr"] = float(lr)
        self._opt["param_groups"][0]["initial_lr"] = float(initial_lr if initial_lr is not None else lr)
        self._opt["state"] = self._state if m._step > 0 else {}
        ckpt["net"], ckpt["optimizer"] = self._net, self._opt

        self._go.clear()

        def work():
            try:
                self._go.wait(timeout=2.0)            # (released by go() / wait(); the timeout only bounds a caller that does neither)
                torch.save(ckpt, path + ".tmp")
                os.replace(path + ".tmp", path)
            except BaseException as e:                # surfaced by the next wait()
                self._err = e

        self._thread = self._threading.Thread(target=work, name="cf-checkpoint", daemon=False)
        self._thread.start()


def validation_metrics(val_out, val_lab, regression):
    """The numbers train.py:277-318 prints and stores after an epoch, from the logits [n, n_out] and labels [n] of the validation
    split (numpy arrays): -> (loss, score array, dict of metrics in percent).  Same definitions as the reference's calls --
    nn.CrossEntropyLoss / nn.MSELoss (for the regressor the reference compares [n,1] outputs with [n] labels, i.e. a broadcast
    [n,n] mean, train.py:280-283: `last_val_loss` in the checkpoint is that number, so it is kept), sklearn's accuracy / ROC AUC /
    average precision / r2, scipy's pearsonr (tests/test_metrics_cpu.py pins them to those) -- in a few numpy passes: the sklearn
    calls cost 0.3 s on a 4,700-gene split, twice the GPU time of the whole epoch in front of them, and a torch CPU op would wake
    torch's intra-op pool next to the step loop."""
    z = np.asarray(val_out, dtype=np.float32)
    if regression:
        pred, lab = z.reshape(-1), np.asarray(val_lab, dtype=np.float32).reshape(-1)
        p64, l64 = pred.astype(np.float64), lab.astype(np.float64)
        # mean over the [n, n] broadcast of (p_i - l_j)^2, without materialising it
        loss = np.float32((p64 ** 2).mean() - 2.0 * p64.mean() * l64.mean() + (l64 ** 2).mean())
        ss_res, ss_tot = float(((l64 - p64) ** 2).sum()), float(((l64 - l64.mean()) ** 2).sum())
        r2 = (1.0 - ss_res / ss_tot) * 100 if ss_tot > 0 else float("nan")
        pc, lc = p64 - p64.mean(), l64 - l64.mean()
        den = float(np.sqrt((pc ** 2).sum() * (lc ** 2).sum()))
        r = float((pc * lc).sum()) / den * 100 if den > 0 else float("nan")
        return loss, pred.copy(), {"r2": r2, "r": r}
    lab = np.asarray(val_lab).astype(np.int64).reshape(-1)
    zmax = z.max(axis=1, keepdims=True)
    lse = (zmax[:, 0] + np.log(np.exp(z - zmax).sum(axis=1))).astype(np.float32)
    loss = np.float32((lse - z[np.arange(z.shape[0]), lab]).astype(np.float64).mean())
    score, pred = _softmax1(z), z.argmax(axis=1)
    acc = float((pred == lab).mean()) * 100
    try:
        auc, ap = (100 * v for v in binary_auc_ap(lab, score))
    except ValueError:
        auc = ap = float("nan")
    return loss, score, {"acc": acc, "auc": auc, "ap": ap}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-o", "--output", required=True)
    parser.add_argument("-c", "--config", required=True)
    parser.add_argument("--exp-id", required=True)
    parser.add_argument("-m", "--meta", required=True)
    parser.add_argument("-d", "--npy-dir", required=True)
    parser.add_argument("--fold", type=int, required=True)
    parser.add_argument("--binsizes", nargs="+", type=int, default=[2000, 500, 100])
    parser.add_argument("--regression", action="store_true", default=False)
    parser.add_argument("--use-wandb", action="store_true", default=False)
    parser.add_argument("--timing", action="store_true", default=os.environ.get("CF_TRAIN_TIMING") == "1",
                        help="print a wall-clock breakdown of the run (store load, set-up, per epoch: steps / validation / metrics / checkpoint); "
                             "adds a device synchronisation at each of those boundaries")
    parser.add_argument("--store", default=None, help="packed store written by `python -m chromoformer_amd.pack` "
                        "(default: <npy-dir>/chromoformer.cfstore when it exists and matches the configuration)")
    parser.add_argument("--dp-shard", choices=["static", "global"], default="static",
                        help="data-parallel runs: `static` = every rank owns (and loads) 1/world of the training genes and draws its "
                             "batches from them; `global` = every rank holds the whole split and takes its slice of each global batch "
                             "of the epoch permutation (reproduces the single-process batch composition exactly)")
    args = parser.parse_args(argv)

    if args.use_wandb is False:
        os.environ["WANDB_MODE"] = "disabled"
    wandb = _wandb(args.use_wandb)
    with open(args.config) as f:
        config = yaml.safe_load(f)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:      # (multi-process GPU work on this platform needs dmabuf IPC; in place before the HIP runtime starts)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the same two hooks as bench.py: CF_SHARE_DEVICE=1 puts every rank on device 0 and CF_DIST_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device) -- how the multi-process entrypoint is exercised on a one-GPU box
    if os.environ.get("CF_SHARE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("CF_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD
    say = print if rank == 0 else (lambda *a, **k: None)
    import time
    marks = [("start", time.perf_counter())]

    def mark(what):      # --timing: wall clock between named boundaries (device drained first, so that a phase owns its GPU work)
        if args.timing:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            marks.append((what, time.perf_counter()))
    say(config)

    config["exp_id"] = args.exp_id
    seed, num_epoch, bsz, gamma = config["seed"], config["num_epoch"], config["bsz"], config["gamma"]
    i_max, w_prom, w_max, n_feats = config["i_max"], config["w_prom"], config["w_max"], config["n_feats"]
    d_emb = config["embed"]["d_model"]

    if rank == 0:
        # a marker (or a half-written file) of an EARLIER run to the same output must not vouch for this one
        for stale in (args.output + ".done", args.output + ".tmp"):
            if os.path.exists(stale):
                os.remove(stale)
    seed_everything(seed)
    wandb.init(project="chromoformer-refactoring", entity="dohlee", group=args.exp_id)
    wandb.config.update(args)
    wandb.config.update(config)

    meta = pd.read_csv(args.meta).sample(frac=1, random_state=seed).reset_index(drop=True)
    if args.regression and "expression" not in meta.columns:
        raise ValueError("`expression` column is required for training ChromoformerRegression model.")
    say("Target genes:", len(set(meta.gene_id.unique())))
    qs = [meta[meta.split == k].gene_id.tolist() for k in (1, 2, 3, 4)]
    train_genes = qs[(args.fold + 0) % 4] + qs[(args.fold + 1) % 4] + qs[(args.fold + 2) % 4]
    val_genes = qs[(args.fold + 3) % 4]
    say(len(train_genes), len(val_genes))

    from . import pack
    packed = pack.find(args.npy_dir, args.store, args.binsizes, i_max, w_prom, w_max, n_feats, train_genes + val_genes, meta=meta)
    say("packed store:", packed.path if packed is not None else "none (binning the raw .npy files): %s" % (pack.last_skip_reason or "not looked for"))

    def store_of(genes):
        if packed is not None:          # a row gather out of the memory-mapped store, resident in HBM afterwards
            return packed.store(genes, device=torch.device("cuda", local), regression=args.regression)
        ds = ChromoformerDataset(args.meta, args.npy_dir, genes, n_feats, i_max, args.binsizes, w_prom, w_max,
                                 regression=args.regression)
        # raw fp16 signals are binned on the GPU (cf_bin_regions) and the split stays resident in HBM
        return GeneStore(ds, progress=(rank == 0), device=torch.device("cuda", local), resident=True)

    # Data parallel: a rank loads what it will touch.  Training genes: its static shard (every world-th gene), or the whole
    # split when the single-process batch composition is to be reproduced; validation genes: its contiguous slice.
    static = world > 1 and args.dp_shard == "static"
    train_store = store_of(static_shard(train_genes, rank, world) if static else train_genes)
    n_val = len(val_genes)
    per_val = (n_val + world - 1) // world
    val_lo, val_hi = min(n_val, rank * per_val), min(n_val, (rank + 1) * per_val)
    val_store = store_of(val_genes[val_lo:val_hi])
    mark("store load (train + validation splits resident in HBM)")
    label_of = {r["gene_id"]: (np.log2(r["expression"] + 1) if args.regression else r["label"]) for r in meta.to_dict("records")}
    val_labels = torch.tensor([label_of[g] for g in val_genes], dtype=torch.float32 if args.regression else torch.int64)

    Model = ChromoformerRegressor if args.regression else ChromoformerClassifier
    model = Model(n_feats, d_emb, config["d_head"], config["embed"], config["pairwise_interaction"], config["regulation"],
                  binsizes=args.binsizes, seed=42, i_max=i_max, w_max=w_max, max_batch=bsz)
    model.cuda(local)
    criterion = nn.MSELoss() if args.regression else nn.CrossEntropyLoss()
    # (single GPU: the step is a replayed graph + two eager launches; data parallel: three graphs with collectives between them replay 4 % SLOWER than the
    #  same launches issued eagerly -- 0.578 against 0.557 ms on a one-rank RCCL group, profiles/r05j_bench_dp.json --, so the ranks issue eagerly)
    trainer = Trainer(model, lr=float(config["lr"]), gamma=gamma, world_size=world, process_group=pg, use_graph=(world == 1 and pg is None))
    feed = EpochFeed(model, train_store, bsz)          # batches are gathered from the resident split inside the step graph
    if world > 1:                                      # one line per rank: what it loaded (stderr; stdout stays rank 0's, as in the reference)
        n_steps = (len(train_genes) // world // bsz) if static else (len(train_genes) // (world * bsz))
        print("[rank %d/%d] dp-shard %s: train store %d of %d genes, validation slice %d of %d genes, %d steps per epoch"
              % (rank, world, args.dp_shard, len(train_store), len(train_genes), len(val_store), n_val, n_steps), file=sys.stderr, flush=True)

    writer = CheckpointWriter(model) if rank == 0 else None
    mark("model, trainer, feed")
    val_score = val_label = val_loss = None
    for epoch in range(1, num_epoch):
        perm = epoch_permutation(len(train_genes))            # the same draws on every rank
        if static:
            batches = static_epoch_batches(perm, rank, world, bsz)
        else:
            batches = shard_indices(perm, rank, world, bsz, drop_last=True)
        train_epoch(trainer, feed, batches,
                    lambda lo, la, ls: _report_train(say, wandb, epoch, float(ls.numpy().mean()), trainer.lr, lo, la, args.regression),
                    after_issue=writer.go if writer is not None else None)      # the previous epoch's checkpoint is written under this one's steps
        mark("epoch %d: %d steps (permutation, begin_epoch, steps, running metrics)" % (epoch, len(batches)))

        # validation (sharded over ranks, gathered on every rank)
        _draw_loader_seed()                                         # the val DataLoader's base seed draw
        val_out, val_lab = _validate(model, trainer, val_store, n_val, bsz, world), val_labels.clone()
        mark("epoch %d: validation forward (%d genes)" % (epoch, n_val))
        val_label = val_lab.numpy()
        v_loss, val_score, vm = validation_metrics(val_out.numpy(), val_label, args.regression)
        val_loss = torch.tensor(v_loss)                             # (a 0-dim float32 tensor, as criterion(...) returns)
        if args.regression:
            val_r2, val_r = vm["r2"], vm["r"]
            say(f"Validation loss={val_loss:.4f}, r2={val_r2:.4f}, r={val_r:.4f}")
            wandb.log({"val/loss": val_loss, "val/r2": val_r2, "val/r": val_r})
            ckpt = {"net": None, "optimizer": None, "epoch": epoch, "last_val_loss": val_loss, "last_val_r2": val_r2,
                    "val_score": val_score, "val_label": val_label}
        else:
            val_acc, val_auc, val_ap = vm["acc"], vm["auc"], vm["ap"]
            say(f"Validation loss={val_loss:.4f}, acc={val_acc:.4f}, auc={val_auc:.4f}, ap={val_ap:.4f}")
            wandb.log({"val/loss": val_loss, "val/acc": val_acc, "val/auc": val_auc, "val/ap": val_ap, "val/epoch": epoch})
            ckpt = {"net": None, "optimizer": None, "epoch": epoch, "last_val_loss": val_loss, "last_val_auc": val_auc,
                    "val_score": val_score, "val_label": val_label}
        mark("epoch %d: validation metrics" % epoch)
        if rank == 0:
            # three flat device-to-host copies here; the reference's per-tensor layout, torch.save to a temporary name and the rename
            # (a job killed mid-write never leaves a truncated file under the final name) on a worker thread under the next epoch
            writer.submit(args.output, ckpt, trainer.lr, float(config["lr"]))
            if os.environ.get("CF_CKPT_SYNC") == "1":      # (A/B switch: the save on the critical path, as the reference has it)
                writer.wait()
        mark("epoch %d: checkpoint hand-off (flat copies to pinned memory; layout + torch.save on a worker thread)" % epoch)
        trainer.scheduler_step()

    if rank == 0:
        writer.wait()                                 # the last checkpoint is on disk (or its error is raised) before the run vouches for it
        mark("last checkpoint on disk")
        with open(args.output + ".done", "w") as f:
            f.write("epochs %d\n" % max(0, num_epoch - 1))
    if args.timing and rank == 0:
        total = marks[-1][1] - marks[0][1]
        say("wall-clock breakdown (%.2f s):" % total)
        for (_, t0), (what, t1) in zip(marks[:-1], marks[1:]):
            say("  %8.3f s  %5.1f %%  %s" % (t1 - t0, 100 * (t1 - t0) / total, what))
    if val_loss is not None:
        key = "last_val_r2" if args.regression else "last_val_auc"
        wandb.summary.update({"last_val_loss": val_loss, key: ckpt[key]})
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def train_epoch(trainer, feed, batches, report=None, every=10, after_issue=None):
    """The optimisation steps of one epoch (train.py:171-232): per step four host calls (graph replay, cf_rider_arm, the trunk's backward launch with its riders, reduction + AdamW) and no
    host synchronisation.  Every `every` steps an event is recorded; `report(logits, labels, losses)` receives each window
    of `every` steps as soon as its event has completed (polled, never waited for inside the epoch), read from the feed's
    pinned-host logs: the host-side metrics never leave the GPU idle, the printed lines are the reference's, a little late."""
    feed.begin_epoch(batches, trainer.stream)
    pending = []                                          # (first step, last step + 1, event after the last step)
    for k in range(1, len(batches) + 1):
        trainer.step(feed.slot)
        if report is not None and k % every == 0:
            pending.append((k - every, k, trainer.stream.record_event()))
            while pending and pending[0][2].query():
                lo, hi, _ = pending.pop(0)
                report(*feed.window(lo, hi))
    if after_issue is not None:                           # every step of the epoch is queued: background host work may take the interpreter
        after_issue()
    for lo, hi, ev in pending:
        ev.synchronize()
        report(*feed.window(lo, hi))


def binary_auc_ap(label, score):
    """(ROC AUC, average precision) of a binary problem, the definitions of sklearn.metrics.roc_auc_score /
    average_precision_score (tied scores share a threshold) in a few numpy passes: the running train metrics are computed
    every tenth step, and three sklearn calls (~25 ms of input validation on 640 samples) would hold up a loop whose step
    takes 0.75 ms.  Raises ValueError for a single-class window, as sklearn does."""
    y = np.asarray(label).astype(bool).ravel()
    s = np.asarray(score, dtype=np.float64).ravel()
    n_pos = int(y.sum())
    n_neg = y.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("only one class present")
    order = np.argsort(-s, kind="stable")
    s, y = s[order], y[order]
    last = np.r_[np.nonzero(np.diff(s))[0], y.size - 1]          # last index of every group of tied scores
    tp = np.cumsum(y)[last].astype(np.float64)
    fp = (last + 1) - tp
    tpr, fpr = np.r_[0.0, tp / n_pos], np.r_[0.0, fp / n_neg]
    auc = float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))
    precision, recall = tp / (last + 1), tp / n_pos
    ap = float(np.sum(np.diff(np.r_[0.0, recall]) * precision))
    return auc, ap


def _softmax1(logits):
    """Column 1 of a row softmax, in numpy (float32, maximum subtracted, as torch's CPU softmax does)."""
    z = np.asarray(logits, dtype=np.float32)
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e[:, 1] / e.sum(axis=1)


def _report_train(say, wandb, epoch, batch_loss, lr, out, label, regression):
    # numpy only: a torch CPU op wakes torch's intra-op thread pool, whose workers then spin on every host core and keep the
    # HIP runtime's submission thread off the CPU -- measured +0.17 .. 0.33 ms per step on a 0.72 ms step (tools/event_probe.py)
    out, label = out.numpy(), label.numpy()
    if regression:
        pred, lab = out.reshape(-1).astype(np.float64), label.reshape(-1).astype(np.float64)
        ss_res, ss_tot = float(((lab - pred) ** 2).sum()), float(((lab - lab.mean()) ** 2).sum())
        r2 = (1.0 - ss_res / ss_tot) * 100 if ss_tot > 0 else float("nan")
        pc, lc = pred - pred.mean(), lab - lab.mean()
        den = float(np.sqrt((pc ** 2).sum() * (lc ** 2).sum()))
        r = float((pc * lc).sum()) / den * 100 if den > 0 else float("nan")
        say(f"E{epoch} {batch_loss:.4f}, lr={lr}, r2={r2:.4f}, r={r:.4f}")
        wandb.log({"train/loss": batch_loss, "train/r2": r2, "train/r": r})
    else:
        score, pred = _softmax1(out), out.argmax(axis=1)
        lab = label
        acc = float((pred == lab).mean()) * 100
        try:
            auc, ap = (100 * v for v in binary_auc_ap(lab, score))
        except ValueError:          # a window with a single class
            auc = ap = float("nan")
        say(f"E{epoch} {batch_loss:.4f}, lr={lr}, acc={acc:.4f}, auc={auc:.4f}, ap={ap:.4f}")
        wandb.log({"train/loss": batch_loss, "train/acc": acc, "train/auc": auc, "train/ap": ap})


def _validate(model, trainer, store, n_total, bsz, world):
    """Forward over the rank's slice of the validation genes (`store` holds exactly that slice: ceil(n_total / world)
    consecutive genes per rank; the tail batch is kept, train.py:140); returns the CPU logits [n_total, n_out] of ALL
    validation genes in dataset order on every rank."""
    per = (n_total + world - 1) // world
    mine = trainer.evaluate_store(store, bsz)         # one cf_gather_batch + cf_forward per batch, from the resident slice
    if world > 1:
        pad = torch.zeros(per, model.n_out, device=model._device)
        pad[: mine.shape[0]] = mine
        parts = [torch.zeros_like(pad) for _ in range(world)]
        torch.distributed.all_gather(parts, pad)
        mine = torch.cat([p[: max(0, min(n_total, (r + 1) * per) - min(n_total, r * per))] for r, p in enumerate(parts)])
    return mine.cpu()


if __name__ == "__main__":
    sys.exit(main())
