"""Headline benchmark: genes/s of Chromoformer training (forward + loss + backward + gradient
all-reduce + AdamW), default config, bsz = 64 genes per GPU, synthetic 7-mark inputs resident
in HBM (BASELINE.json: configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1: starts its own N ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for how `roofline` is defined).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

BSZ = 64


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(step_guard_s=15.0):
    """The CPU oracle (dense PyTorch restatement of the reference, `kind: "port"`) timed on this host's cores as SURVEY.md
    section 8-d specifies: the benchmark's own workload (default config, dense synthetic batch of 64 genes), forward + backward
    + AdamW, 1 warm-up + 3 timed steps, once on all usable cores and once on 8 threads (the reference's loader count,
    README.md:136).  Guard: if the warm-up step of a setting takes longer than `step_guard_s` the timed steps of that
    setting shrink to 8 genes (the rate per gene is flat in B on the CPU), and `sample` says so."""
    from oracle import chromoformer_oracle as orc
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    full = orc.synthetic_batch(BSZ, seed=1234, regime="dense")
    small = orc.synthetic_batch(8, seed=1234, regime="dense")

    def measure(threads):
        torch.set_num_threads(threads)
        P = orc.init_params(None, 42, False)
        for t in P.values():
            t.requires_grad_(True)
        opt = orc.make_optimizer(P, 3e-5)
        t0 = time.perf_counter()
        orc.train_step(P, opt, full)                                   # warm-up (allocator, thread pool)
        warm = time.perf_counter() - t0
        batch, b, fired = (full, BSZ, False) if warm <= step_guard_s else (small, 8, True)
        t0 = time.perf_counter()
        for _ in range(3):
            orc.train_step(P, opt, batch)
        el = time.perf_counter() - t0
        return 3 * b / el, "%d threads: warm-up step of %d genes %.1f s, 3 timed steps of %d genes %.1f s -> %.2f genes/s%s" % (
            threads, BSZ, warm, b, el, 3 * b / el, " (guard fired: warm-up > %.0f s)" % step_guard_s if fired else "")

    cores = usable_cores()
    logical = os.cpu_count() or 0
    limited = " -- cgroup-limited: this process may use %d of the host's %d logical CPUs, so `cores` is NOT all physical cores" % (cores, logical) if cores < logical else ""
    rate_all, txt_all = measure(cores)
    rate_8, txt_8 = measure(min(8, cores))
    return {"value": round(rate_all, 3), "unit": "genes/s", "cores": cores, "kind": "port", "value_8_threads": round(rate_8, 3),
            "sample": "forward + backward + AdamW of the CPU oracle, default config, dense synthetic batch; %s; %s (host: %d logical CPUs, %s)%s"
                      % (txt_all, txt_8, logical, cpu, limited)}


def train_loop_rate(model, lr, steps, store_genes, regime, dev):
    """Steady-state genes/s of the SHIPPED training loop (chromoformer_amd.train.train_epoch: epoch permutation, in-graph
    batch gather from a resident split, step, AdamW, lagged metric windows with the reference's sklearn metrics every
    tenth step) over a synthetic resident store -- what `python -m chromoformer_amd.train` runs between validations."""
    from chromoformer_amd.engine import EpochFeed, Trainer
    from chromoformer_amd.synth import synthetic_store
    from chromoformer_amd.train import _report_train, epoch_permutation, train_epoch
    from chromoformer_amd.data import shard_indices
    store = synthetic_store(store_genes, dev, seed=4321, regime=regime)
    trainer = Trainer(model, lr=lr)
    feed = EpochFeed(model, store, BSZ)
    quiet = lambda *a, **k: None
    wb = type("W", (), {"log": staticmethod(quiet)})
    report = lambda lo, la, ls: _report_train(quiet, wb, 1, float(ls.numpy().mean()), trainer.lr, lo, la, False)
    torch.manual_seed(7)
    train_epoch(trainer, feed, shard_indices(epoch_permutation(len(store)), 0, 1, BSZ)[:30], report)      # warm-up (captures the graph)
    torch.cuda.synchronize()
    done, t0 = 0, time.perf_counter()
    while done < steps:
        batches = shard_indices(epoch_permutation(len(store)), 0, 1, BSZ)[: steps - done]
        train_epoch(trainer, feed, batches, report)
        done += len(batches)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": round(BSZ * done / el, 1), "unit": "genes/s", "ms_per_step": round(1e3 * el / done, 4), "steps": done,
            "store_genes": store_genes, "host_calls_per_step": ((4 if trainer.rider_tiles else 2) if trainer.fuse_one else 4) if trainer.fuse_opt else 3,
            "what": "chromoformer_amd.train.train_epoch over a resident synthetic split: the batch of step k + 1 gathered behind the tiles of step k's "
                    "reduction launch (cf_gather_batch_next; the first batch of an epoch by a launch of its own), the step log written by the trunk's backward "
                    "launch (cf_record_step_bwd), running metrics (train.py:205-232) on every 10-step window"}


def val_auroc_run(n_genes=256, epochs=3):
    """BASELINE.json's metric names `val AUROC` beside the rate: a SHORT end-to-end run of the shipped entrypoint
    (`chromoformer_amd.train.main`: raw fp16 .npy regions in the reference's on-disk format -> GPU binning -> `epochs` epochs at bsz 64
    -> validation on the held-out fold -> checkpoint) on a synthetic Roadmap-style cell line with a planted signal (expressed genes
    carry more marks; tests/synth_data.py -- the real Roadmap tracks are not in the image).  Reports the last validation AUROC the
    entrypoint printed; a number about the plumbing (labels, folds, metrics, optimiser all wired the reference's way), not about
    biology."""
    import contextlib
    import io
    import re
    import tempfile
    import yaml
    from tests.synth_data import make_dataset
    from chromoformer_amd import train as cf_train
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        meta = make_dataset(os.path.join(d, "npy"), n_genes=n_genes, seed=7)
        t_data = time.perf_counter() - t0
        cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
        cfg["bsz"], cfg["num_epoch"] = 64, epochs + 1
        yaml.safe_dump(cfg, open(os.path.join(d, "cfg.yaml"), "w"))
        buf = io.StringIO()
        t1 = time.perf_counter()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
            cf_train.main(["-o", os.path.join(d, "ck.pt"), "-c", os.path.join(d, "cfg.yaml"), "--exp-id", "bench", "-m", meta,
                           "-d", os.path.join(d, "npy"), "--fold", "0"])
        t_train = time.perf_counter() - t1
    aucs = [float(m.group(1)) for m in re.finditer(r"Validation loss=[^\n]*?auc=([0-9.]+)", buf.getvalue())]
    return {"value": aucs[-1] if aucs else None, "unit": "percent", "per_epoch": aucs, "epochs": epochs,
            "train_genes": n_genes * 3 // 4, "val_genes": n_genes // 4, "seconds": {"dataset": round(t_data, 1), "train.main": round(t_train, 1)},
            "what": "chromoformer_amd.train.main on a synthetic planted-signal cell line (tests/synth_data.make_dataset), fold 0"}


def dp_path_ms(model, batch, steps, warmup, dev):
    """Fixed overhead of the data-parallel code path, measurable on ONE GPU: the same step through a one-rank RCCL group
    (two graphs, two all-reduces of the 16.6 MB / 4.6 MB buckets -- in-place no-ops for RCCL at world 1 but launched --,
    three event waits), overlapped and serialised schedule, graph replay and eager launches.  Every leg runs at least 200
    steps whatever --steps says (a 20-step sample let one 12 ms stall double a leg in BENCH_r02.json), and an event behind
    every step gives the longest single step of the leg (`max_step_ms`), so that a stall shows as what it is."""
    import socket
    import torch.distributed as dist
    from chromoformer_amd.engine import Trainer
    if dist.is_initialized():
        return None
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    steps, warmup = max(200, steps), max(20, warmup)
    out, worst = {}, {}
    try:
        # (overlapped_eager is the schedule chromoformer_amd.train runs: since round 6 with the Regulation halves' weight-gradient reductions on the
        #  side stream, `overlapped_eager_main_reduce` is the round-5 form of it)
        for key, overlap, graph, side_red in (("overlapped", True, True, None), ("serialised", False, True, None), ("overlapped_eager", True, False, None),
                                              ("overlapped_eager_main_reduce", True, False, False), ("serialised_eager", False, False, None)):
            trainer = Trainer(model, lr=3e-5, world_size=1, process_group=dist.group.WORLD, overlap_allreduce=overlap, use_graph=graph, dp_side_reduce=side_red)
            slot = trainer.stage(batch)
            for _ in range(warmup):
                trainer.step(slot)
            torch.cuda.synchronize()
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            t0 = time.perf_counter()
            marks[0].record(trainer.stream)
            for k in range(steps):
                trainer.step(slot)
                marks[k + 1].record(trainer.stream)
            torch.cuda.synchronize()
            out[key] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
            worst[key] = round(max(marks[k].elapsed_time(marks[k + 1]) for k in range(steps)), 4)
    finally:
        dist.destroy_process_group()
    out["max_step_ms"] = worst
    out["steps"] = steps
    return out


def stress_record(steps, warmup):
    """BASELINE.json configs[3] -- i_max = 16, one 100-bp resolution over 80 kb (800-bin sequences), bsz 128 -- as SURVEY.md
    section 8-d defines it (the reference cannot run it): the roofline run of the DENSE attention core, all L x L rows,
    forward + backward, on the N = 128 * 17 sequences of a batch (cf_op_attention_fwd / _bwd, csrc/cf_attn.h).  A step is
    one forward + backward over the batch; `value` is genes/s of that core alone, `roofline` its algorithmic flops
    (4 N H L^2 dh forward, 10 N H L^2 dh backward -- the usual accounting of attention kernels: dV, dP, dQ, dK and the QK^T that a backward
    without a stored P executes again, once, in the one-pass kernel; `frac_no_recompute` leaves that re-execution out: 12 instead of 14)
    against the f32 MFMA peak, from HIP events on the launch stream; the forward and the backward also timed on their own."""
    import ctypes as C
    from chromoformer_amd import _lib
    B, S, H, L = 128, 16, 2, 800
    N = B * (S + 1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev).manual_seed(0)
    proj = torch.randn(N, L, 3 * H * 64, device=dev, generator=g)
    q, k, v = proj[:, :, :128], proj[:, :, 128:256], proj[:, :, 256:]
    d_o = torch.randn(N, L, H * 64, device=dev, generator=g)
    o = torch.empty(N, L, H * 64, device=dev)
    stats = torch.empty(N, H, L, 2, device=dev)
    dproj = torch.zeros_like(proj)
    dq, dk, dv = dproj[:, :, :128], dproj[:, :, 128:256], dproj[:, :, 256:]
    ws = torch.empty(N * H * L, device=dev)
    valid = torch.ones(N, L, dtype=torch.uint8, device=dev)
    lib = _lib.lib()
    sh = _lib.cf_attn_shape(N, H, L, L, 3 * H * 64, 3 * H * 64, 3 * H * 64, H * 64)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())

    def fwd():
        _lib.check(lib.cf_op_attention_fwd(C.byref(sh), p(q), p(k), p(v), p(valid), p(valid), None, p(o), p(stats), st), "cf_op_attention_fwd")

    def bwd():
        _lib.check(lib.cf_op_attention_bwd(C.byref(sh), p(q), p(k), p(v), p(valid), p(valid), None, p(o), p(stats), p(d_o), p(dq), p(dk), p(dv),
                                           p(ws), st), "cf_op_attention_bwd")

    def step():
        fwd()
        bwd()

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1) / n

    steps, warmup = min(steps, 20), min(warmup, 3)
    for _ in range(warmup):
        step()
    el, dev_ms = timed(step, steps)
    _, fwd_ms = timed(fwd, max(3, steps // 2))
    _, bwd_ms = timed(bwd, max(3, steps // 2))
    flops = 14.0 * N * H * L * L * 64
    ach = flops / (dev_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    try:      # HBM-side bytes of one forward + backward (the three kernels), a committed PMC constant like the default line's `roofline.traffic`
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_stress.json")))
        traffic = float(sum(tj["kernels"][k] for k in ("k_attn_fwd", "k_attn_delta", "k_attn_bwd2")))
        traffic_src = "profiles/pmc_traffic_stress.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run (committed constant, not measured in this run)"
    except (OSError, KeyError, ValueError):
        pass
    io_gb = 8 * N * L * H * 64 * 4 / 1e9                      # Q, K, V, O, dO read + dQ, dK, dV written, once each
    out = {"metric": "genes/sec through the dense attention core, forward + backward (stress config: bsz=128, i_max=16, 800 bins)",
           "value": round(B * steps / el, 1), "unit": "genes/s", "n_gpus": 1,
           "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * el / steps, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "stress config (BASELINE configs[3]): dense attention core forward + backward, all rows, "
                                  "N = 128 x 17 = %d sequences x %d heads, L = %d, dh = 64 -- attention kernels only, not a training step" % (N, H, L),
                      "parallelism": "dp1", "global_batch": B},
           "roofline": {"kernel": "k_attn_fwd + k_attn_delta + k_attn_bwd2 (dQ, dK, dV in one pass per (sequence, head), 128 keys per pass)", "bound": "mfma", "achieved": round(ach, 3), "peak": 157.3,
                        "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "frac_no_recompute": round(ach * 12.0 / 14.0 / 157.3, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_us": round(dev_ms * 1e3, 1),
                        "fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3),
                        "fwd_tflops": round(4.0 * N * H * L * L * 64 / (fwd_ms * 1e-3) / 1e12, 2),
                        "bwd_tflops_no_recompute": round(8.0 * N * H * L * L * 64 / (bwd_ms * 1e-3) / 1e12, 2),
                        "compulsory_io_gb_per_step": round(io_gb, 3),
                        "algorithmic_gflop_per_launch": round(flops / 1e9, 2)}}
    del proj, dproj, d_o, o, stats, ws, valid, q, k, v, dq, dk, dv
    torch.cuda.empty_cache()
    return out


def stress_main(args, json_out):
    json_out.write(json.dumps(stress_record(args.steps, args.warmup)) + "\n")
    json_out.flush()


def binning_record(regions=4096, reps=5):
    """The HBM-bound kernel of the path: cf_bin_regions_multi, raw fp16 [7, 40000] promoter-sized regions
    -> mean + log1p bins at 2000 / 500 / 100 bp in ONE pass over the raw bytes.  Algorithmic bytes = 2 B per raw sample + (4 * 7 + 1) B per
    output bin (features + mask byte); spot-checked against a torch expression of data.py:80-83 on the device."""
    import ctypes as C
    import numpy as np
    from chromoformer_amd import _lib
    from chromoformer_amd.data import BIN_JOB_MULTI
    dev = torch.device("cuda", 0)
    R, F, LEN = regions, 7, 40000
    raw = torch.empty(R, F, LEN, device=dev, dtype=torch.float16)
    for lo in range(0, R, 512):                       # (in pieces: the fp32 temporary of the whole array would be 4.6 GB)
        raw[lo:lo + 512] = (torch.rand(min(512, R - lo), F, LEN, device=dev) * 4).half()
    bsz = (2000, 500, 100)
    Ls = [LEN // b for b in bsz]
    feats = [torch.empty(R, L, F, device=dev) for L in Ls]
    masks = [torch.empty(R, L, dtype=torch.uint8, device=dev) for L in Ls]
    mj = np.zeros(R, dtype=BIN_JOB_MULTI)
    mj["raw"] = raw.data_ptr() + np.arange(R, dtype=np.uint64) * (F * LEN * 2)
    mj["ld"], mj["col0"], mj["ncols"], mj["flip"] = LEN, 0, LEN, np.arange(R) % 2
    for r, L in enumerate(Ls):
        mj["out"][:, r] = feats[r].data_ptr() + np.arange(R, dtype=np.uint64) * (L * F * 4)
        mj["mask"][:, r] = masks[r].data_ptr() + np.arange(R, dtype=np.uint64) * L
    tab = torch.from_numpy(mj.view(np.uint8)).to(dev)
    cb, cl = (C.c_int * 3)(*bsz), (C.c_int * 3)(*Ls)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.check(lib.cf_bin_regions_multi(C.c_void_p(tab.data_ptr()), R, F, 3, cb, cl, LEN, st), "cf_bin_regions_multi")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = R * (F * LEN * 2 + sum(L * F * 4 + L for L in Ls))
    err = 0.0
    for r, b in enumerate(bsz):
        ref = torch.log1p(raw[:8].float().reshape(8, F, Ls[r], b).mean(3)).permute(0, 2, 1)
        ref[1::2] = ref[1::2].flip(1)
        err = max(err, float((feats[r][:8] - ref).abs().max()))
    out = {"kernel": "k_bin_multi", "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
           "frac": round(nbytes / ms / 1e6 / 8000, 4), "avg_launch_us": round(ms * 1e3, 1), "algorithmic_mb_per_launch": round(nbytes / 1e6, 1),
           "workload": "%d regions x fp16 [7, 40000] -> bins of 2000 / 500 / 100 bp, one launch" % R, "max_abs_err_vs_torch": err}
    del raw, feats, masks, tab
    torch.cuda.empty_cache()
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed.run environment: start the N ranks as fresh child processes
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), relay rank 0's ONE JSON line and the
    children's exit code.  This parent never initialises the GPU (torch.cuda.device_count() only counts devices) and never replaces itself."""
    import socket
    import subprocess
    share = os.environ.get("CF_SHARE_DEVICE") == "1"
    n_dev = torch.cuda.device_count()
    if not share and n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- one rank per GPU (set CF_SHARE_DEVICE=1 CF_DIST_BACKEND=gloo "
                         "only for the one-device test mode)" % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    rc, out = 1, ""
    for attempt in range(3):                     # a fresh port if the rendezvous itself failed
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        # (the ranks get a process group of their own: if they hang -- a collective that never completes -- exactly that group is ended after
        #  `--launch-timeout-s`, nothing else; the parent itself holds no GPU state)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=args.launch_timeout_s)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, err = proc.communicate()
            sys.stderr.write(err[-20000:])
            raise SystemExit("bench.py --gpus %d: the ranks did not finish within %d s (--launch-timeout-s); their process group was ended" % (args.gpus, args.launch_timeout_s))
        sys.stderr.write(err[-20000:])
        if rc == 0 or not any(k in err for k in ("EADDRINUSE", "address already in use", "RendezvousConnectionError")):
            break
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if rc == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: the ranks printed %d JSON lines, expected 1\n" % len(lines))
        rc = 1
    for ln in lines:
        sys.stdout.write(ln + "\n")
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step as hipGraphs in the timed region (split around the event-bracketed roofline kernel: measured "
                         "+8..12 us per step).  Default: eager launches, on one GPU and under data parallelism (what chromoformer_amd.train runs there: "
                         "replay with collectives between the graphs measured 4 %% slower on the one-rank proxy, `dp_path_ms_per_step`); on one GPU the "
                         "one-graph replay time is reported beside it as `graph_replay_ms_per_step`; `config.hip_graph` says which was timed")
    ap.add_argument("--eager", "--no-graph", dest="graph", action="store_false", help="issue the launches of a step one by one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--regime", default="dense", choices=["dense", "realistic"])
    ap.add_argument("--config", default="default", choices=["default", "stress"],
                    help="default = the headline benchmark (BASELINE configs[1]); stress = the 800-bin dense attention roofline run (configs[3])")
    ap.add_argument("--train-loop-steps", type=int, default=2000, help="steps of the shipped training loop timed for the `train_loop` key (0 = skip)")
    ap.add_argument("--train-loop-genes", type=int, default=16384, help="genes in the synthetic resident split of the `train_loop` leg")
    ap.add_argument("--dp-path", action="store_true", default=None, help="also time the data-parallel code path on a one-rank RCCL group "
                    "(`dp_path_ms_per_step`; default: on for --gpus 1)")
    ap.add_argument("--no-dp-path", dest="dp_path", action="store_false")
    ap.add_argument("--roofline-kernel", default="k_reg_bwd", help="kernel timed with HIP events: k_reg_bwd (dominant), k_reg_fwd (needs --no-graph), "
                    "k_trunk_fwd, k_trunk_bwd (the centre-row trunk; bwd includes its rider tiles' time, not their flops), k_wgrad, k_adamw")
    ap.add_argument("--val-auroc", dest="val_auroc", action="store_true", default=True,
                    help="also run the entrypoint end to end on a small planted-signal cell line and report its validation AUROC (`val_auroc`; ~10 s)")
    ap.add_argument("--no-val-auroc", dest="val_auroc", action="store_false")
    ap.add_argument("--prewarm-s", type=float, default=0.5,
                    help="un-timed device pre-warm in front of the W warm-up steps: the same step issued for this many seconds of wall clock (a fresh box "
                         "runs its first ~100 ms of kernels 5 %% slower: clock / power state); reported as `prewarm_s` / `prewarm_steps`; 0 = none.  "
                         "--steps / --warmup are honoured as given and what is timed does not change")
    ap.add_argument("--launch-timeout-s", type=int, default=1500, help="--gpus N > 1 started without torch.distributed.run: end the ranks this process started after so many seconds")
    ap.add_argument("--no-extras", action="store_true", help="skip the `stress` and `binning` sub-records of the default line (~3 s)")
    args = ap.parse_args()
    # (multi-process GPU work on this platform needs dmabuf IPC: the variable must be in place before the HIP runtime starts in any rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # `python bench.py --gpus N` with no torch.distributed.run environment around it: this process starts the N ranks itself,
    # BEFORE anything touches the GPU (tests/test_dp_gpu.py::test_bench_launches_its_own_ranks)
    if args.config == "default" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])

    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 (RCCL prints a
    # version banner there when a communicator is created, libraries print warnings) is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.config == "stress":
        return stress_main(args, json_out)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch %d ranks, or run `python bench.py --gpus %d` on its own: it starts them)"
                         % (args.gpus, world, args.gpus, args.gpus))
    # test hooks (tests/test_dp_gpu.py drives the multi-process path on a one-GPU box): every rank on device 0, gloo
    # instead of RCCL (RCCL refuses two ranks on one device)
    share = os.environ.get("CF_SHARE_DEVICE") == "1"
    if share:
        local = 0
    backend = os.environ.get("CF_DIST_BACKEND", "nccl")
    if world > 1 and not share and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible to rank %d -- one rank per GPU (set CF_SHARE_DEVICE=1 CF_DIST_BACKEND=gloo "
                         "only for the one-device test mode)" % (world, torch.cuda.device_count(), rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD
        # self-check 1: every rank drives its OWN device (a launcher that hands two ranks the same LOCAL_RANK would still "scale")
        props = torch.cuda.get_device_properties(local)
        ident = "%s/%d/%s" % (os.uname().nodename, local, getattr(props, "uuid", "") or getattr(props, "pci_bus_id", ""))
        idents = [None] * world
        torch.distributed.all_gather_object(idents, ident)
        if not share and len(set(idents)) != world:
            raise SystemExit("bench.py: %d ranks on %d distinct devices: %s" % (world, len(set(idents)), idents))

    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    from chromoformer_amd.synth import synthetic_batch

    if args.graph is None:
        args.graph = False                 # eager launches: what chromoformer_amd.train runs under data parallelism (train.py:318) and what wins on one GPU
    model = ChromoformerClassifier(seed=42, max_batch=BSZ).cuda(local)
    batch = synthetic_batch(BSZ, seed=1234 + rank, regime=args.regime)
    trainer = Trainer(model, lr=3e-5, world_size=world, process_group=pg, use_graph=args.graph,
                      timed_kernel=args.roofline_kernel)
    slot = trainer.stage(batch)            # inputs resident in HBM before the timed region

    # the same W + K steps on the box as it comes (no pre-warm): what this command timed up to round 4 -- reported beside the headline as
    # `cold_ms_per_step`, never as `value`
    for _ in range(args.warmup):
        trainer.step(slot)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    tc0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(slot)
    torch.cuda.synchronize()
    cold_ms = 1e3 * (time.perf_counter() - tc0) / max(1, args.steps)
    # device pre-warm (disclosed, un-timed, wall-clock based): the driver's `--steps 20 --warmup 5` is 14 ms of GPU work on a box that has been
    # idle -- its kernels run ~5 % longer than 100 ms later (BENCH_r04: 134.3 against 128.3 us for k_reg8_bwd)
    # (under data parallelism every rank must issue the SAME number of steps -- each one is a set of all-reduces --: a count derived from rank 0's
    # measured cold step time, broadcast)
    prewarm_steps, tp0 = 0, time.perf_counter()
    fixed = None
    if world > 1:
        n = torch.tensor([int(args.prewarm_s / max(cold_ms * 1e-3, 1e-5)) // 10 * 10], device=dev, dtype=torch.int64)
        torch.distributed.broadcast(n, 0)
        fixed = min(int(n.item()), 5000)
    while args.prewarm_s > 0 and (prewarm_steps < fixed if fixed is not None else time.perf_counter() - tp0 < args.prewarm_s):
        for _ in range(10):
            trainer.step(slot)
        torch.cuda.synchronize()
        prewarm_steps += 10
    prewarm_s = time.perf_counter() - tp0 if prewarm_steps else 0.0
    for _ in range(args.warmup):
        trainer.step(slot)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    trainer.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(slot)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = t.item()
    kernel_ms, kernel_n = trainer.timing_read()

    # self-check 2 (outside the timed region): data parallelism means every rank holds the SAME parameters after the same number of
    # steps -- an all-reduce that silently did nothing (or ran on a subset of the ranks) shows here.  Bit-level checksum of the
    # trainable range: the sum of its words as integers.
    checks = None
    if world > 1:
        words = model._flat[: model._layout.n_active].view(torch.int32).to(torch.int64)
        mine = torch.stack([words.sum(), (words * (torch.arange(words.numel(), device=dev) % 65521 + 1)).sum()])
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        agree = all(bool(torch.equal(e, every[0])) for e in every)
        checks = {"param_checksum_agree": agree, "param_checksum": [int(v) for v in every[0].tolist()], "devices": idents,
                  "rccl_ranks": world if backend == "nccl" else 0, "backend": backend}
        if not agree:
            raise SystemExit("bench.py: ranks disagree on the parameters after %d steps: %s" % (args.warmup + args.steps, [e.tolist() for e in every]))

    if rank == 0:
        ms = 1e3 * el / args.steps
        value = BSZ * world * args.steps / el
        roof = trainer.roofline(args.roofline_kernel, kernel_ms, kernel_n, BSZ, steps=args.steps)
        out = {
            "metric": "genes/sec training (bsz=64, default config)", "value": round(value, 1), "unit": "genes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_s": round(prewarm_s, 3), "prewarm_steps": prewarm_steps,
            "ms_per_step": round(ms, 4), "cold_ms_per_step": round(cold_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "default config (d_emb 128, i_max 8, binsizes 2000/500/100 -> L 20/80/400), bsz 64 genes per GPU, "
                                   "%s synthetic 7-mark signals, fwd+loss+bwd+allreduce+AdamW" % args.regime,
                       "parallelism": "dp%d" % world, "global_batch": BSZ * world, "hip_graph": bool(trainer.use_graph),
                       "dp_halves": bool(trainer.halves), "dp_early_opt": bool(trainer.dp_early_opt), "dp_overlap_allreduce": bool(trainer.overlap_allreduce),
                       "dp_side_reduce": bool(trainer.dp_side_reduce),
                       "launches_per_step": sum(model.launch_counts()),      # (of the last step: counted at the launch sites, graph replays included)
                       "adamw": "in the epilogue of the gradient reductions (cf_reduce_opt_part)" if trainer.fuse_opt else "own launches"},
            "roofline": roof,
            # BASELINE.json's north_star asks for ">= 60 % HBM utilisation in the attention kernels": no kernel of this step is HBM-bound -- the centre-row
            # attention reads its 7-mark features once (the only HBM stream, 7-10.6 % of HBM bandwidth while it runs) and is bound by the matrix pipe of
            # its CU and by a chain of dependent phases (DESIGN.md sections 2, 4, 9); the bound of every kernel reported here is the f32 MFMA peak
            "attention_hbm_note": "centre-row attention is MFMA / latency-bound, not HBM-bound: 7-10.6 % of HBM bandwidth measured on the current kernels run as "
                                  "launches of their own (CF_TRUNK=0; tools/attc_bandwidth.sh -> profiles/r05k_attc_bandwidth.csv: PMC bytes / duration per launch); "
                                  "the HBM-bound kernel of the path is the binning of raw signals (profiles/*_binning.json: 70 % of 8 TB/s)",
            "loss": round(float(trainer.last_loss()), 6),
        }
        if checks is not None:
            out["dp_self_check"] = checks
        if world == 1 and not args.graph:      # the same step replayed as ONE hipGraph + AdamW (what train_epoch does), no events inside
            tg = Trainer(model, lr=3e-5)
            sg = tg.stage(batch)
            for _ in range(max(args.warmup, 50)):      # (capture + the first replays of a new graph: un-timed, as the headline leg's pre-warm)
                tg.step(sg)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                tg.step(sg)
            torch.cuda.synchronize()
            out["graph_replay_ms_per_step"] = round(1e3 * (time.perf_counter() - tg0) / args.steps, 4)
        if world == 1 and args.dp_path is not False:
            out["dp_path_ms_per_step"] = dp_path_ms(model, batch, args.steps, args.warmup, dev)
        if world == 1 and args.train_loop_steps > 0:
            out["train_loop"] = train_loop_rate(model, 3e-5, args.train_loop_steps, args.train_loop_genes, args.regime, dev)
        if world == 1 and args.val_auroc:
            out["val_auroc"] = val_auroc_run()
        if world == 1 and not args.no_extras:
            # BASELINE configs[3] and the HBM-bound kernel of the path in the SAME driver-run line (each about a second of GPU time)
            sr = stress_record(10, 2)
            out["stress"] = dict(sr["roofline"], genes_per_s=sr["value"], ms_per_step=sr["ms_per_step"], workload=sr["config"]["workload"])
            out["binning"] = binning_record()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
