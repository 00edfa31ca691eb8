"""Step engine on the GPU: hipGraph replay (one graph per step; two with the bucketed RCCL all-reduce of the
data-parallel path, the early bucket reduced on a side stream), and HIP-event timing nodes inside the graphs, must
all give the very same bits as the eager single-stream step."""
import socket

import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu
B = 8


def _run(steps=3, **kw):
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    trainer = Trainer(model, lr=3e-5, **kw)
    slots = [trainer.stage(orc.synthetic_batch(B, seed=7 + i, regime="realistic")) for i in range(2)]
    losses = []
    for i in range(steps):
        _, loss = trainer.step(slots[i % 2])
        with torch.cuda.stream(trainer.stream):
            losses.append(loss.clone())
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sd["<exp_avg>"], sd["<exp_avg_sq>"] = model._mflat.cpu().clone(), model._vflat.cpu().clone()      # the optimiser state as well
    if kw.get("keep_grads") or not trainer.fuse_opt:
        sd["<grads>"] = model._gflat.cpu().clone()
    return sd, [float(x) for x in losses]


def test_graph_replay_and_event_timing_are_bit_identical_to_the_eager_step():
    ref, ref_loss = _run(use_graph=False)
    for kw in (dict(use_graph=True), dict(use_graph=True, timed_kernel="k_reg_bwd"), dict(use_graph=False, timed_kernel="k_wgrad"),
               dict(use_graph=True, opt_in_graph=True), dict(use_graph=True, opt_in_graph=True, overlap_opt=True),
               dict(use_graph=False, opt_in_graph=True, overlap_opt=True)):
        got, loss = _run(**kw)
        assert loss == ref_loss, kw
        for k in ref:
            assert torch.equal(ref[k], got[k]), (kw, k)


def test_round3_schedules_and_switches_are_bit_identical():
    """The merged launch (one bucket's reduction beside the other bucket's AdamW), the reduction overlapped with the trunk backward on
    the side stream, and the library switches that only move work between launches -- all the same arithmetic in the same order."""
    import os
    ref, ref_loss = _run(use_graph=False, merge_opt=False)
    for kw in (dict(use_graph=False, merge_opt=True, fuse_opt=False), dict(use_graph=True, merge_opt=True, fuse_opt=False),
               dict(use_graph=True, merge_opt=True, overlap_reduce=True), dict(use_graph=False, merge_opt=True, overlap_reduce=True),
               # AdamW in the epilogue of the gradient reductions (the default single-GPU step), with and without the gradient stores
               dict(use_graph=False, fuse_opt=True), dict(use_graph=True, fuse_opt=True), dict(use_graph=True, fuse_opt=True, keep_grads=True),
               dict(use_graph=True, fuse_opt=True, fuse_one=False), dict(use_graph=False, fuse_opt=True, fuse_one=False, keep_grads=True),
               # without / with riders in the trunk's backward launch (B = 8: 16 teams; 1,008 uniform tiles)
               dict(use_graph=True, fuse_opt=True, rider_tiles=0), dict(use_graph=True, fuse_opt=True, rider_tiles=100, keep_grads=True),
               dict(use_graph=False, fuse_opt=True, rider_tiles=5000)):
        got, loss = _run(**kw)
        assert loss == ref_loss, kw
        for k in got:
            assert torch.equal(ref[k], got[k]), (kw, k)
    for env in ({"CF_DEFER_RETILE": "0"}, {"CF_XCD_REDUCE": "1"}):      # read when the model is constructed
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got, loss = _run(use_graph=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert loss == ref_loss, env
        for k in got:
            assert torch.equal(ref[k], got[k]), (env, k)


def test_graph_embedded_events_time_every_replay():
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    tr = Trainer(model, timed_kernel="k_reg_bwd")
    slot = tr.stage(orc.synthetic_batch(B, seed=3, regime="dense"))
    tr.step(slot)
    tr.timing_reset()
    for _ in range(7):
        tr.step(slot)
    ms, n = tr.timing_read()
    assert n == 7
    assert 0.01 < ms / n < 5.0          # a Regulation backward launch at B = 8 takes a fraction of a millisecond
    assert tr.timing_read()[1] == 7     # reading does not reset


def test_grad_buckets_partition_the_active_range():
    import ctypes as C
    from chromoformer_amd import ChromoformerClassifier, _lib
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    tr = Trainer(model)
    pe, reg = tr._buckets[_lib.BUCKET_PE], tr._buckets[_lib.BUCKET_REG]
    assert pe.data_ptr() == model.active_grads().data_ptr()
    assert reg.data_ptr() == pe.data_ptr() + 4 * pe.numel()
    assert pe.numel() + reg.numel() == model.active_grads().numel()
    names = [d["name"] for d in model._table if d["trainable"] and d["offset"] >= pe.numel()]
    assert names and all(n.startswith(("regulation.", "fc_head.")) for n in names)
    assert reg.numel() > 3 * pe.numel()          # the early bucket carries most of the bytes
    # ... and the Regulation + head bucket in halves: [lower layers | upper layers + head], adjacent, the upper half named by layer index
    L, off, n = _lib.lib(), C.c_longlong(), C.c_longlong()
    half = L.cf_reg_halves(model._handle)
    assert half == 3
    _lib.check(L.cf_grad_bucket(model._handle, _lib.BUCKET_REG_LO, C.byref(off), C.byref(n)), "cf_grad_bucket")
    lo = (off.value, n.value)
    _lib.check(L.cf_grad_bucket(model._handle, _lib.BUCKET_REG_HI, C.byref(off), C.byref(n)), "cf_grad_bucket")
    hi = (off.value, n.value)
    assert lo[0] == pe.numel() and hi[0] == lo[0] + lo[1] and hi[0] + hi[1] == model.active_grads().numel()
    for d in model._table:
        if not d["trainable"] or not d["name"].startswith(("regulation.", "fc_head.")):
            continue
        upper = d["name"].startswith("fc_head.") or int(d["name"].split(".transformer.layers.")[1].split(".")[0]) >= half
        assert (d["offset"] >= hi[0]) == upper, d["name"]


def test_one_rank_rccl_all_reduce_in_the_step():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ref, ref_loss = _run(use_graph=True)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got, loss = _run(use_graph=True, world_size=1, process_group=dist.group.WORLD)
        got2, loss2 = _run(use_graph=False, world_size=1, process_group=dist.group.WORLD, timed_kernel="k_reg_bwd")
        got3, loss3 = _run(use_graph=True, world_size=1, process_group=dist.group.WORLD, opt_in_graph=True, overlap_opt=True)
        # serialised schedule: both all-reduces behind the whole backward pass (Trainer(overlap_allreduce=False) / CF_DP_OVERLAP=0)
        got4, loss4 = _run(use_graph=True, world_size=1, process_group=dist.group.WORLD, overlap_allreduce=False)
        got5, loss5 = _run(use_graph=False, world_size=1, process_group=dist.group.WORLD, overlap_allreduce=False, opt_in_graph=True)
        # the schedules above send the Regulation + head bucket in two halves (round 5: the upper layers' gradients are on the wire while the
        # lower half of the Regulation backward runs); the two-bucket schedule of rounds 2-4, overlapped and serialised, eager and replayed
        more = [_run(use_graph=g, world_size=1, process_group=dist.group.WORLD, overlap_allreduce=o, dp_halves=False)
                for g, o in ((True, True), (False, True), (True, False))]
        more.append(_run(use_graph=False, world_size=1, process_group=dist.group.WORLD, overlap_allreduce=True))      # halves, eager
        # (round 6: that one reduces the two Regulation halves' weight gradients on the side stream, beside the next backward launch; the round-5
        #  form with every reduction on the main stream: dp_side_reduce=False / CF_DP_SIDE_REDUCE=0)
        more.append(_run(use_graph=False, world_size=1, process_group=dist.group.WORLD, dp_side_reduce=False))
        more.append(_run(use_graph=False, world_size=1, process_group=dist.group.WORLD, dp_side_reduce=True, dp_early_opt=False))
        # (all of the above step the Regulation + head range on the side stream, straight behind its last all-reduce; the round-4 form, on the main
        #  stream behind everything: dp_early_opt=False / CF_DP_EARLY_OPT=0)
        more += [_run(use_graph=g, world_size=1, process_group=dist.group.WORLD, dp_early_opt=False, dp_halves=hv) for g, hv in ((True, True), (False, False))]
    finally:
        dist.destroy_process_group()
    assert loss == ref_loss and loss2 == ref_loss and loss3 == ref_loss and loss4 == ref_loss and loss5 == ref_loss
    for k in ref:
        assert torch.equal(ref[k], got[k]) and torch.equal(ref[k], got2[k]) and torch.equal(ref[k], got3[k]), k
        assert torch.equal(ref[k], got4[k]) and torch.equal(ref[k], got5[k]), k
    for sd, ls in more:
        assert ls == ref_loss
        for k in ref:
            assert torch.equal(ref[k], sd[k]), k


def test_roofline_of_the_halved_regulation_backward_agrees_with_the_single_launch():
    """Data parallel in halves launches the Regulation backward twice per step under one timing key; the flop count is the whole stack's.  The
    roofline entry is normalised per step: it must agree with the single-launch (one-GPU) entry, not read twice its value."""
    import torch.distributed as dist
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    batch = orc.synthetic_batch(B, seed=7, regime="dense")

    def measure(**kw):
        model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
        tr = Trainer(model, lr=3e-5, use_graph=False, timed_kernel="k_reg_bwd", **kw)
        slot = tr.stage(batch)
        for _ in range(20):
            tr.step(slot)
        torch.cuda.synchronize()
        tr.timing_reset()
        for _ in range(40):
            tr.step(slot)
        torch.cuda.synchronize()
        ms, n = tr.timing_read()
        return tr.roofline("k_reg_bwd", ms, n, B, steps=40), tr.halves

    one, _ = measure()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dp, halves = measure(world_size=1, process_group=dist.group.WORLD)
    finally:
        dist.destroy_process_group()
    assert halves and dp["launches_per_step"] == 2 and one["launches_per_step"] == 1
    assert dp["algorithmic_gflop_per_step"] == one["algorithmic_gflop_per_step"]
    # two launches over half the layers each take a little longer than one over all of them (a second launch ramp), never half as long
    assert 0.6 * one["frac"] < dp["frac"] < 1.15 * one["frac"], (one, dp)


def test_riders_must_be_followed_by_the_fused_reduction_of_the_same_step():
    """cf_rider_arm hands part of a bucket to the next trunk launch, AdamW included: a plain reduction, or a fused one for another step,
    must be refused afterwards (the bucket would be reduced twice / stepped with two sets of scalars)."""
    import ctypes as C
    from chromoformer_amd import ChromoformerClassifier, _lib
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    tr = Trainer(model, use_graph=False, rider_tiles=64)
    slot = tr.stage(orc.synthetic_batch(B, seed=3, regime="dense"))
    tr.step(slot)                                   # a regular step with riders
    L, h, st = _lib.lib(), model._handle, tr.stream.cuda_stream
    hp = (3e-5, 0.9, 0.999, 1e-8, 0.01)
    with torch.cuda.stream(tr.stream):
        tr._seq_early(slot, st, reduce=False)
        assert L.cf_rider_arm(h, *hp, 2, 0, 64) == 0
        tr._part(slot, st, 4)                       # the trunk launch takes 64 tiles (and steps them with the scalars of step 2)
        assert L.cf_backward_reduce_part(h, slot.B, _lib.BUCKET_REG, st) != 0
        assert b"riders" in L.cf_last_error()
        assert L.cf_reduce_opt_part(h, slot.B, _lib.BUCKET_REG | _lib.BUCKET_PE, *hp, 3, 0, st) != 0      # another step's scalars
        assert L.cf_rider_arm(h, *hp, 3, 0, 64) != 0                                                      # nor a second arming
        assert L.cf_reduce_opt_part(h, slot.B, _lib.BUCKET_REG | _lib.BUCKET_PE, *hp, 2, 0, st) == 0      # the call that belongs there
    torch.cuda.synchronize()
    sd = model.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values())


def test_tiled_copies_kept_by_the_optimiser_survive_outside_writes():
    """Trainer(keep_tiled=True), the default: the fused optimiser writes the tiled copies of the Embedding + Pairwise weights and forward passes stop
    re-tiling them (one launch less per step).  Parameters written through torch between two steps (an in-place op, load_state_dict) are noticed
    by the version counters; an edit through `.data` needs model.params_changed().  Same bits as a trainer that re-tiles in every forward pass."""
    import ctypes as C
    from chromoformer_amd import ChromoformerClassifier, _lib
    from chromoformer_amd.engine import Trainer

    def run(keep, graph):
        model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
        tr = Trainer(model, lr=1e-3, keep_tiled=keep, use_graph=graph)
        assert tr.keep_tiled == keep
        slots = [tr.stage(orc.synthetic_batch(B, seed=7 + i, regime="realistic")) for i in range(2)]
        named = dict(model.named_parameters())
        losses, counts = [], []
        for i in range(7):
            torch.cuda.synchronize()      # (the edits below run on torch's default stream, the steps on the trainer's)
            if i == 2:      # an in-place op on an Embedding and on a Pairwise weight
                with torch.no_grad():
                    named["embed.100.transformer.layers.0.self_att.att.weight"].mul_(1.25)
                    named["pairwise_interaction.500.transformer.layers.1.ff.l1.weight"].add_(0.01)
            if i == 4:      # a checkpoint load
                sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
                sd["pairwise_interaction.2000.lin_proj_p.weight"] = sd["pairwise_interaction.2000.lin_proj_p.weight"] * 0.5
                model.load_state_dict(sd)
            if i == 5:      # behind torch's back: the caller has to say so
                named["embed.2000.transformer.layers.0.ff.l2.weight"].data.mul_(0.9)
                model.params_changed()
            torch.cuda.synchronize()
            _, loss = tr.step(slots[i % 2])
            with torch.cuda.stream(tr.stream):
                losses.append(loss.clone())
            nf, nb, no = C.c_int(), C.c_int(), C.c_int()
            _lib.check(_lib.lib().cf_launch_counts(model._handle, C.byref(nf), C.byref(nb), C.byref(no)), "cf_launch_counts")
            counts.append(nf.value)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        logits = tr.evaluate(slots[0])
        tr.stream.synchronize()      # (evaluate runs on the trainer's stream)
        return sd, [float(x) for x in losses], counts, logits.cpu().clone()

    ref, ref_loss, ref_counts, ref_logits = run(False, False)
    for graph in (False, True):
        got, loss, counts, logits = run(True, graph)
        assert loss == ref_loss, (graph, loss, ref_loss)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (graph, k)
        assert torch.equal(logits, ref_logits)
        if not graph:      # forward launches of an eager step: prologue + trunk + Regulation against trunk + Regulation
            assert ref_counts[1] == counts[1] + 1, (ref_counts, counts)


def test_two_trainers_with_different_tiling_modes_on_one_model():
    """keep_tiled is a mode of the model, the captured step graphs belong to a Trainer.  A second Trainer built with the mode off (bench.py's
    data-parallel proxy does that on the shared model) must not leave the first one replaying graphs that hold no re-tiling while the
    optimiser has stopped writing the tiled copies: every Trainer puts its own mode back in front of a step.  Same bits as one Trainer alone."""
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    batches = [orc.synthetic_batch(B, seed=11 + i, regime="realistic") for i in range(2)]

    def run(interleave):
        model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
        t1 = Trainer(model, lr=1e-3, use_graph=True)
        s1 = [t1.stage(b) for b in batches]
        losses = []
        def step(tr, sl):
            _, loss = tr.step(sl)
            tr.stream.synchronize()      # (the step runs on the trainer's own non-blocking stream)
            losses.append(float(loss))

        for i in range(3):
            step(t1, s1[i % 2])
        if interleave:
            t2 = Trainer(model, lr=1e-3, use_graph=True, keep_tiled=False)      # switches the model's mode off
            assert t1.keep_tiled and not t2.keep_tiled
        else:
            t2 = t1
        s2 = [t2.stage(b) for b in batches] if interleave else s1
        for i in range(3, 9):
            tr, sl = (t2, s2) if i % 2 else (t1, s1)
            step(tr, sl[i % 2])
        torch.cuda.synchronize()
        return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    l0, sd0 = run(False)
    l1, sd1 = run(True)
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_head_ride_counters_are_put_back_to_zero_between_launches(monkeypatch):
    """The head ride's arrival counters are monotonic (no in-kernel zeroing: it raced with a second process on the device) and would drift after 1.4e9
    launches; every 2^28 launches a stream-ordered memset in front of a launch puts them back.  With CF_RIDE_RESET_EVERY=2 that happens every other step,
    replayed and eager: same parameters, moments and losses, bit for bit, as without."""
    ref, ref_loss = _run(steps=6, use_graph=True)
    monkeypatch.setenv("CF_RIDE_RESET_EVERY", "2")
    for g in (True, False):
        got, loss = _run(steps=6, use_graph=g)
        assert loss == ref_loss
        for k in ref:
            assert torch.equal(ref[k], got[k]), (g, k)
    # two Trainers (two streams) driving one handle in turn: the reset is ordered against the other stream's launches as well (ride_tick records /
    # waits an event per foreign stream) -- same bits again
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    trs = [Trainer(model, lr=3e-5, use_graph=False) for _ in range(2)]
    slots = [[tr.stage(orc.synthetic_batch(B, seed=7 + i, regime="realistic")) for i in range(2)] for tr in trs]
    losses = []
    for i in range(6):
        tr = trs[i % 2]
        if i:
            tr.stream.wait_stream(trs[(i - 1) % 2].stream)         # the steps themselves are ordered by the caller (they share the model's workspace)
        _, loss = tr.step(slots[i % 2][i % 2])
        with torch.cuda.stream(tr.stream):
            losses.append(loss.clone())
    torch.cuda.synchronize()
    assert [float(x) for x in losses] == ref_loss
    sd = model.state_dict()
    for k in sd:
        assert torch.equal(ref[k], sd[k].cpu()), k
