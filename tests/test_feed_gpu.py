"""cf_gather_batch / cf_record_step (the batch gather and the step log inside the graph) and the training loop built on them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_genes=300, bsz=8, regime="realistic"):
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import EpochFeed, Trainer
    from chromoformer_amd.synth import synthetic_store
    dev = torch.device("cuda", 0)
    model = ChromoformerClassifier(seed=42, max_batch=bsz).cuda(0)
    store = synthetic_store(n_genes, dev, seed=5, regime=regime)
    trainer = Trainer(model, lr=3e-5)
    return model, store, trainer, EpochFeed(model, store, bsz)


def test_gather_walks_the_epoch_order_bit_exactly():
    import ctypes as C
    from chromoformer_amd import _lib
    model, store, trainer, feed = _setup()
    rng = np.random.default_rng(0)
    batches = [rng.choice(len(store), size=8, replace=False).tolist() for _ in range(5)]
    feed.begin_epoch(batches, trainer.stream)
    L, slot = _lib.lib(), feed.slot
    for k, idx in enumerate(batches):
        with torch.cuda.stream(trainer.stream):
            _lib.check(L.cf_gather_batch(model._handle, C.byref(feed.struct), feed.order.data_ptr(), feed.cursor.data_ptr(),
                                         C.byref(slot.struct), slot.label.data_ptr(), trainer.stream.cuda_stream), "cf_gather_batch")
        trainer.stream.synchronize()
        assert feed.cursor.tolist() == [k + 1, 5, 0, 0]
        ii = torch.tensor(idx, device=store.freq.device)
        for r in range(3):
            assert torch.equal(slot.pf[r], store.pf[r][ii]) and torch.equal(slot.cf[r], store.cf[r][ii])
            assert torch.equal(slot.pm[r], store.pm[r][ii]) and torch.equal(slot.cm[r], store.cm[r][ii])
            assert torch.equal(slot.im[r], store.im[ii])
        assert torch.equal(slot.freq, store.freq[ii]) and torch.equal(slot.label, store.label[ii])


def test_a_step_past_the_epoch_or_a_bad_gene_index_touches_nothing():
    """The gather and the step log are bounded on the device (cursor[1] batches, store.n_genes genes) and on the host
    (Trainer.step refuses an exhausted feed): one call too many neither reads the order / the store out of bounds nor writes
    past the pinned host logs."""
    import ctypes as C
    from chromoformer_amd import _lib
    model, store, trainer, feed = _setup(n_genes=40)
    batches = [list(range(8)), list(range(8, 16))]
    feed.begin_epoch(batches, trainer.stream)
    L, slot, st = _lib.lib(), feed.slot, trainer.stream.cuda_stream

    def gather():
        with torch.cuda.stream(trainer.stream):
            _lib.check(L.cf_gather_batch(model._handle, C.byref(feed.struct), feed.order.data_ptr(), feed.cursor.data_ptr(),
                                         C.byref(slot.struct), slot.label.data_ptr(), st), "cf_gather_batch")
        trainer.stream.synchronize()

    gather()
    gather()
    assert feed.cursor.tolist() == [2, 2, 0, 0]
    before = [t.clone() for t in slot.cf] + [slot.freq.clone(), slot.label.clone()]
    log_before = feed.logits_log.clone()
    gather()                                                    # a third batch that does not exist
    with torch.cuda.stream(trainer.stream):
        _lib.check(L.cf_record_step(model._handle, feed.cursor.data_ptr(), slot.logits.data_ptr(), slot.label.data_ptr(), slot.loss.data_ptr(),
                                    slot.B, feed.logits_log.data_ptr(), feed.labels_log.data_ptr(), feed.loss_log.data_ptr(), st), "cf_record_step")
    trainer.stream.synchronize()
    assert feed.cursor.tolist() == [2, 2, 1, 0] and feed.check() == 1
    assert all(torch.equal(a, b) for a, b in zip(before, [t for t in slot.cf] + [slot.freq, slot.label]))
    assert torch.equal(feed.logits_log, log_before)
    with pytest.raises(RuntimeError, match="error flags 1"):
        feed.begin_epoch(batches, trainer.stream)
    # a gene index outside the store: skipped and flagged
    feed.begin_epoch([[0, 1, 2, 3, 4, 5, 6, 40]], trainer.stream)
    gather()
    assert feed.check() == 2
    ii = torch.arange(7, device=store.freq.device)
    assert torch.equal(slot.freq[:7], store.freq[ii])
    feed.cursor[2] = 0
    # host-side guard of the training loop
    feed.begin_epoch(batches, trainer.stream)
    trainer.step(slot)
    trainer.step(slot)
    with pytest.raises(RuntimeError, match="exhausted"):
        trainer.step(slot)
    torch.cuda.synchronize()
    assert feed.check() == 0


def test_feed_steps_equal_staged_steps_and_log_every_step():
    """Training through the feed (gather + record inside the graph) is bit-identical to staging the same batches by hand."""
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    from chromoformer_amd.train import train_epoch
    model, store, trainer, feed = _setup()
    rng = np.random.default_rng(1)
    batches = [rng.choice(len(store), size=8, replace=False).tolist() for _ in range(12)]
    windows = []
    train_epoch(trainer, feed, batches, lambda lo, la, ls: windows.append((lo.clone(), la.clone(), ls.clone())), every=5)
    torch.cuda.synchronize()
    ref = ChromoformerClassifier(seed=42, max_batch=8).cuda(0)
    tr2 = Trainer(ref, lr=3e-5)
    logits, losses = [], []
    for idx in batches:
        slot = tr2.stage(store.batch(idx))
        lo, ls = tr2.step(slot)
        tr2.stream.synchronize()
        logits.append(lo.cpu().clone())
        losses.append(ls.cpu().clone())
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref.state_dict()[k]), k
    assert len(windows) == 2 and windows[0][0].shape == (40, 2)
    assert torch.equal(torch.cat([w[0] for w in windows]), torch.cat(logits[:10]))
    assert torch.equal(torch.cat([w[2] for w in windows]), torch.cat(losses[:10]))
    want = torch.cat([store.label[torch.tensor(b, device=store.label.device)].cpu() for b in batches[:10]])
    assert torch.equal(torch.cat([w[1] for w in windows]), want)


@pytest.mark.parametrize("use_graph", [True, False])
def test_a_forward_pass_between_two_fed_steps_does_not_move_the_feed(use_graph):
    """The pre-gathered feed keeps 'a batch is waiting, the next trunk launch advances the cursor' on the handle.  A forward pass over
    ANOTHER batch in between (validation mid-epoch through the public Trainer API) must leave that state to the step it was queued
    for: consumed there, the replayed step graph advanced the cursor a second time and the epoch silently skipped a batch."""
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import EpochFeed, Trainer
    from chromoformer_amd.synth import synthetic_store
    dev = torch.device("cuda", 0)
    store = synthetic_store(120, dev, seed=5, regime="realistic")
    val = synthetic_store(24, dev, seed=6, regime="realistic")
    rng = np.random.default_rng(3)
    batches = [rng.choice(len(store), size=8, replace=False).tolist() for _ in range(6)]

    def run(interleave):
        model = ChromoformerClassifier(seed=42, max_batch=8).cuda(0)
        trainer = Trainer(model, lr=3e-5, use_graph=use_graph)
        feed = EpochFeed(model, store, 8)
        feed.begin_epoch(batches, trainer.stream)
        outs = []
        for k in range(len(batches)):
            lo, ls = trainer.step(feed.slot)
            trainer.stream.synchronize()
            outs.append((lo.cpu().clone(), ls.cpu().clone()))
            if interleave and k in (1, 2, 4):
                trainer.evaluate_store(val, 8)
        torch.cuda.synchronize()
        assert feed.check() == 0 and int(feed.cursor[0].item()) == len(batches)
        return outs, {k: v.clone() for k, v in model.state_dict().items()}

    plain, sd0 = run(False)
    mixed, sd1 = run(True)
    for (a, b), (c, d) in zip(plain, mixed):
        assert torch.equal(a, c) and torch.equal(b, d)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
