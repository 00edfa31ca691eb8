"""Step engine: device-resident batch slots, hipGraph replay of the static launch sequence,
data-parallel gradient all-reduce (RCCL through torch.distributed), AdamW + StepLR.

Replaces the body of the reference's training loop (chromoformer/train.py:171-196):
per-tensor ``.cuda()`` copies, ``zero_grad``/``forward``/``backward``/``step`` and the
per-step host synchronisations.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied), or None when there is no entry."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


_TRAFFIC_SOURCE = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command "
                   "(committed constant, not measured in this run)")


class Slot:
    """One batch resident in HBM at fixed addresses (what a captured graph reads)."""

    def __init__(self, model, B):
        dev = model._device
        S, T, F = model.i_max, model.i_max + 1, model.n_feats
        self.B = B
        self.pf = [torch.zeros(B, 1, L, F, device=dev) for L in model.n_bins]
        self.cf = [torch.zeros(B, S, L, F, device=dev) for L in model.n_bins]
        self.pm = [torch.zeros(B, L, dtype=torch.uint8, device=dev) for L in model.n_bins]      # centre query row only
        self.cm = [torch.zeros(B, S, L, dtype=torch.uint8, device=dev) for L in model.n_bins]
        self.im = [torch.zeros(B, T, T, dtype=torch.uint8, device=dev) for _ in model.n_bins]
        self.freq = torch.zeros(B, T, T, device=dev)
        self.label = torch.zeros(B, dtype=torch.float32 if model.n_out == 1 else torch.int64, device=dev)
        self.logits = torch.zeros(B, model.n_out, device=dev)
        self.loss = torch.zeros(1, device=dev)
        bs = _lib.cf_batch()
        bs.B = B
        for r, L in enumerate(model.n_bins):
            bs.promoter_feats[r], bs.pcre_feats[r] = self.pf[r].data_ptr(), self.cf[r].data_ptr()
            bs.promoter_mask_row[r], bs.promoter_mask_stride[r] = self.pm[r].data_ptr(), L
            bs.pcre_mask_row[r], bs.pcre_mask_stride[r] = self.cm[r].data_ptr(), L
            bs.interaction_mask[r] = self.im[r].data_ptr()
        bs.interaction_freq = self.freq.data_ptr()
        self.struct = bs
        self.graph = None
        # The zero fills above ran on the CURRENT stream; the slot is filled and consumed on other (non-blocking) streams,
        # which do not wait for it.  Without this a late fill kernel wipes a batch that has already been copied in.
        torch.cuda.current_stream(dev).synchronize()

    def fill(self, model, d, non_blocking=True):
        """Copy a reference-layout batch dict (host or device) into the slot.

        The copies are asynchronous on the current stream, so the sources must outlive them: the slot keeps a reference
        to everything it was filled from until the next fill (callers routinely pass temporaries -- a gathered batch, a
        sliced dict -- and a freed pageable host block that is reused before hipMemcpyAsync has read it shows up as a
        silently wrong batch)."""
        self._hold = hold = []

        def put(dst, src):
            src = src.reshape(dst.shape)         # may materialise a temporary (a strided mask row, say): it is held as well
            if src.dtype != dst.dtype and not src.is_cuda:
                src = src.to(dst.dtype)          # bool -> uint8 on the host, explicitly, so that this temporary is held too
            hold.append(src)
            dst.copy_(src, non_blocking=non_blocking)

        for r, b in enumerate(model.binsizes):
            L = model.n_bins[r]
            put(self.pf[r], d["promoter_feats"][b])
            put(self.cf[r], d["pcre_feats"][b])
            pm, cm = d["promoter_pad_masks"][b], d["pcre_pad_masks"][b]
            if pm.dim() == 5:
                if model._kws[0]["n_layers"] > 1:
                    # A slot keeps the CENTRE query row of a pad mask.  That is all a one-layer Embedding reads (net.py:59); with more layers
                    # every row of the promoter mask is read (modules.py:71-73), and the all-rows path rebuilds it from the centre row as
                    # not(valid x valid) -- the dataset's form (data.py:156-161).  Any other [L, L] mask would be silently replaced: refuse it
                    # here by name; model(...) / model.pack_batch(...) hand the library the full tensor and take any bool mask.
                    full = pm[:, 0, 0].to(torch.bool)
                    valid = ~full[:, L // 2, :]
                    if not torch.equal(full, ~(valid[:, :, None] & valid[:, None, :])):
                        raise ValueError("Slot.fill: promoter_pad_masks[%d] is not of the dataset's form not(valid x valid) with a valid centre bin; a "
                                         "slot keeps only the centre row, which does not determine such a mask for embed.n_layers = %d > 1 "
                                         "(use model(...) / model.pack_batch(...), which pass the full [B, 1, L, L] mask)" % (b, model._kws[0]["n_layers"]))
                pm, cm = pm[:, 0, 0, L // 2, :], cm[:, :, 0, L // 2, :]
            put(self.pm[r], pm)
            put(self.cm[r], cm)
            put(self.im[r], d["interaction_masks"][b])
        put(self.freq, d["interaction_freq"])
        if "label" in d:
            put(self.label, d["label"].reshape(-1))
        return self


class EpochFeed:
    """The batches of an epoch taken from a device-resident GeneStore INSIDE the step graph (cf_gather_batch): the gene
    order of the epoch is uploaded once, a device-side cursor walks it, and every step's logits / labels / loss are
    appended to per-epoch logs (cf_record_step).  A training step is then four host calls -- graph replay, cf_rider_arm, the trunk's backward launch, reduction + AdamW --
    instead of the reference's DataLoader round trip plus per-tensor .cuda() copies (train.py:137-140, 171-177)."""

    def __init__(self, model, store, bsz, max_batches=None):
        dev = model._device
        self.model, self.store, self.B = model, store, bsz
        self.slot = Slot(model, bsz)
        self.slot.feed = self
        self.cap = max_batches or (len(store) // bsz + 1)
        self.order = torch.zeros(self.cap * bsz, dtype=torch.int32, device=dev)
        self._order_host = torch.zeros(self.cap * bsz, dtype=torch.int32).pin_memory()
        self.cursor = torch.zeros(4, dtype=torch.int32, device=dev)          # [next batch, batches uploaded, error flags, reserved]
        self._cursor_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.taken = 0                                                        # steps issued since begin_epoch (host-side guard)
        self.pregathered = False                                              # the next step's batch is in the slot already (Trainer._pregather)
        # the step logs live in pinned HOST memory (device-addressable): cf_record_step writes a KB per step straight into
        # it, the loop reads it once an event says the window is complete -- no device-to-host copy, which would queue
        # behind the replayed graphs (1.6 ms per copy on a busy GPU)
        self.logits_log = torch.zeros(self.cap, bsz, model.n_out).pin_memory()
        self.labels_log = torch.zeros(self.cap, bsz, dtype=self.slot.label.dtype).pin_memory()
        self.loss_log = torch.zeros(self.cap).pin_memory()
        self.struct = store.struct()
        self.n_batches = 0
        torch.cuda.current_stream(dev).synchronize()      # the fills above ran on the current stream (see Slot)

    def begin_epoch(self, batches, stream):
        """Upload the epoch's batches (equal-length lists of gene indices) and rewind the cursor, ordered on `stream`."""
        if len(batches) > self.cap:
            raise ValueError("epoch of %d batches exceeds the feed's capacity %d" % (len(batches), self.cap))
        if any(len(b) != self.B for b in batches):
            raise ValueError("every batch of an EpochFeed epoch must hold exactly %d genes" % self.B)
        # the order goes through ONE pinned staging buffer, filled with a numpy conversion: between two epochs the GPU is
        # idle until this returns (a Python list comprehension + a fresh pinned allocation cost ~5 ms per epoch, 3 % of an
        # 18,000-gene epoch).  The previous epoch's copy out of the buffer has completed: its windows have all been read.
        n = len(batches) * self.B
        if n:
            self._order_host[:n] = torch.from_numpy(np.asarray(batches, dtype=np.int32).reshape(-1))
        flags = self.check(stream)                         # of the epoch that just ended (waits for its last steps on `stream`)
        self._cursor_host[0], self._cursor_host[1], self._cursor_host[2] = 0, len(batches), 0
        with torch.cuda.stream(stream):
            if n:
                self.order[:n].copy_(self._order_host[:n], non_blocking=True)
            self.cursor.copy_(self._cursor_host, non_blocking=True)
        self.n_batches, self.taken = len(batches), 0
        self.pregathered = False
        if flags:
            raise RuntimeError("EpochFeed: the previous epoch set error flags %d on the device (1: a step past the epoch, "
                               "2: a gene index outside the store)" % flags)

    def check(self, stream=None):
        """Error flags the gather kernel has set (cursor[2]); read behind the work queued on `stream` (the trainer's: it does
        not synchronise with the current stream), which it waits for."""
        if stream is None:
            torch.cuda.synchronize(self.cursor.device)
            return int(self.cursor[2].item())
        with torch.cuda.stream(stream):
            return int(self.cursor[2].item())

    def rewind(self):
        """Back to the first batch of the uploaded epoch (the eager validation pass before a capture consumed one)."""
        self.cursor[0] = 0                                 # (on the current stream: the caller synchronises before capturing)
        self.taken = 0
        self.pregathered = False

    def window(self, lo, hi):
        """(logits [n, n_out], labels [n], losses [hi - lo]) of steps lo .. hi - 1 of the epoch, on the host (the caller has
        made sure those steps are complete)."""
        return (self.logits_log[lo:hi].reshape(-1, self.model.n_out).clone(), self.labels_log[lo:hi].reshape(-1).clone(),
                self.loss_log[lo:hi].clone())


class Trainer:
    """One optimisation step = forward, loss, backward, gradient reductions, [all-reduce], AdamW.

    Single GPU: everything up to the optimiser is one captured hipGraph (the launch sequence is static); AdamW is
    launched eagerly behind it because its bias corrections change every step (`opt_in_graph=True` moves it inside,
    reading the scalars from device memory -- measured slower, see __init__).
    Data parallel: two graphs.  The first ends with the Regulation + head gradient bucket (78 % of the bytes), whose
    RCCL all-reduce then runs on a side stream while the second graph (Pairwise + Embedding backward and their
    bucket) runs on the main one.
    ``timed_kernel``: bench.py's roofline kernel; its launches are bracketed by HIP events.  The library splits the
    capture around a Regulation kernel and launches it eagerly between the two graph pieces (event-record nodes
    inside a graph cost ~60 us per replay with this runtime; an eager launch between two graphs costs nothing)."""

    def __init__(self, model, lr=3e-5, gamma=0.87, world_size=1, process_group=None, use_graph=True,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, timed_kernel=None, opt_in_graph=False, overlap_opt=False,
                 overlap_allreduce=None, merge_opt=True, overlap_reduce=None, fuse_opt=None, keep_grads=False, fuse_one=None, rider_tiles=None,
                 keep_tiled=None, dp_halves=None, dp_early_opt=None, dp_side_reduce=None):
        self.model, self.lr, self.gamma = model, float(lr), gamma
        self.world, self.pg, self.use_graph = world_size, process_group, use_graph
        self.dp = world_size > 1 or process_group is not None
        # Data parallel: the early (Regulation + head) bucket's all-reduce normally runs on the side stream UNDER the Pairwise +
        # Embedding backward.  Those kernels are latency-bound and sensitive to co-running work (a weight-gradient launch
        # beside them tripled k_attc2<bwd> once), and RCCL's kernels occupy CUs too: with
        # overlap_allreduce=False (or CF_DP_OVERLAP=0 in the environment) both all-reduces are issued behind the whole
        # backward pass instead.  Same arithmetic, same bits; only the schedule differs.
        if overlap_allreduce is None:
            import os
            overlap_allreduce = os.environ.get("CF_DP_OVERLAP", "1") != "0"
        self.overlap_allreduce = bool(overlap_allreduce)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self._L = _lib.lib()
        self._last = None
        # hipGraph capture needs a non-default stream; all work of the trainer runs on it
        self.stream = torch.cuda.Stream(device=model._device)
        self.side = torch.cuda.Stream(device=model._device)       # all-reduce of the early gradient bucket
        torch.cuda.synchronize(model._device)      # parameter upload / buffer fills of the model ran on the default stream
        # four events, created once: a fresh torch.cuda.Event per step is a hipEventCreate on the critical path of the host loop
        self._ev_fork, self._ev_join, self._ev_late, self._ev_done = (torch.cuda.Event() for _ in range(4))
        # Data parallel, round 5: the Regulation + head bucket goes on the wire in two HALVES.  Its all-reduce (16.6 MB, ~0.17 ms per-link
        # bound on eight ranks) is the critical path of the step; as one piece it starts behind the whole Regulation backward and the
        # reduction of 1,008 tiles (~180 us after that launch began).  With the backward as two launches -- upper half of the layers, then
        # the lower half -- the upper half's + the head's tiles are reduced and sent while the lower half still runs (~90 us earlier),
        # the lower half's follow under the Pairwise + Embedding backward.  Same gradients, same bits (tests/test_engine_gpu.py);
        # dp_halves=False / CF_DP_HALVES=0: the two-bucket schedule of rounds 2-4.
        if dp_halves is None:
            import os
            dp_halves = os.environ.get("CF_DP_HALVES", "1") != "0"
        self.halves = bool(dp_halves) and self.dp and not opt_in_graph and int(self._L.cf_reg_halves(model._handle)) > 0
        if dp_early_opt is None:
            import os
            dp_early_opt = os.environ.get("CF_DP_EARLY_OPT", "1") != "0"
        self.dp_early_opt = bool(dp_early_opt) and self.dp
        # Data parallel, round 6: the weight-gradient reductions of the two Regulation halves run on the SIDE stream, in front of the all-reduce that
        # ships them -- beside the next backward launch of the main stream (192 workgroups on 256 CUs) instead of between two of them.  The main
        # stream's chain loses two `k_reduce` launches (2 x 30 us); the all-reduces start where they did (the reduction that precedes each is the
        # same launch, one stream over).  Same kernels on the same data in the same order per bucket: same bits (tests/test_engine_gpu.py).  Eager
        # launches only (what chromoformer_amd.train and bench.py run under data parallelism); dp_side_reduce=False / CF_DP_SIDE_REDUCE=0: round 5.
        if dp_side_reduce is None:
            import os
            dp_side_reduce = os.environ.get("CF_DP_SIDE_REDUCE", "1") != "0"
        self.dp_side_reduce = bool(dp_side_reduce) and self.halves and self.overlap_allreduce and not use_graph and not opt_in_graph
        self._ev_mid = torch.cuda.Event()
        self._buckets = {}
        for b in (_lib.BUCKET_REG, _lib.BUCKET_PE) + ((_lib.BUCKET_REG_HI, _lib.BUCKET_REG_LO) if self.halves else ()):
            off, n = C.c_longlong(), C.c_longlong()
            _lib.check(self._L.cf_grad_bucket(model._handle, b, C.byref(off), C.byref(n)), "cf_grad_bucket")
            self._buckets[b] = model._gflat[off.value: off.value + n.value]
        self.timed_kernel = timed_kernel
        if timed_kernel in ("k_wgrad", "k_colsum", "k_adamw", "k_trunk_fwd", "k_trunk_bwd"):
            self.use_graph = False        # launched once per bucket: timed on the eager path, where every launch gets its events
        # Measured and left off by default: AdamW inside the graph (scalars from device memory) costs +5 us per step (one
        # more launch per bucket plus cf_adamw_set), and running the early bucket's update as a parallel branch under the
        # Pairwise + Embedding backward costs +36 us -- its 116 MB stream evicts the L2-resident tables those
        # latency-bound kernels live on.  The default is one eager launch over the whole range behind the graph.
        self.opt_in_graph = opt_in_graph and timed_kernel != "k_adamw"
        self.overlap_opt = overlap_opt and self.opt_in_graph
        # Single GPU: the Embedding + Pairwise bucket's gradient reduction (~300 latency-bound tiles, 21 us on a mostly idle chip)
        # and the Regulation + head bucket's AdamW (a 116 MB stream, 19 us) are independent of each other: ONE launch runs them
        # side by side (cf_reduce_adamw_part), the small bucket's AdamW follows.  Three host calls per step instead of two.
        self.merge_opt = bool(merge_opt) and not self.dp and not self.opt_in_graph and timed_kernel not in ("k_wgrad", "k_colsum", "k_adamw")
        # Single GPU: the Regulation + head bucket's gradient reduction (55 us) depends on the Regulation backward only, the Pairwise +
        # Embedding backward (k_trunk_bwd, 135 us on 192 of the 256 CUs) does not depend on it: with overlap_reduce the reduction is
        # issued on the side stream BEHIND the trunk launch and runs beside it (CF_OVERLAP_REDUCE=1; see DESIGN.md for the numbers).
        if overlap_reduce is None:
            import os
            overlap_reduce = os.environ.get("CF_OVERLAP_REDUCE", "0") != "0"
        self.overlap_reduce = bool(overlap_reduce) and self.merge_opt
        # Single GPU: AdamW in the EPILOGUE of the two gradient reductions (cf_reduce_opt_part): the tile that finishes a gradient
        # element updates the parameter and its moments -- no optimiser launches, no second pass over gradients and state.  The step
        # is then [graph: gather, forward, head, Regulation backward] -> reduction + AdamW of the Regulation + head bucket -> Pairwise +
        # Embedding backward -> reduction + AdamW of their bucket.  The flat gradient buffer is only written with keep_grads=True.
        # Bit-identical parameters and moments (tests/test_engine_gpu.py); CF_FUSE_OPT=0 restores the merged schedule above.
        if fuse_opt is None:
            import os
            fuse_opt = os.environ.get("CF_FUSE_OPT", "1") != "0"
        self.fuse_opt = (bool(fuse_opt) and self.merge_opt and not self.overlap_reduce and model._kws[0].get("n_layers", 1) == 1)
        self.keep_grads = bool(keep_grads)
        # ... and both buckets' tiles in ONE launch behind the whole backward pass (the Embedding + Pairwise bucket's ~300 latency-bound
        # tiles fill the gaps of the Regulation bucket's 1,008): [graph: gather, forward, head, both backward launches] -> reduction +
        # AdamW.  7 launches, 2 host calls per step; 0.600 -> 0.590 ms.  fuse_one=False / CF_FUSE_ONE=0: one launch per bucket.
        if fuse_one is None:
            import os
            fuse_one = os.environ.get("CF_FUSE_ONE", "1") != "0"
        self.fuse_one = bool(fuse_one)
        # ... and part of the Regulation bucket's tiles as RIDERS of the trunk's backward launch (cf_rider_arm): the trunk's 192 workgroups
        # occupy one CU each for ~130 us of latency chains, the other 64 CUs reduce (and step) rider_tiles of that bucket's 1,008
        # weight-gradient tiles meanwhile, one tile per wave.  0.590 -> 0.564 ms; more than one tile per rider wave outlasts the trunk
        # (768: 0.650 ms).  rider_tiles=0 / CF_RIDER_TILES=0: everything in the reduction launch.
        if rider_tiles is None:
            import os
            rider_tiles = int(os.environ.get("CF_RIDER_TILES", "512"))
        self.rider_tiles = int(rider_tiles) if self.fuse_opt and self.fuse_one else 0
        if self.rider_tiles > 0 and self._L.cf_rider_arm(model._handle, 0.0, 0.9, 0.999, 1e-8, 0.0, 1, 0, 0) != 0:
            self.rider_tiles = 0          # (configurations without the fused trunk kernels)
        # riders fit on the CUs the trunk's n_res x B one-per-CU workgroups leave idle: sized from the device's CU count (256 on an
        # MI355X; a partition or another part has fewer / more), none when the trunk alone fills the device
        self._n_cu = int(self._L.cf_cu_count(model._handle)) or 256
        # ... and the fused optimiser keeps the tiled copies of the Embedding + Pairwise weights fresh (its epilogue writes every stepped
        # element in both layouts): no re-tiling launch in front of a step -- 6 -> 5 launches where no batch gather shares that launch.
        # keep_tiled=False / CF_KEEP_TILED=0: every forward pass re-tiles (model.keep_tiled has the contract).
        if keep_tiled is None:
            import os
            keep_tiled = os.environ.get("CF_KEEP_TILED", "1") != "0"
        # Data parallel (round 5): the separate AdamW launch over the Embedding + Pairwise bucket writes the tiled copies as well
        # (k_adamw_tiled), so the mode -- and the step without a re-tiling launch in front -- holds there too.
        tiled_dp = self.dp and not self.opt_in_graph
        self.keep_tiled = model.keep_tiled(True) if (keep_tiled and (self.fuse_opt or tiled_dp)) else (model.keep_tiled(False) and False)
        self._t_ms, self._t_n = 0.0, 0
        _lib.check(self._L.cf_timing_select(model._handle, timed_kernel.encode() if timed_kernel else None), "cf_timing_select")

    def stage(self, batch, slot=None):
        slot = slot or Slot(self.model, batch["interaction_freq"].shape[0])
        with torch.cuda.stream(self.stream):
            return slot.fill(self.model, batch)

    def _stream(self):
        return self.stream.cuda_stream

    # ---- launch sequences (eager or under capture)
    def _part(self, slot, st, parts):
        m, L = self.model, self._L
        feed = getattr(slot, "feed", None)
        if feed is not None and (parts & 4):
            # the step log (logits / labels / loss of the step into the feed's pinned-host logs) is written by the Pairwise + Embedding
            # backward launch itself (cf_record_step_bwd: no launch of its own; the loss is final once the Regulation backward has run)
            _lib.check(L.cf_record_step_bwd(m._handle, feed.cursor.data_ptr(), slot.logits.data_ptr(), slot.label.data_ptr(),
                                            slot.loss.data_ptr(), slot.B, feed.logits_log.data_ptr(), feed.labels_log.data_ptr(),
                                            feed.loss_log.data_ptr(), st), "cf_record_step_bwd")
        _lib.check(L.cf_backward_part(m._handle, C.byref(slot.struct), slot.label.data_ptr(), 1.0 / self.world,
                                      slot.loss.data_ptr(), parts, st), "cf_backward_part")

    def _reduce(self, slot, st, buckets):
        _lib.check(self._L.cf_backward_reduce_part(self.model._handle, slot.B, buckets, st), "cf_backward_reduce_part")

    def _seq_early(self, slot, st, reduce=True, gather=True):     # [batch gather,] forward, loss, head + Regulation backward, [step log,] Regulation + head gradient bucket
        m, L = self.model, self._L
        feed = getattr(slot, "feed", None) if gather else None      # (gather=False: the batch is in place already, see _pregather)
        if feed is not None:
            # (shares a launch with the prologue of the forward below: cf_gather_batch_fwd)
            _lib.check(L.cf_gather_batch_fwd(m._handle, C.byref(feed.struct), feed.order.data_ptr(), feed.cursor.data_ptr(),
                                             C.byref(slot.struct), slot.label.data_ptr(), st), "cf_gather_batch_fwd")
        # the training forward: head forward, loss and head backward at the tail of the Regulation launch where the library can
        # (cf_head_rides), else left to cf_backward_part, where the three are one launch; part 1 below is a no-op in the first case
        _lib.check(L.cf_forward_train(m._handle, C.byref(slot.struct), slot.logits.data_ptr(), slot.label.data_ptr(), 1.0 / self.world,
                                      slot.loss.data_ptr(), st), "cf_forward_train")
        if self.halves:      # data parallel: head + upper half of the Regulation layers, and their gradient bucket
            self._part(slot, st, 1 | _lib.PART_REG_HI)
            if reduce:
                self._reduce(slot, st, _lib.BUCKET_REG_HI)
            return
        self._part(slot, st, 3)
        if reduce:
            self._reduce(slot, st, _lib.BUCKET_REG)

    def _seq_mid(self, slot, st, reduce=True):       # data parallel in halves: lower half of the Regulation layers and their gradient bucket
        self._part(slot, st, _lib.PART_REG_LO)
        if reduce:
            self._reduce(slot, st, _lib.BUCKET_REG_LO)

    def _seq_main(self, slot, st):      # single GPU, merged optimiser: everything up to the Pairwise + Embedding backward (its bucket is reduced beside AdamW)
        if self.overlap_reduce:
            side = self.side.cuda_stream
            self._seq_early(slot, st, reduce=False)
            self._wait(side, st)                             # fork behind the Regulation backward
            self._part(slot, st, 4)                          # the trunk's 192 big workgroups are dispatched first ...
            self._reduce(slot, side, _lib.BUCKET_REG)        # ... the reduction tiles fill what is left
            self._wait(st, side)                             # join
            return
        self._seq_early(slot, st)
        self._part(slot, st, 4)

    def _seq_late(self, slot, st):      # Pairwise + Embedding backward and their gradient bucket
        self._part(slot, st, 4)
        self._reduce(slot, st, _lib.BUCKET_PE)

    def _opt(self, bucket, st):
        _lib.check(self._L.cf_adamw_step_dev(self.model._handle, bucket, st), "cf_adamw_step_dev")

    def _wait(self, waiter, signaller):
        _lib.check(self._L.cf_stream_wait(self.model._handle, waiter, signaller), "cf_stream_wait")

    def _seq_all(self, slot, st, opt=True):
        """The whole single-GPU step.  With `opt`, AdamW of the early bucket (78 % of the 150 MB of optimiser traffic) runs
        on the side stream under the Pairwise + Embedding backward -- an HBM stream next to latency-bound kernels."""
        side = self.side.cuda_stream
        self._seq_early(slot, st)
        if self.halves:
            self._seq_mid(slot, st)
        if opt:
            if self.overlap_opt:
                self._wait(side, st)
                self._opt(_lib.BUCKET_REG, side)
            else:
                self._opt(_lib.BUCKET_REG, st)
        self._seq_late(slot, st)
        if opt:
            self._opt(_lib.BUCKET_PE, st)
            if self.overlap_opt:
                self._wait(st, side)

    def _capture(self, fn, slot, st):
        m, L = self.model, self._L
        gid = C.c_int()
        _lib.check(L.cf_capture_begin(m._handle, st), "cf_capture_begin")
        try:
            fn(slot, st)
        finally:
            _lib.check(L.cf_capture_end(m._handle, st, C.byref(gid)), "cf_capture_end")
        return gid.value

    def _launch(self, gid, st):
        _lib.check(self._L.cf_graph_launch(self.model._handle, gid, st), "cf_graph_launch")

    def step(self, slot):
        """One optimisation step on a staged batch (train.py:182-196)."""
        with torch.cuda.stream(self.stream):
            return self._step(slot)

    def _step(self, slot):
        m, L = self.model, self._L
        st = self._stream()
        # keep_tiled is a mode of the MODEL, the captured graphs are this Trainer's: another Trainer on the same model may have switched the
        # mode since (its constructor does).  A graph captured with the mode on holds no Embedding + Pairwise re-tiling and would replay on
        # stale tiled weights once the optimiser stops writing them -- so the mode this Trainer was built (and captured) with is put back
        # in front of every step; switching it drops the version stamp, and _sync_tiled below rebuilds the copies once.
        if bool(getattr(m, "_keep_tiled", False)) != bool(self.keep_tiled):
            m.keep_tiled(self.keep_tiled)
        m._sync_tiled(st)          # (keep_tiled: parameters written through torch since the last step are re-tiled here, also in front of a graph replay)
        feed = getattr(slot, "feed", None)
        if feed is not None:
            if feed.taken >= feed.n_batches:           # the device-side bound would skip the gather; refuse on the host, loudly
                raise RuntimeError("Trainer.step: the feed's epoch of %d batches is exhausted (call begin_epoch)" % feed.n_batches)
        # Fused single-GPU step fed from an EpochFeed: the batch of step k + 1 is gathered inside step k's last launch (cf_gather_batch_next,
        # behind the reduction tiles), the first batch of an epoch by a launch of its own (cf_gather_batch_only): no launch in front of a step.
        pre = feed is not None and self.fuse_opt and self.fuse_one
        gargs = None
        if pre:
            gargs = (m._handle, C.byref(feed.struct), feed.order.data_ptr(), feed.cursor.data_ptr(), C.byref(slot.struct), slot.label.data_ptr(), st)
        if self.use_graph and slot.graph is None:
            self._seq_all(slot, st, opt=False)         # eager once (validates the arguments before anything is captured)
            if getattr(slot, "feed", None) is not None:
                slot.feed.rewind()                     # the validation pass consumed a batch: rewind (no parameter was updated)
            torch.cuda.synchronize()
            first = self._seq_early if self.dp else (self._seq_main if self.merge_opt else (lambda s_, t_: self._seq_all(s_, t_, opt=self.opt_in_graph)))
            if self.fuse_opt and self.fuse_one and self.rider_tiles == 0:
                first = lambda s_, t_: (self._seq_early(s_, t_, reduce=False, gather=not pre), self._part(s_, t_, 4))
            elif self.fuse_opt:
                first = lambda s_, t_: self._seq_early(s_, t_, reduce=False, gather=not pre)
            if pre:      # (the captured trunk launch advances the cursor because a gathered batch is waiting at capture time)
                _lib.check(L.cf_gather_batch_only(*gargs), "cf_gather_batch_only")
                feed.pregathered = True
            slot.graph = {"first": self._capture(first, slot, st), "mid": self._capture(self._seq_mid, slot, st) if self.halves else None,
                          "late": self._capture(self._seq_late, slot, st) if self.dp else None}
        if feed is not None:
            feed.taken += 1
        oig = self.opt_in_graph
        if oig:      # this step's AdamW scalars go to device memory before anything is replayed
            m._step += 1
            _lib.check(L.cf_adamw_set(m._handle, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step, st), "cf_adamw_set")
        if self.fuse_opt:
            if pre and not feed.pregathered:           # first step of an epoch
                _lib.check(L.cf_gather_batch_only(*gargs), "cf_gather_batch_only")
            if self.use_graph:
                self._launch(slot.graph["first"], st)
            else:
                self._seq_early(slot, st, reduce=False, gather=not pre)
            if pre:                                    # the next step's batch rides in this step's reduction launch
                feed.pregathered = feed.taken < feed.n_batches
                if feed.pregathered:
                    _lib.check(L.cf_gather_batch_next(*gargs), "cf_gather_batch_next")
            m._step += 1
            hp = (self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step)
            kg = 1 if self.keep_grads else 0
            if self.fuse_one and self.rider_tiles > 0:
                # riders: part of the Regulation bucket's tiles (with their AdamW) inside the trunk's backward launch, on the CUs it leaves idle
                # (one tile per rider wave, eight waves per idle CU: 256 CUs minus the trunk's n_res x B workgroups)
                n_rd = min(self.rider_tiles, 8 * max(0, self._n_cu - len(m.binsizes) * slot.B))
                _lib.check(L.cf_rider_arm(m._handle, *hp, kg, n_rd), "cf_rider_arm")      # (n_rd = 0 disarms: no idle CU)
                self._part(slot, st, 4)
                _lib.check(L.cf_reduce_opt_part(m._handle, slot.B, _lib.BUCKET_REG | _lib.BUCKET_PE, *hp, kg, st), "cf_reduce_opt_part")
            elif self.fuse_one:       # both buckets' tiles in ONE launch behind the whole backward pass
                if not self.use_graph:
                    self._part(slot, st, 4)
                _lib.check(L.cf_reduce_opt_part(m._handle, slot.B, _lib.BUCKET_REG | _lib.BUCKET_PE, *hp, kg, st), "cf_reduce_opt_part")
            else:
                _lib.check(L.cf_reduce_opt_part(m._handle, slot.B, _lib.BUCKET_REG, *hp, kg, st), "cf_reduce_opt_part")
                self._part(slot, st, 4)
                _lib.check(L.cf_reduce_opt_part(m._handle, slot.B, _lib.BUCKET_PE, *hp, kg, st), "cf_reduce_opt_part")
        elif not self.dp and self.merge_opt:
            if self.use_graph:
                self._launch(slot.graph["first"], st)
            else:
                self._seq_main(slot, st)
            m._step += 1
            hp = (self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step)
            _lib.check(L.cf_reduce_adamw_part(m._handle, slot.B, _lib.BUCKET_PE, *hp, _lib.BUCKET_REG, st), "cf_reduce_adamw_part")
            _lib.check(L.cf_adamw_step_part(m._handle, *hp, _lib.BUCKET_PE, st), "cf_adamw_step_part")
        elif not self.dp:
            if self.use_graph:
                self._launch(slot.graph["first"], st)
            else:
                self._seq_all(slot, st, opt=oig)
        else:
            side_red = self.dp_side_reduce and not self.use_graph
            if self.use_graph:
                self._launch(slot.graph["first"], st)
            else:
                self._seq_early(slot, st, reduce=not side_red)
            # (dp_early_opt: AdamW over the Regulation + head range on the SIDE stream, straight behind that range's last all-reduce -- under the
            #  Pairwise + Embedding backward -- instead of on the main stream behind everything: its streaming loads and stores (CF_ADAM_NT) no longer
            #  sweep the L2s the trunk works in, which is what made this lose in round 3)
            early_opt = self.dp_early_opt and not oig and self.overlap_allreduce
            hp_next = (self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step + 1)

            def early_allreduce(bucket=_lib.BUCKET_REG, ev=None, last=True):      # an early bucket is complete: all-reduce it on the side stream
                ev = ev or self._ev_fork
                ev.record(self.stream)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ev)
                    if side_red:      # the bucket's weight-gradient tiles, here instead of on the main stream (dp_side_reduce)
                        self._reduce(slot, self.side.cuda_stream, bucket)
                    torch.distributed.all_reduce(self._buckets[bucket], group=self.pg)     # SUM; dloss carries 1/world
                    if last:
                        if self.overlap_opt:
                            self._opt(_lib.BUCKET_REG, self.side.cuda_stream)
                        elif early_opt:
                            _lib.check(L.cf_adamw_step_part(m._handle, *hp_next, _lib.BUCKET_REG, self.side.cuda_stream), "cf_adamw_step_part")
                        self._ev_join.record(self.side)

            if self.halves:             # the upper half + head are on the wire while the lower half's backward runs
                if self.overlap_allreduce:
                    early_allreduce(_lib.BUCKET_REG_HI, self._ev_fork, last=False)
                if self.use_graph:
                    self._launch(slot.graph["mid"], st)
                else:
                    self._seq_mid(slot, st, reduce=not side_red)
                if self.overlap_allreduce:
                    early_allreduce(_lib.BUCKET_REG_LO, self._ev_mid, last=True)
            elif self.overlap_allreduce:
                early_allreduce()       # ... under the rest of the backward pass
            if self.use_graph:
                self._launch(slot.graph["late"], st)
            else:
                self._seq_late(slot, st)
            if not self.overlap_allreduce:      # ... behind it (serialised schedule)
                if self.halves:
                    early_allreduce(_lib.BUCKET_REG_HI, self._ev_fork, last=False)
                    early_allreduce(_lib.BUCKET_REG_LO, self._ev_mid, last=True)
                else:
                    early_allreduce()
            if oig:
                torch.distributed.all_reduce(self._buckets[_lib.BUCKET_PE], group=self.pg)
                self._opt(_lib.BUCKET_PE, st)
                self.stream.wait_event(self._ev_join)
                if not self.overlap_opt:
                    self._opt(_lib.BUCKET_REG, st)
            elif early_opt:
                # the Regulation + head range is stepped on the side stream (early_allreduce above): nothing is left for the main stream but the late
                # bucket, so its all-reduce is issued from HERE -- main -> RCCL's stream -> main instead of main -> side -> RCCL -> side -> main: two
                # cross-stream event hops less on the tail of every step (the transfers themselves queue behind the early ones on RCCL's stream)
                torch.distributed.all_reduce(self._buckets[_lib.BUCKET_PE], group=self.pg)
                self.stream.wait_event(self._ev_join)            # early buckets reduced and stepped
                m._step += 1
                hp = (self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step)
                _lib.check(L.cf_adamw_step_part(m._handle, *hp, _lib.BUCKET_PE, st), "cf_adamw_step_part")
            else:
                # the late bucket's all-reduce goes to the side stream as well (behind the early one) and the main stream
                # steps the early bucket meanwhile: only the late bucket's 6 us of AdamW wait for the second all-reduce
                ev_late, ev_done = self._ev_late, self._ev_done
                ev_late.record(self.stream)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ev_late)
                    torch.distributed.all_reduce(self._buckets[_lib.BUCKET_PE], group=self.pg)
                    ev_done.record(self.side)
                self.stream.wait_event(self._ev_join)            # early bucket reduced (dp_early_opt: and stepped)
                m._step += 1
                hp = (self.lr, self.betas[0], self.betas[1], self.eps, self.wd, m._step)
                if not early_opt:
                    _lib.check(L.cf_adamw_step_part(m._handle, *hp, _lib.BUCKET_REG, st), "cf_adamw_step_part")
                self.stream.wait_event(ev_done)                  # late bucket reduced
                _lib.check(L.cf_adamw_step_part(m._handle, *hp, _lib.BUCKET_PE, st), "cf_adamw_step_part")
        if not oig and not self.dp and not self.merge_opt:
            m.adamw_step(self.lr, self.betas, self.eps, self.wd)
        self._last = slot
        m._mark_grads(self.fuse_opt and not self.keep_grads)
        return slot.logits, slot.loss

    def evaluate(self, slot):
        m = self.model
        with torch.cuda.stream(self.stream):
            m._sync_tiled(self._stream())
            _lib.check(self._L.cf_forward(m._handle, C.byref(slot.struct), slot.logits.data_ptr(), 0, self._stream()), "cf_forward")
        return slot.logits

    def evaluate_store(self, store, bsz):
        """Forward-only pass over ALL genes of a device-resident store in store order (validation, train.py:236-275; prediction,
        demo/run_demo.py:98-114) -> logits [len(store), n_out] on the device.  Per batch: one cf_gather_batch out of the
        resident arrays (the same launch the training graph starts with) and cf_forward writing its logits straight into the
        result -- instead of ~17 ATen index / copy launches per batch; the tail batch is kept (train.py:140)."""
        m, L = self.model, self._L
        n = len(store)
        out = torch.empty(n, m.n_out, device=m._device)
        if n == 0:
            return out
        struct = store.struct()                                   # raises for a host-side store
        cache = self.__dict__.setdefault("_eval_slots", {})
        cursors = []
        # labels travel only when the store's label kind is the model's (a prediction run may open a store packed for the other
        # task: the gather would copy 4-byte labels out of an 8-byte column)
        with_labels = bool(getattr(store, "regression", m.n_out == 1)) == (m.n_out == 1)
        with torch.cuda.stream(self.stream):
            m._sync_tiled(self._stream())
            order = torch.arange(n, dtype=torch.int32, device=m._device)
            n_full = n // bsz
            for B, lo, nb in ((bsz, 0, n_full), (n - n_full * bsz, n_full * bsz, 1)):
                if B == 0 or nb == 0:
                    continue
                slot = cache.get(B)
                if slot is None:
                    slot = cache[B] = Slot(m, B)
                cursor = torch.tensor([0, nb, 0, 0], dtype=torch.int32).to(m._device)
                cursors.append(cursor)
                st = self._stream()
                for k in range(nb):
                    _lib.check(L.cf_gather_batch(m._handle, C.byref(struct), order[lo:].data_ptr(), cursor.data_ptr(), C.byref(slot.struct),
                                                 slot.label.data_ptr() if with_labels else None, st), "cf_gather_batch")
                    _lib.check(L.cf_forward(m._handle, C.byref(slot.struct), out[lo + k * B:].data_ptr(), 0, st), "cf_forward")
            self.stream.synchronize()                             # `order` / `cursor` may be released; the result is complete
            # the gather is bounded on the device and SKIPS what it cannot serve (batch index past the upload, gene index outside the
            # store), leaving the previous batch's rows in the slot: its error flags decide whether the logits mean anything
            flags = [int(c[2].item()) for c in cursors]
            if any(flags):
                raise RuntimeError("evaluate_store: the device-side gather reported errors (flags %s): store / order mismatch, "
                                   "the logits of the skipped genes would be those of an earlier batch" % flags)
        return out

    def scheduler_step(self):
        """StepLR(step_size=1, gamma) (train.py:158, 344)."""
        self.lr *= self.gamma

    def last_loss(self):
        return self._last.loss.item() if self._last is not None else float("nan")

    # ------------------------------------------------------------------ measurement
    def timing_reset(self):
        self.timing_read()
        self._t_ms, self._t_n = 0.0, 0

    def timing_read(self):
        """-> (summed duration in ms, launches) of `timed_kernel` since the last reset (waits for the recorded events)."""
        ms, n = C.c_float(), C.c_int()
        _lib.check(self._L.cf_timing_read(self.model._handle, C.byref(ms), C.byref(n)), "cf_timing_read")
        self._t_ms += ms.value
        self._t_n += n.value
        return self._t_ms, self._t_n

    #: rocprofv3 names of the kernels bench.py can time (the `cf_timing_select` keys are the stage names)
    ROCPROF_NAMES = {"k_reg_bwd": "k_reg8_bwd", "k_reg_fwd": "k_reg8_fwd", "k_wgrad": "k_reduce_opt / k_reduce (weight-gradient tiles)"}

    def roofline(self, kernel, total_ms, launches, B, steps=None):
        """Roofline entry of bench.py for the kernel timed with HIP events.  `achieved` is normalised PER STEP when a
        step launches the kernel more than once (data parallel in halves: the Regulation backward is two launches over half the layers
        each, the weight-gradient reduction one per bucket) -- the flop count of `cf_kernel_flops` is the whole stack's, so it is divided by
        the time ALL of a step's launches of that kernel took, not by one launch's."""
        if not launches:
            return None
        per_step = max(1, int(round(launches / steps))) if steps else (2 if kernel == "k_wgrad" else 1)
        avg_s = total_ms / launches * 1e-3 * per_step            # the kernel's time per step
        name = self.ROCPROF_NAMES.get(kernel, kernel)
        if kernel in ("k_wgrad", "k_reg_fwd", "k_reg_bwd", "k_trunk_fwd", "k_trunk_bwd"):
            flops = self._L.cf_kernel_flops(self.model._handle, kernel.encode(), B)
            ach = flops / avg_s / 1e12
            return {"kernel": name, "timing_key": kernel, "bound": "mfma", "achieved": round(ach, 3), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(ach / 157.3, 4), "traffic": _pmc_traffic(kernel), "traffic_source": _TRAFFIC_SOURCE,
                    "avg_launch_us": round(avg_s * 1e6 / per_step, 2), "launches_per_step": per_step, "us_per_step": round(avg_s * 1e6, 2),
                    "algorithmic_gflop_per_launch": round(flops / 1e9 / per_step, 4), "algorithmic_gflop_per_step": round(flops / 1e9, 4),
                    "flop_accounting": "executed flops only: ~0.28 GFLOP/gene for a training step (centre query row, K/V projections absorbed into "
                                       "the query, last Regulation layer reduced to token 0); SURVEY 8-d's F_alg,train = 1.97 GFLOP/gene does not apply"}
        if kernel == "k_adamw":
            nbytes = 7.0 * 4.0 * self.model._layout.n_active
            ach = nbytes / avg_s / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(ach / 8000.0, 4), "traffic": _pmc_traffic(kernel), "traffic_source": _TRAFFIC_SOURCE,
                    "avg_launch_us": round(avg_s * 1e6, 2),
                    "algorithmic_mb_per_launch": round(nbytes / 1e6, 3)}
        return {"kernel": name, "avg_launch_us": round(avg_s * 1e6 / per_step, 2), "launches_per_step": per_step}
