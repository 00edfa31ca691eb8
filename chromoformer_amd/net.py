"""ChromoformerClassifier / ChromoformerRegressor on the MI355X HIP path.

Host-side mirror of the reference model interface (/root/reference/chromoformer/net.py:
273-428): same constructor arguments, same six-argument ``forward``, same
``state_dict()`` keys / shapes / order, same seeded initial weights.  All arithmetic is
done by libchromoformer_hip.so (csrc/); torch supplies device memory, streams and the
autograd hook only.  There is no PyTorch fallback: without the library or without a GPU
the model raises.

Two ways to train:
  * drop-in: ``out = model(...); loss = criterion(out, y); loss.backward()`` then a torch
    optimiser over ``model.parameters()`` (what the reference's train.py does);
  * fused:   ``model.train_step(batch, labels, lr)`` -- forward, loss, backward, optional
    gradient all-reduce and AdamW entirely inside the library (what bench.py and
    chromoformer_amd.train use).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib

_DEFAULT_EMBED = {"n_layers": 1, "n_heads": 2, "d_model": 128, "d_ff": 128}
_DEFAULT_PAIR = {"n_layers": 2, "n_heads": 2, "d_model": 128, "d_ff": 256}
_DEFAULT_REG = {"n_layers": 6, "n_heads": 8, "d_model": 256, "d_ff": 256}


def _positional_table(n_pos, dim):
    """Sinusoid table added to the token embeddings (net.py:23-29), float32 on the host."""
    pe = torch.zeros(n_pos, dim)
    pos = torch.arange(0, n_pos, 1).unsqueeze(1)
    k = torch.exp(-np.log(10000) * torch.arange(0, dim, 2) / dim)
    pe[:, 0::2] = torch.sin(pos * k)
    pe[:, 1::2] = torch.cos(pos * k)
    return pe.contiguous()


def _init_tensor(shape, fan_in, kind):
    """Default nn.Linear / nn.LayerNorm initialisers (same RNG draws as torch's)."""
    if kind == "ones":
        return torch.ones(shape)
    if kind == "zeros":
        return torch.zeros(shape)
    if kind == "weight":
        bound = math.sqrt(3.0) * math.sqrt(2.0 / (1 + math.sqrt(5) ** 2)) / math.sqrt(fan_in)  # kaiming_uniform_(a=sqrt(5))
    else:
        bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape).uniform_(-bound, bound)


def _init_kind(name, shape):
    leaf = name.rsplit(".", 2)[-2:]
    if leaf[-1] == "gamma_f":
        return "ones", 0
    if leaf[0] == "ln":
        return ("ones", 0) if leaf[1] == "weight" else ("zeros", 0)
    if leaf[1] == "weight":
        return "weight", shape[1]
    return "bias", None  # fan_in resolved from the sibling weight


class _BackwardHook(torch.autograd.Function):
    """Lets ``loss.backward()`` of the caller drive cf_backward_from()."""

    @staticmethod
    def forward(ctx, anchor, model, batch_struct, keep):
        ctx.model, ctx.batch_struct, ctx.keep = model, batch_struct, keep
        return model._run_forward(batch_struct, save=True)

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        dl = dlogits.contiguous().float()
        st = torch.cuda.current_stream(m._device).cuda_stream
        _lib.check(_lib.lib().cf_backward_from(m._handle, C.byref(ctx.batch_struct), dl.data_ptr(), st), "cf_backward_from")
        m._publish_grads()
        return torch.zeros_like(m._anchor), None, None, None


class ChromoformerBase(nn.Module):
    n_out = 2

    def __init__(self, n_feats=7, d_emb=128, d_head=128, embed_kws=None, pairwise_interaction_kws=None,
                 regulation_kws=None, binsizes=(2000, 500, 100), seed=42, i_max=8, w_max=40000, max_batch=64):
        super().__init__()
        torch.manual_seed(seed)
        embed = dict(_DEFAULT_EMBED if embed_kws is None else embed_kws)
        pair = dict(_DEFAULT_PAIR if pairwise_interaction_kws is None else pairwise_interaction_kws)
        reg = dict(_DEFAULT_REG if regulation_kws is None else regulation_kws)
        embed["d_model"] = d_emb                      # net.py:305
        self.binsizes = [int(b) for b in binsizes]    # accepts the CLI's strings (train.py:33)
        self.n_feats, self.d_emb, self.d_head = n_feats, d_emb, d_head
        self.i_max, self.w_max = i_max, w_max
        self.n_bins = [w_max // b for b in self.binsizes]
        self._kws = (embed, pair, reg)
        self._max_batch = max_batch
        self._handle = None
        self._device = None
        self._cfg = _lib.make_config(n_feats, d_emb, d_head, self.n_out, self.binsizes, self.n_bins, i_max, embed, pair, reg,
                                     max_batch)
        self._layout, self._table = _lib.param_layout(self._cfg)
        self._step = 0
        # seeded initial values, drawn in the reference's construction order (= state_dict order);
        # the regressor draws the 2-way head first and then a fresh 1-way head (net.py:411-428)
        draws = OrderedDict()
        entries = list(self._table)
        if self.n_out == 1:
            cfg2 = _lib.make_config(n_feats, d_emb, d_head, 2, self.binsizes, self.n_bins, i_max, embed, pair, reg, max_batch)
            entries = entries[:-4] + _lib.param_layout(cfg2)[1][-4:] + entries[-4:]
        fan = None
        for i, e in enumerate(entries):
            kind, fan_in = _init_kind(e["name"], e["shape"])
            if kind == "weight":
                fan = fan_in
            t = _init_tensor(e["shape"], fan if kind == "bias" else fan_in, kind)
            draws[e["name"]] = t          # later (regressor head) draws overwrite the earlier ones
        for e in self._table:
            self._register(e["name"], nn.Parameter(draws[e["name"]]))

    # ----------------------------------------------------------------- module plumbing
    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)

    def _named(self):
        return OrderedDict(self.named_parameters())

    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: chromoformer_amd has no CPU fallback (use the oracle/ only for checking)")
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else
                           (device if isinstance(device, int) else torch.device(device).index or 0))
        return self._materialize(dev)

    def to(self, *args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)) and torch.device(a).type == "cuda":
                dev = torch.device(a)
        if dev is None:
            raise RuntimeError("chromoformer_amd runs on an MI355X only; .to() accepts a cuda device")
        return self.cuda(dev.index)

    def _materialize(self, dev):
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: chromoformer_amd has no CPU fallback (use the oracle/ only for checking)")
        named = self._named()
        lay = self._layout
        with torch.cuda.device(dev):
            flat = torch.zeros(lay.n_total, device=dev)
            for e in self._table:
                flat[e["offset"]:e["offset"] + e["numel"]].copy_(named[e["name"]].detach().reshape(-1))
            self._flat = flat
            self._gflat = torch.zeros(lay.n_total, device=dev)
            self._mflat = torch.zeros(lay.n_total, device=dev)
            self._vflat = torch.zeros(lay.n_total, device=dev)
            self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            self._loss_buf = torch.zeros(1, device=dev)
            for e in self._table:
                named[e["name"]].data = flat[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])
            if self._handle is not None:
                _lib.lib().cf_destroy(self._handle)
            pes = [_positional_table(n, self.d_emb).numpy() for n in self.n_bins]
            ptrs = (C.c_void_p * len(pes))(*[p.ctypes.data for p in pes])
            h = C.c_void_p()
            _lib.check(_lib.lib().cf_create(C.byref(self._cfg), ptrs, C.byref(h)), "cf_create")
            self._handle = h
            _lib.check(_lib.lib().cf_bind(h, flat.data_ptr(), self._gflat.data_ptr(), self._mflat.data_ptr(),
                                          self._vflat.data_ptr()), "cf_bind")
        self._device = dev
        return self

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().cf_destroy(self._handle)
        except Exception:
            pass

    def _publish_grads(self):
        named = self._named()
        for e in self._table:
            if e["trainable"]:
                named[e["name"]].grad = self._gflat[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)   # copy_ into the flat-buffer views
        return out

    # ----------------------------------------------------------------- batches
    def _rows(self, mask, L, lead):
        """Centre-query-row pointer + stride of a pad mask: the reference's [.., 1, L, L] bool
        tensor (zero copy) or a compact [.., L] row array."""
        m = mask
        if m.dtype != torch.bool and m.dtype != torch.uint8:
            raise TypeError("pad masks must be bool / uint8")
        if not m.is_cuda:
            m = m.to(self._device)
        m = m.contiguous()
        if m.dim() >= 2 and m.shape[-1] == L and m.shape[-2] == L and m.numel() == lead * L * L:
            return m, m.data_ptr() + (L // 2) * L, L * L
        if m.numel() == lead * L:
            return m, m.data_ptr(), L
        raise ValueError("unexpected pad-mask shape %s" % (tuple(mask.shape),))

    def _pack(self, promoter_feats, promoter_pad_masks, pcre_feats, pcre_pad_masks, interaction_masks, interaction_freq):
        if self._handle is None:
            raise RuntimeError("call .cuda() first: the Chromoformer HIP path needs device buffers")
        bs = _lib.cf_batch()
        keep = []
        S, T = self.i_max, self.i_max + 1
        B = None

        def dev_f32(t):
            t = t.to(self._device, torch.float32).contiguous()
            keep.append(t)
            return t

        for r, b in enumerate(self.binsizes):
            L = self.n_bins[r]
            pf, cf = dev_f32(promoter_feats[b]), dev_f32(pcre_feats[b])
            B = pf.shape[0] if B is None else B
            if pf.numel() != B * L * self.n_feats or cf.numel() != B * S * L * self.n_feats:
                raise ValueError("feature shapes do not match the configuration at binsize %d" % b)
            bs.promoter_feats[r], bs.pcre_feats[r] = pf.data_ptr(), cf.data_ptr()
            m, ptr, stride = self._rows(promoter_pad_masks[b], L, B)
            keep.append(m)
            bs.promoter_mask_row[r], bs.promoter_mask_stride[r] = ptr, stride
            m, ptr, stride = self._rows(pcre_pad_masks[b], L, B * S)
            keep.append(m)
            bs.pcre_mask_row[r], bs.pcre_mask_stride[r] = ptr, stride
            im = interaction_masks[b].to(self._device).contiguous()
            if im.numel() != B * T * T:
                raise ValueError("interaction mask shape mismatch at binsize %d" % b)
            keep.append(im)
            bs.interaction_mask[r] = im.data_ptr()
        fr = dev_f32(interaction_freq)
        if fr.numel() != B * T * T:
            raise ValueError("interaction_freq shape mismatch")
        bs.interaction_freq = fr.data_ptr()
        bs.B = B
        if B > self._max_batch:
            raise ValueError("batch of %d genes exceeds max_batch=%d given at construction" % (B, self._max_batch))
        return bs, keep

    def pack_batch(self, d):
        """Pack a reference-layout batch dict (data.py:205-212) once, for repeated steps."""
        return self._pack(d["promoter_feats"], d["promoter_pad_masks"], d["pcre_feats"], d["pcre_pad_masks"],
                          d["interaction_masks"], d["interaction_freq"])

    # ----------------------------------------------------------------- forward / backward
    # ----------------------------------------------------------------- tiled weight copies kept fresh by the fused optimiser (cf_keep_tiled)
    def keep_tiled(self, on=True):
        """The fused optimiser writes the tiled copies of the Embedding + Pairwise weights it steps, forward passes stop re-tiling them
        (Trainer: no launch in front of a training step).  Whatever else writes parameters THROUGH TORCH -- load_state_dict, an optimiser,
        an in-place op on a parameter -- is noticed by the version counters (_sync_tiled, in front of every forward pass); an edit through
        `.data` or a raw pointer is not: call params_changed() after it.  Returns whether the library offers the mode for this model."""
        ok = _lib.lib().cf_keep_tiled(self._handle, 1 if on else 0) == 0
        self._keep_tiled = bool(on) and ok
        self._pver = None
        return self._keep_tiled

    def params_changed(self):
        if self._handle is not None:
            _lib.check(_lib.lib().cf_params_changed(self._handle), "cf_params_changed")
        self._pver = None

    def _sync_tiled(self, st):
        """In front of every forward pass of a model in keep_tiled mode: have parameters been written through torch since the library last
        saw them?  (sum of the version counters of the ~370 parameters: ~20 us of host time.)  If so the tiled copies are rebuilt at once, on
        the stream the pass will run on -- also in front of a hipGraph that was captured without the re-tiling."""
        if not getattr(self, "_keep_tiled", False):
            return
        if getattr(self, "_plist", None) is None:
            self._plist = [p for _, p in self.named_parameters()]
        ver = sum(p._version for p in self._plist)
        if ver != self._pver:
            _lib.check(_lib.lib().cf_params_changed(self._handle), "cf_params_changed")
            _lib.check(_lib.lib().cf_retile_early(self._handle, st), "cf_retile_early")
            self._pver = ver

    def _run_forward(self, bs, save):
        """save: False / 0 = inference, True / 1 = keep activations, 2 = as 1 with the head left to the cf_backward that follows."""
        out = torch.empty(bs.B, self.n_out, device=self._device)
        st = torch.cuda.current_stream(self._device).cuda_stream
        self._sync_tiled(st)
        _lib.check(_lib.lib().cf_forward(self._handle, C.byref(bs), out.data_ptr(), int(save), st), "cf_forward")
        return out

    def forward(self, promoter_feats, promoter_pad_masks, pcre_feats, pcre_pad_masks, interaction_masks, interaction_freq):
        bs, keep = self._pack(promoter_feats, promoter_pad_masks, pcre_feats, pcre_pad_masks, interaction_masks, interaction_freq)
        if torch.is_grad_enabled():
            return _BackwardHook.apply(self._anchor, self, bs, keep)
        return self._run_forward(bs, save=False)

    def embed_full(self, promoter_feats, promoter_pad_masks):
        """EmbeddingTransformer's first return value (net.py:57-59): {binsize: [B, 1, L, 128]}, the embedding of every
        promoter bin (all rows of every Embedding layer through the dense transformer layer; forward only)."""
        if self._handle is None:
            raise RuntimeError("call .cuda() first: the Chromoformer HIP path needs device buffers")
        bs = _lib.cf_batch()
        keep, B = [], None
        for r, b in enumerate(self.binsizes):
            L = self.n_bins[r]
            pf = promoter_feats[b].to(self._device, torch.float32).contiguous()
            B = pf.shape[0] if B is None else B
            if pf.numel() != B * L * self.n_feats:
                raise ValueError("promoter feature shape does not match the configuration at binsize %d" % b)
            m, ptr, stride = self._rows(promoter_pad_masks[b], L, B)
            keep += [pf, m]
            bs.promoter_feats[r], bs.promoter_mask_row[r], bs.promoter_mask_stride[r] = pf.data_ptr(), ptr, stride
            # cf_embed_full touches the promoter side only; the other pointers just have to be non-null
            bs.pcre_feats[r], bs.pcre_mask_row[r], bs.pcre_mask_stride[r], bs.interaction_mask[r] = pf.data_ptr(), ptr, stride, ptr
        bs.interaction_freq = keep[0].data_ptr()
        bs.B = B
        if B > self._max_batch:
            raise ValueError("batch of %d genes exceeds max_batch=%d given at construction" % (B, self._max_batch))
        outs = [torch.empty(B, 1, L, self.d_emb, device=self._device) for L in self.n_bins]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        st = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(_lib.lib().cf_embed_full(self._handle, C.byref(bs), ptrs, st), "cf_embed_full")
        return {b: o for b, o in zip(self.binsizes, outs)}

    def forward_backward(self, packed, labels, loss_scale=1.0):
        """Fused forward + loss + backward.  Returns (logits, loss tensor on device)."""
        bs, _ = packed
        labels = labels.to(self._device)
        labels = labels.float().contiguous() if self.n_out == 1 else labels.long().contiguous()
        st = torch.cuda.current_stream(self._device).cuda_stream
        # head forward + loss + head backward at the tail of the Regulation forward launch where the library can (cf_forward_train),
        # else as one launch inside the cf_backward below; logits / loss are filled by whichever runs
        logits = torch.empty(bs.B, self.n_out, device=self._device)
        self._sync_tiled(st)
        _lib.check(_lib.lib().cf_forward_train(self._handle, C.byref(bs), logits.data_ptr(), labels.data_ptr(), float(loss_scale),
                                               self._loss_buf.data_ptr(), st), "cf_forward_train")
        _lib.check(_lib.lib().cf_backward(self._handle, C.byref(bs), labels.data_ptr(), float(loss_scale),
                                          self._loss_buf.data_ptr(), st), "cf_backward")
        self._grads_stale = False
        return logits, self._loss_buf

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self._step += 1
        st = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(_lib.lib().cf_adamw_step(self._handle, float(lr), betas[0], betas[1], eps, weight_decay, self._step, st),
                   "cf_adamw_step")

    def active_grads(self):
        """The contiguous gradient range the optimiser / all-reduce operate on.  Raises after a step that never wrote it (the
        single-GPU Trainer applies AdamW in the epilogue of the gradient reductions and stores gradients only with
        keep_grads=True): stale values must not reach clip_grad_norm_ / logging / a custom all-reduce silently."""
        if getattr(self, "_grads_stale", False):
            raise RuntimeError("the gradient buffer was not written by the last step (Trainer(fuse_opt=True, keep_grads=False) "
                               "updates parameters inside the gradient reductions); construct the Trainer with keep_grads=True")
        return self._gflat[: self._layout.n_active]

    def _mark_grads(self, stale):
        """Called by the Trainer after every step: `stale` = the step did not store its gradients."""
        if stale and not getattr(self, "_grads_stale", False):
            for p in self._named().values():      # `.grad` views of the flat buffer (loss.backward() publishes them) would read old values
                p.grad = None
        self._grads_stale = bool(stale)

    def train_step(self, packed, labels, lr, process_group=None, world_size=1):
        """zero_grad -> forward -> loss -> backward -> [all-reduce] -> AdamW (train.py:182-196)."""
        logits, loss = self.forward_backward(packed, labels, loss_scale=1.0 / world_size)
        if world_size > 1:
            torch.distributed.all_reduce(self.active_grads(), group=process_group)
        self.adamw_step(lr)
        return logits, loss

    def debug_buffer(self, name):
        n = C.c_longlong()
        _lib.check(_lib.lib().cf_debug_copy(self._handle, name.encode(), None, C.byref(n), None), "cf_debug_copy")
        out = torch.empty(n.value, device=self._device)
        st = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(_lib.lib().cf_debug_copy(self._handle, name.encode(), out.data_ptr(), C.byref(n), st), "cf_debug_copy")
        return out

    def launch_counts(self):
        f, b, o = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().cf_launch_counts(self._handle, C.byref(f), C.byref(b), C.byref(o)), "cf_launch_counts")
        return f.value, b.value, o.value


class ChromoformerClassifier(ChromoformerBase):
    n_out = 2


class ChromoformerRegressor(ChromoformerBase):
    n_out = 1


_LEGACY_PREFIX = (("embed.", "embed"), ("pairwise_interaction.", "pw_int"), ("regulation.", "reg"))


class Chromoformer(ChromoformerClassifier):
    """The reference's original class (net.py:156-270): flat constructor arguments, a 16-positional-argument ``forward``
    (five tensors per resolution 2000 / 500 / 100, then the interaction frequencies) and modules named ``embed2000``,
    ``pw_int2000``, ``reg2000`` ... in its ``state_dict``.  Same construction order, so the same seed gives the same
    weights as ChromoformerClassifier (the reference's own smoke block checks that equality, net.py:431-568)."""

    def __init__(self, n_feats=7, embed_n_layers=1, embed_n_heads=2, embed_d_model=128, embed_d_ff=128, pw_int_n_layers=2,
                 pw_int_n_heads=2, pw_int_d_model=128, pw_int_d_ff=256, reg_n_layers=6, reg_n_heads=8, reg_d_model=256,
                 reg_d_ff=256, head_n_feats=128, seed=42, **kwargs):
        super().__init__(n_feats, embed_d_model, head_n_feats,
                         dict(n_layers=embed_n_layers, n_heads=embed_n_heads, d_model=embed_d_model, d_ff=embed_d_ff),
                         dict(n_layers=pw_int_n_layers, n_heads=pw_int_n_heads, d_model=pw_int_d_model, d_ff=pw_int_d_ff),
                         dict(n_layers=reg_n_layers, n_heads=reg_n_heads, d_model=reg_d_model, d_ff=reg_d_ff),
                         binsizes=(2000, 500, 100), seed=seed, **kwargs)

    def forward(self, x_p_2000, pad_mask_p_2000, x_pcre_2000, pad_mask_pcre_2000, interaction_mask_2000,
                x_p_500, pad_mask_p_500, x_pcre_500, pad_mask_pcre_500, interaction_mask_500,
                x_p_100, pad_mask_p_100, x_pcre_100, pad_mask_pcre_100, interaction_mask_100, interaction_freq):
        return super().forward({2000: x_p_2000, 500: x_p_500, 100: x_p_100},
                               {2000: pad_mask_p_2000, 500: pad_mask_p_500, 100: pad_mask_p_100},
                               {2000: x_pcre_2000, 500: x_pcre_500, 100: x_pcre_100},
                               {2000: pad_mask_pcre_2000, 500: pad_mask_pcre_500, 100: pad_mask_pcre_100},
                               {2000: interaction_mask_2000, 500: interaction_mask_500, 100: interaction_mask_100}, interaction_freq)

    @staticmethod
    def _legacy_key(k):
        for new, old in _LEGACY_PREFIX:
            if k.startswith(new):
                return old + k[len(new):]
        return k

    @staticmethod
    def _current_key(k):
        for new, old in _LEGACY_PREFIX:
            if k.startswith(old) and k[len(old):len(old) + 1].isdigit():
                return new + k[len(old):]
        return k

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        return type(sd)((self._legacy_key(k), v) for k, v in sd.items())

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict(type(state_dict)((self._current_key(k), v) for k, v in state_dict.items()), strict=strict)
