"""ctypes binding of libchromoformer_hip.so (include/chromoformer_hip.h).

The library is the product; this file is plumbing.  There is no fallback: if the shared
object is missing or a call fails, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CF_LIB_PATH") or os.path.join(_HERE, "libchromoformer_hip.so")      # override: A/B experiments only
FLAGS_PATH = LIB_PATH + ".flags"       # the CF_HIPCC_FLAGS the library was built with (part of the reuse check of build())
CSRC = os.path.join(_HERE, "csrc")
MAX_RES = 3
BUCKET_REG, BUCKET_PE = 1, 2
BUCKET_REG_HI, BUCKET_REG_LO = 4, 8      # the Regulation bucket in halves (cf_reg_halves): upper layers + head, lower layers
PART_REG_HI, PART_REG_LO = 8, 16        # cf_backward_part: the Regulation backward as two launches


class cf_config(C.Structure):
    _fields_ = [
        ("n_feats", C.c_int), ("d_emb", C.c_int), ("d_head", C.c_int), ("n_out", C.c_int), ("n_res", C.c_int),
        ("binsizes", C.c_int * MAX_RES), ("n_bins", C.c_int * MAX_RES), ("i_max", C.c_int),
        ("embed_layers", C.c_int), ("embed_heads", C.c_int), ("embed_dmodel", C.c_int), ("embed_dff", C.c_int),
        ("pair_layers", C.c_int), ("pair_heads", C.c_int), ("pair_dmodel", C.c_int), ("pair_dff", C.c_int),
        ("reg_layers", C.c_int), ("reg_heads", C.c_int), ("reg_dmodel", C.c_int), ("reg_dff", C.c_int),
        ("max_batch", C.c_int),
    ]


class cf_param_desc(C.Structure):
    _fields_ = [("name", C.c_char * 120), ("ndim", C.c_int), ("shape", C.c_int * 2), ("offset", C.c_longlong),
                ("numel", C.c_longlong), ("trainable", C.c_int)]


class cf_layout(C.Structure):
    _fields_ = [("n_tensors", C.c_int), ("n_total", C.c_longlong), ("n_active", C.c_longlong), ("n_elems", C.c_longlong)]


class cf_attn_shape(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int),
                ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int)]


class cf_dense_layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "wkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")] + [("d_ff", C.c_int)]


class cf_dense_layer_grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "wkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")]


class cf_batch(C.Structure):
    _fields_ = [
        ("B", C.c_int),
        ("promoter_feats", C.c_void_p * MAX_RES), ("pcre_feats", C.c_void_p * MAX_RES),
        ("promoter_mask_row", C.c_void_p * MAX_RES), ("promoter_mask_stride", C.c_longlong * MAX_RES),
        ("pcre_mask_row", C.c_void_p * MAX_RES), ("pcre_mask_stride", C.c_longlong * MAX_RES),
        ("interaction_mask", C.c_void_p * MAX_RES), ("interaction_freq", C.c_void_p),
    ]


class cf_store(C.Structure):
    _fields_ = [
        ("n_genes", C.c_longlong),
        ("promoter_feats", C.c_void_p * MAX_RES), ("pcre_feats", C.c_void_p * MAX_RES),
        ("promoter_mask", C.c_void_p * MAX_RES), ("pcre_mask", C.c_void_p * MAX_RES),
        ("interaction_mask", C.c_void_p), ("interaction_freq", C.c_void_p), ("labels", C.c_void_p),
    ]


#: every symbol include/chromoformer_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "cf_abi_version": (C.c_int, []),
    "cf_last_error": (C.c_char_p, []),
    "cf_param_layout": (C.c_int, [C.POINTER(cf_config), C.POINTER(cf_layout), C.POINTER(cf_param_desc), C.c_int]),
    "cf_create": (C.c_int, [C.POINTER(cf_config), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "cf_destroy": (None, [C.c_void_p]),
    "cf_bind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_forward": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_int, C.c_void_p]),
    "cf_backward": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cf_backward_chain": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cf_backward_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cf_reg_halves": (C.c_int, [C.c_void_p]),
    "cf_grad_bucket": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "cf_backward_reduce_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cf_adamw_step_part": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_int, C.c_void_p]),
    "cf_adamw_set": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_void_p]),
    "cf_adamw_step_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cf_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_backward_part": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "cf_forward_train": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "cf_head_rides": (C.c_int, [C.c_void_p]),
    "cf_kernel_flops": (C.c_double, [C.c_void_p, C.c_char_p, C.c_int]),
    "cf_cu_count": (C.c_int, [C.c_void_p]),
    "cf_keep_tiled": (C.c_int, [C.c_void_p, C.c_int]),
    "cf_params_changed": (C.c_int, [C.c_void_p]),
    "cf_retile_early": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cf_capture_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cf_capture_end": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "cf_graph_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cf_timing_select": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cf_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "cf_wgrad_flops": (C.c_double, [C.c_void_p, C.c_int]),
    "cf_backward_from": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p]),
    "cf_adamw_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_void_p]),
    "cf_debug_copy": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p]),
    "cf_debug_names": (C.c_char_p, [C.c_void_p]),
    "cf_reduce_adamw_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_int,
                                       C.c_void_p]),
    "cf_reduce_opt_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_int,
                                     C.c_void_p]),
    "cf_rider_arm": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_int, C.c_int]),
    "cf_launch_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cf_op_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cf_op_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cf_op_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cf_bin_regions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cf_bin_regions_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p]),
    "cf_embed_full": (C.c_int, [C.c_void_p, C.POINTER(cf_batch), C.POINTER(C.c_void_p), C.c_void_p]),
    "cf_gather_batch": (C.c_int, [C.c_void_p, C.POINTER(cf_store), C.c_void_p, C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p]),
    "cf_gather_batch_fwd": (C.c_int, [C.c_void_p, C.POINTER(cf_store), C.c_void_p, C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p]),
    "cf_gather_batch_next": (C.c_int, [C.c_void_p, C.POINTER(cf_store), C.c_void_p, C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p]),
    "cf_gather_batch_only": (C.c_int, [C.c_void_p, C.POINTER(cf_store), C.c_void_p, C.c_void_p, C.POINTER(cf_batch), C.c_void_p, C.c_void_p]),
    "cf_record_step_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_record_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_op_dense_layer_workspace": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "cf_op_dense_layer_fwd": (C.c_int, [C.POINTER(cf_dense_layer)] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 3),
    "cf_op_dense_layer_train_workspace": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "cf_op_dense_layer_fwd_train": (C.c_int, [C.POINTER(cf_dense_layer)] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 3),
    "cf_op_dense_layer_bwd": (C.c_int, [C.POINTER(cf_dense_layer)] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 3
                              + [C.POINTER(cf_dense_layer_grads)] + [C.c_void_p] * 3),
    "cf_op_attention_fwd": (C.c_int, [C.POINTER(cf_attn_shape)] + [C.c_void_p] * 9),
    "cf_op_attention_bwd": (C.c_int, [C.POINTER(cf_attn_shape)] + [C.c_void_p] * 14),
}

_lib = None


def build(force=False, verbose=False):
    """Compile csrc/ for gfx950 into libchromoformer_hip.so (hipcc cross-compiles without a GPU).  The library is reused only when
    it is newer than every source AND was built with the same extra flags (CF_HIPCC_FLAGS, recorded in a sidecar file): an A/B
    experiment can neither pick up the shipped library silently nor leave its own behind as the shipped one.  Says which it did."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "chromoformer_hip.h"))
    flags = " ".join(os.environ.get("CF_HIPCC_FLAGS", "").split())
    try:
        built_with = open(FLAGS_PATH).read().strip()
    except OSError:
        built_with = ""
    fresh = os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh and built_with == flags:
        if verbose:
            print("build: reused %s (newer than all %d sources, flags '%s')" % (LIB_PATH, len(srcs), flags))
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed",
           os.path.join(CSRC, "cf_api.hip"), "-o", LIB_PATH]
    cmd[6:6] = flags.split()      # experiments (-DCF_...); the shipped library is built without
    if verbose:
        why = "forced" if force else ("flags changed: '%s' -> '%s'" % (built_with, flags) if fresh else ("no library yet" if not os.path.exists(LIB_PATH) else "sources newer than the library"))
        print("build: compiling (%s): %s" % (why, " ".join(cmd)))
    subprocess.run(cmd, check=True)
    with open(FLAGS_PATH, "w") as f:
        f.write(flags + "\n")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libchromoformer_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU / PyTorch fallback for the Chromoformer hot path" % LIB_PATH)
        # the library on disk must be the one the current environment asks for: an A/B build (CF_HIPCC_FLAGS) left behind by an experiment is
        # refused instead of silently used by every later run that does not call build()
        flags = " ".join(os.environ.get("CF_HIPCC_FLAGS", "").split())
        try:
            built_with = open(FLAGS_PATH).read().strip()
        except OSError:
            built_with = ""      # (a library without the sidecar was built by build() before the sidecar existed, or by hand: plain flags)
        if built_with != flags:
            raise RuntimeError("libchromoformer_hip.so was built with CF_HIPCC_FLAGS='%s' but this process runs with '%s': rebuild "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`) or set the variable to match" % (built_with, flags))
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.cf_abi_version() != 1:
            raise RuntimeError("libchromoformer_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libchromoformer_hip: %s failed: %s" % (what, lib().cf_last_error().decode()))


def make_config(n_feats, d_emb, d_head, n_out, binsizes, n_bins, i_max, embed, pair, reg, max_batch):
    c = cf_config()
    c.n_feats, c.d_emb, c.d_head, c.n_out, c.n_res = n_feats, d_emb, d_head, n_out, len(binsizes)
    if len(binsizes) > MAX_RES:
        raise ValueError("at most %d resolutions" % MAX_RES)
    for i, (b, n) in enumerate(zip(binsizes, n_bins)):
        c.binsizes[i], c.n_bins[i] = int(b), int(n)
    c.i_max = i_max
    c.embed_layers, c.embed_heads, c.embed_dmodel, c.embed_dff = (embed[k] for k in ("n_layers", "n_heads", "d_model", "d_ff"))
    c.pair_layers, c.pair_heads, c.pair_dmodel, c.pair_dff = (pair[k] for k in ("n_layers", "n_heads", "d_model", "d_ff"))
    c.reg_layers, c.reg_heads, c.reg_dmodel, c.reg_dff = (reg[k] for k in ("n_layers", "n_heads", "d_model", "d_ff"))
    c.max_batch = max_batch
    return c


def param_layout(cfg):
    """-> (cf_layout, [dict(name, shape, offset, numel, trainable)]) in state_dict order.  Host only."""
    L = lib()
    lay = cf_layout()
    check(L.cf_param_layout(C.byref(cfg), C.byref(lay), None, 0), "cf_param_layout")
    tab = (cf_param_desc * lay.n_tensors)()
    check(L.cf_param_layout(C.byref(cfg), C.byref(lay), tab, lay.n_tensors), "cf_param_layout")
    out = []
    for d in tab:
        out.append(dict(name=d.name.decode(), shape=tuple(d.shape[:d.ndim]), offset=int(d.offset), numel=int(d.numel),
                        trainable=bool(d.trainable)))
    return lay, out
