"""Stress configuration (BASELINE.json configs[3]): the dense attention core at 800-bin
sequences, all L x L rows, forward + backward, timed with HIP (torch) events on the launch stream.
    python tools/stress_bench.py [--n-seq 2176] [--reps 5]
i_max = 16, bsz = 128 -> N = 128 * (1 + 16) = 2176 sequences, h = 2, dh = 64, L = 800.
Flops, the usual accounting of attention kernels: forward 4 N H L^2 dh (QK^T and PV), backward 10 N H L^2 dh = 2.5 x forward (dV, dP, dQ, dK and
the QK^T that a backward without a stored P has to execute again).  `frac` uses these 14; `frac_no_recompute` counts the backward without that
re-execution (8 N H L^2 dh: SURVEY.md section 8d's "training = 3 x forward").  Also timed: whole dense layers, forward
(cf_op_dense_layer_fwd: projections, attention, out-projection, LN, FFN, LN over all rows)."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n-seq", type=int, default=2176)
ap.add_argument("--L", type=int, default=800)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
N, H, L = args.n_seq, 2, args.L
dev = torch.device("cuda", 0)
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
    _lib.check(lib.cf_op_attention_fwd(C.byref(sh), p(q), p(k), p(v), p(valid), p(valid), None, p(o), p(stats), st), "fwd")


def bwd():
    _lib.check(lib.cf_op_attention_bwd(C.byref(sh), p(q), p(k), p(v), p(valid), p(valid), None, p(o), p(stats), p(d_o), p(dq), p(dk), p(dv), p(ws), st), "bwd")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


# whole dense layers, forward: the Embedding layer on 128 promoters, the Pairwise layer on 128 x 16 (promoter, pCRE) pairs
def layer_bench(n_seq, dff):
    wts = {k: torch.randn(*shape, device=dev, generator=g) * 0.05 for k, shape in
           dict(wq=(128, 128), wkv=(256, 128), wo=(128, 128), bo=(128,), g1=(128,), b1n=(128,), w1=(dff, 128), b1=(dff,), w2=(128, dff), b2=(128,),
                g2=(128,), b2n=(128,)).items()}
    w = _lib.cf_dense_layer()
    for f, k in (("wq", "wq"), ("wkv", "wkv"), ("wo", "wo"), ("bo", "bo"), ("ln1_g", "g1"), ("ln1_b", "b1n"), ("w1", "w1"), ("b1", "b1"), ("w2", "w2"),
                 ("b2", "b2"), ("ln2_g", "g2"), ("ln2_b", "b2n")):
        setattr(w, f, wts[k].data_ptr())
    w.d_ff = dff
    xq = torch.randn(n_seq, L, 128, device=dev, generator=g)
    xk = torch.randn(n_seq, L, 128, device=dev, generator=g)
    y = torch.empty_like(xq)
    ws_l = torch.empty(lib.cf_op_dense_layer_train_workspace(n_seq, L, L, dff), device=dev)
    vl = torch.ones(n_seq, L, dtype=torch.uint8, device=dev)
    run = lambda: _lib.check(lib.cf_op_dense_layer_fwd(C.byref(w), p(xq), p(xk), p(vl), p(vl), None, n_seq, L, L, p(y), p(ws_l), st), "layer")
    ms = timed(run)
    lin = n_seq * (2.0 * L * 128 * 384 + 2.0 * L * 128 * 128 + 4.0 * L * 128 * dff)
    att = n_seq * 4.0 * L * L * 128
    out = {"fwd_ms": round(ms, 3), "fwd_tflops": round((lin + att) / ms / 1e9, 2)}
    # training: forward with saves + backward (input gradients, split-K weight gradients, bias / LayerNorm gradients)
    grads = {k: torch.empty_like(v) for k, v in wts.items()}
    gs = _lib.cf_dense_layer_grads()
    for f, k in (("wq", "wq"), ("wkv", "wkv"), ("wo", "wo"), ("bo", "bo"), ("ln1_g", "g1"), ("ln1_b", "b1n"), ("w1", "w1"), ("b1", "b1"), ("w2", "w2"),
                 ("b2", "b2"), ("ln2_g", "g2"), ("ln2_b", "b2n")):
        setattr(gs, f, grads[k].data_ptr())
    dyl, dxq, dxk = torch.randn_like(xq), torch.empty_like(xq), torch.empty_like(xk)
    tables = torch.empty(1 << 20, device=dev)

    def train():
        _lib.check(lib.cf_op_dense_layer_fwd_train(C.byref(w), p(xq), p(xk), p(vl), p(vl), None, n_seq, L, L, p(y), p(ws_l), st), "fwd_train")
        _lib.check(lib.cf_op_dense_layer_bwd(C.byref(w), p(xq), p(xk), p(vl), p(vl), None, n_seq, L, L, p(dyl), p(dxq), p(dxk), C.byref(gs),
                                             p(ws_l), p(tables), st), "bwd")
    mt = timed(train)
    out.update({"fwd_bwd_ms": round(mt, 3), "fwd_bwd_tflops": round((3.0 * lin + 3.5 * att) / mt / 1e9, 2),
                "fwd_bwd_frac": round((3.0 * lin + 3.5 * att) / mt / 1e9 / 157.3, 4)})
    return out


tf, tb = timed(fwd), timed(bwd)
ff, fb = 4.0 * N * H * L * L * 64, 10.0 * N * H * L * L * 64
print(json.dumps({"workload": "dense attention core, N=%d sequences x %d heads, L=%d, dh=64, f32" % (N, H, L),
                  "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3),
                  "fwd_tflops": round(ff / tf / 1e9, 2), "bwd_tflops": round(fb / tb / 1e9, 2),
                  "fwd_bwd_tflops": round((ff + fb) / (tf + tb) / 1e9, 2), "peak_tflops_f32_mfma": 157.3,
                  "frac": round((ff + fb) / (tf + tb) / 1e9 / 157.3, 4), "frac_no_recompute": round((ff + 0.8 * fb) / (tf + tb) / 1e9 / 157.3, 4),
                  "embedding_layer_fwd (N=%d, d_ff 128)" % max(1, N // 17): layer_bench(max(1, N // 17), 128),
                  "pairwise_layer_fwd (N=%d, d_ff 256)" % (N - max(1, N // 17)): layer_bench(N - max(1, N // 17), 256)}))
