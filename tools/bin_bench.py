"""HBM roofline of the binning kernel (cf_bin_regions): python tools/bin_bench.py [--regions 4096]
Promoter-sized regions (fp16 [7, 40000] = 560 KB each), all three default resolutions; algorithmic bytes =
2 B per input sample + 4 * 7 B per output bin + 1 mask byte per bin, per resolution pass."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import _lib
from chromoformer_amd.data import BIN_JOB

ap = argparse.ArgumentParser()
ap.add_argument("--regions", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
R, F, LEN = a.regions, 7, 40000
raw = (torch.rand(R, F, LEN, device=dev) * 4).half()
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
out = {}
for b in (2000, 500, 100):
    L = LEN // b
    feats = torch.empty(R, L, F, device=dev)
    mask = torch.empty(R, L, dtype=torch.uint8, device=dev)
    jobs = np.zeros(R, dtype=BIN_JOB)
    jobs["raw"] = raw.data_ptr() + np.arange(R, dtype=np.uint64) * (F * LEN * 2)
    jobs["ld"], jobs["col0"], jobs["ncols"], jobs["flip"] = LEN, 0, LEN, np.arange(R) % 2
    jobs["out"] = feats.data_ptr() + np.arange(R, dtype=np.uint64) * (L * F * 4)
    jobs["mask"] = mask.data_ptr() + np.arange(R, dtype=np.uint64) * L
    tab = torch.from_numpy(jobs.view(np.uint8)).to(dev)
    run = lambda: _lib.check(lib.cf_bin_regions(C.c_void_p(tab.data_ptr()), R, F, b, L, st), "cf_bin_regions")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    nbytes = R * (F * LEN * 2 + L * F * 4 + L)
    # spot check against torch on the device
    ref = torch.log1p(raw[:8].float().reshape(8, F, L, b).mean(3)).permute(0, 2, 1)
    ref[1::2] = ref[1::2].flip(1)
    err = float((feats[:8] - ref).abs().max())
    out[b] = {"ms": round(ms, 3), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3), "max_err": err}
# all three resolutions in ONE launch, the raw bytes read once (cf_bin_regions_multi)
from chromoformer_amd.data import BIN_JOB_MULTI
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
run = lambda: _lib.check(lib.cf_bin_regions_multi(C.c_void_p(tab.data_ptr()), R, F, 3, cb, cl, LEN, st), "cf_bin_regions_multi")
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
nbytes = R * (F * LEN * 2 + sum(L * F * 4 + L for L in Ls))
err = 0.0
for r, b in enumerate(bsz):
    ref = torch.log1p(raw[:8].float().reshape(8, F, Ls[r], b).mean(3)).permute(0, 2, 1)
    ref[1::2] = ref[1::2].flip(1)
    err = max(err, float((feats[r][:8] - ref).abs().max()))
one = {"launches": 1, "ms": round(ms, 3), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3), "max_err": err,
       "raw_KB_per_region": F * LEN * 2 / 1000, "three_passes_ms": round(sum(v["ms"] for v in out.values()), 3)}
print(json.dumps({"workload": "%d regions x fp16 [7, 40000]" % R, "one_pass_all_resolutions": one, "per_bin_size": out}))
