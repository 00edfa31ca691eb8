"""End-to-end sanity of the drop-in entrypoint on a synthetic Roadmap-style dataset with a planted signal (expressed genes
carry more marks, tests/synth_data.py): `python -m chromoformer_amd.train` for a few epochs, validation AUROC and wall time
per epoch.  python tools/train_synth_demo.py [--genes 1024] [--epochs 6]"""
import argparse, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
from tests.synth_data import make_dataset

ap = argparse.ArgumentParser()
ap.add_argument("--genes", type=int, default=1024)
ap.add_argument("--epochs", type=int, default=6)
a = ap.parse_args()
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    meta = make_dataset(os.path.join(d, "npy"), n_genes=a.genes, seed=7)
    print("dataset: %d genes written in %.1f s" % (a.genes, time.time() - t0))
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = 64, a.epochs + 1
    yaml.safe_dump(cfg, open(os.path.join(d, "cfg.yaml"), "w"))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "chromoformer_amd.train", "-o", os.path.join(d, "ck.pt"), "-c", os.path.join(d, "cfg.yaml"),
                        "--exp-id", "synth", "-m", meta, "-d", os.path.join(d, "npy"), "--fold", "0"], cwd=ROOT, capture_output=True, text=True)
    print("train.py: rc %d, %.1f s wall (binning on the GPU, %d epochs of %d training genes)" % (r.returncode, time.time() - t0, a.epochs, a.genes * 3 // 4))
    for ln in (r.stdout + r.stderr).splitlines():
        if re.search(r"auc|AUC|Val|val", ln):
            print("  " + ln.strip()[:160])
    if r.returncode:
        print(r.stderr[-2000:])
