"""A/B runs of bench.py under environment variants (library switches are read when a model is constructed):
    python tools/ab_env.py "CF_DEFER_MAX=128" "CF_DEFER_MAX=256" "CF_DEFER_TILES=0" [--rounds 2] [--steps 300]
Prints ms_per_step (eager launches) / graph_replay_ms_per_step per variant and round."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
rounds, steps = int(opt.get("rounds", 2)), int(opt.get("steps", 300))
for rnd in range(rounds):
    for spec in args:
        env = dict(os.environ)
        for kv in spec.split(","):
            if kv and kv != "-":
                k, v = kv.split("=")
                env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "30", "--no-cpu-baseline", "--no-dp-path",
                            "--train-loop-steps", "0", "--no-val-auroc", "--no-extras"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("round %d  %-40s  %.4f ms  graph %.4f ms  deferred %s  loss %s" % (rnd, spec, d["ms_per_step"], d.get("graph_replay_ms_per_step", float("nan")),
                                                                              d["config"].get("deferred_tiles"), d["loss"]), flush=True)
        except Exception:
            print("round %d  %-40s  FAILED rc %d: %s" % (rnd, spec, r.returncode, r.stderr[-300:]), flush=True)
