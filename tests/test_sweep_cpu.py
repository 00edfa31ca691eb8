"""CV-sweep scheduler (replicas over GPUs, chromoformer/Snakefile): job plan, command lines, GPU pinning, resume."""
import argparse
import os

from chromoformer_amd import sweep


class _Proc:
    def __init__(self, cmd, env, log):
        self.cmd, self.env, self.polls = cmd, env, 0

    def poll(self):
        self.polls += 1
        if self.polls < 2:
            return None
        ckpt = self.cmd[self.cmd.index("-o") + 1]
        open(ckpt, "w").write("ckpt")                                        # the training wrote its checkpoint ...
        if not getattr(self, "crash", False):
            open(ckpt + ".done", "w").write("epochs 9\n")                    # ... and reached its last epoch
            return 0
        return 1


def test_plan_commands_and_gpu_assignment(tmp_path):
    jobs = sweep.plan(["E003", "E004"], ["1", "2", "3", "4"], "exp", "1", str(tmp_path / "ckpts"))
    assert len(jobs) == 8 and jobs[0][2].endswith(os.path.join("E003", "exp-E003-conf1-fold1.pt"))
    args = argparse.Namespace(config="c.yaml", exp_id="exp", meta_template="d/{eid}/train.csv", npy_dir_template="d/{eid}/npy",
                              binsizes=None, regression=True, gpus=3, poll=0.0)
    started = []

    def launch(cmd, env, stdout, stderr):
        p = _Proc(cmd, env, stdout)
        started.append(p)
        return p

    done = sweep.run(jobs, args, launch=launch)
    assert len(done) == 8 and set(done.values()) == {0}
    assert {p.env["HIP_VISIBLE_DEVICES"] for p in started} == {"0", "1", "2"}
    c = started[3].cmd                                    # E003 fold 4 -> train.py's fold 0
    assert c[c.index("--fold") + 1] == "0" and "--regression" in c and c[c.index("-m", 3) + 1] == "d/E003/train.csv"
    # resume: everything exists now, nothing is launched again
    started.clear()
    assert sweep.run(jobs, args, launch=launch) == done and not started


def test_unfinished_and_failed_jobs_are_run_again(tmp_path):
    """A checkpoint without its `.done` marker (killed after an early epoch) is not a finished job; a job that exits
    non-zero leaves nothing behind that a later sweep could mistake for a result (chromoformer/Snakefile semantics:
    Snakemake removes incomplete outputs)."""
    jobs = sweep.plan(["E003"], ["1", "2"], "exp", "1", str(tmp_path / "ckpts"))
    args = argparse.Namespace(config="c.yaml", exp_id="exp", meta_template="d/{eid}/train.csv", npy_dir_template="d/{eid}/npy",
                              binsizes=None, regression=False, gpus=2, poll=0.0)
    os.makedirs(os.path.dirname(jobs[0][2]))
    open(jobs[0][2], "w").write("partial: epoch 1 of 9")                     # left by a killed run
    started = []

    def launch(cmd, env, stdout, stderr):
        p = _Proc(cmd, env, stdout)
        p.crash = cmd[cmd.index("-o") + 1] == jobs[1][2]
        started.append(p)
        return p

    done = sweep.run(jobs, args, launch=launch)
    assert len(started) == 2                                                  # the partial checkpoint did not count
    assert done[jobs[0][2]] == 0 and done[jobs[1][2]] == 1
    assert os.path.exists(jobs[0][2] + ".done") and not os.path.exists(jobs[1][2])
    started.clear()
    done = sweep.run(jobs, args, launch=launch)                               # only the failed job runs again
    assert [p.cmd[p.cmd.index("-o") + 1] for p in started] == [jobs[1][2]]


def test_two_jobs_per_gpu(tmp_path):
    """`--jobs-per-gpu 2`: two replicas share a device at a time (measured +25 % sweep throughput on one MI355X); every GPU gets its first job
    before any gets a second one, and never more than two run on one device."""
    jobs = sweep.plan(["E003", "E004"], ["1", "2", "3", "4"], "exp", "1", str(tmp_path / "ckpts"))
    args = argparse.Namespace(config="c.yaml", exp_id="exp", meta_template="d/{eid}/train.csv", npy_dir_template="d/{eid}/npy",
                              binsizes=None, regression=False, gpus=2, jobs_per_gpu=2, poll=0.0)
    started, peak = [], {}

    def launch(cmd, env, stdout, stderr):
        p = _Proc(cmd, env, stdout)
        started.append(p)
        live = [q for q in started if q.polls < 2]
        for g in ("0", "1"):
            peak[g] = max(peak.get(g, 0), sum(q.env["HIP_VISIBLE_DEVICES"] == g for q in live))
        return p

    done = sweep.run(jobs, args, launch=launch)
    assert len(done) == 8 and set(done.values()) == {0}
    assert [p.env["HIP_VISIBLE_DEVICES"] for p in started[:4]] == ["0", "1", "0", "1"]
    assert peak == {"0": 2, "1": 2}
