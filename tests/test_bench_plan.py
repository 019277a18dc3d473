"""bench.py's multi-GPU plumbing that needs no GPU: the strong-scaling partition (both assignment rules) and the refusal to
start N ranks on fewer devices."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))


def test_plan_only_reports_both_assignments():
    r = _bench("--plan-only", "--gpus", "8", "--genomes", "1", "--records", "20000")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["ranks"] == 8 and len(d["ops_per_rank_hash"]) == 8 and len(d["ops_per_rank_lpt"]) == 8
    assert abs(sum(d["ops_per_rank_hash"]) - sum(d["ops_per_rank_lpt"])) < 1
    # 24 contigs of very different sizes over 8 ranks: the hash rule is visibly uneven, the size-aware one is not
    assert d["imbalance_max_over_mean_hash"] > 1.15 and d["imbalance_max_over_mean_lpt"] < 1.1


def test_lpt_keeps_a_target_on_one_rank():
    sys.path.insert(0, ROOT)
    from wgatools_amd import multigpu
    rng = np.random.default_rng(1)
    names, mb = multigpu.human_like_targets(3)
    tid = rng.choice(len(names), size=5000, p=mb / mb.sum())
    rec = [names[k] for k in tid]
    w = rng.integers(1, 10000, 5000)
    for rule in (multigpu.owners(rec, 5), multigpu.owners_lpt(rec, w, 5)):
        seen = {}
        for t, o in zip(rec, rule):
            assert seen.setdefault(t, int(o)) == int(o)
        assert set(int(x) for x in rule) <= set(range(5))
    per = np.bincount(multigpu.owners_lpt(rec, w, 5), weights=w, minlength=5)
    assert per.max() / per.mean() < 1.1


def test_gpus_n_without_a_launcher_refuses_missing_devices():
    r = _bench("--gpus", "2")
    assert r.returncode == 2 and "device(s) visible" in r.stderr


def test_traffic_comes_from_the_newest_rounds_counter_passes():
    """roofline.traffic is read from the committed PMC summary of the newest round whose workload and row-kernel source both
    match what bench.py is about to time; anything else must give null, never an older round's figure."""
    import glob

    sys.path.insert(0, ROOT)
    import bench

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    assert files, "no committed counter passes"
    newest = json.load(open(files[-1]))

    class Args:
        records = newest["workload"]["records"]
        mean_ops = newest["workload"]["mean_ops"]

    class Job:
        n_ops = newest["workload"]["ops"]

    got = bench.pmc_traffic(Args, Job)
    if newest["kernel_source_sha"] == bench.kernel_source_sha():
        assert got == newest["hbm_bytes_per_launch"]
        assert 1.0 <= got / (4 * Job.n_ops) < 20.0
    else:
        assert got is None or got != newest["hbm_bytes_per_launch"]
    Job.n_ops += 1
    assert bench.pmc_traffic(Args, Job) is None
