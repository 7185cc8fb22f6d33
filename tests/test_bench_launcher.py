"""bench.py's process plumbing, where there is no GPU: `--dry-run` replaces the kernels by a sleep and RCCL by gloo;
everything else (self-launch under torch.distributed.run, one child process per measuring leg, barrier + max-over-ranks
timing, the result gather, ONE JSON line on stdout, fault isolation) is the code the GPU run uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
FAST = ["--dry-run", "--steps", "4", "--warmup", "1", "--min-seconds", "0.02"]


def run(cmd, timeout=240):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p, lines


def check_line(d, n):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n and d["steps"] == 4 and d["warmup"] == 1
    assert d["config"]["rccl_ranks_seen"] == n
    assert d["scaling"] == "weak" and d["value"] > 0
    assert "dtw_oracle" in d["config"] and "unpinned" in d["config"]["dtw_oracle"]
    if n > 1:
        # a scaling curve that can be read: every rank's own time, their skew, what the result gather costs
        assert len(d["per_rank"]["ms_per_step"]) == n and d["per_rank"]["min"] <= d["per_rank"]["max"]
        assert d["per_rank"]["skew_max_over_min"] >= 1.0
        assert 0.0 <= d["result_gather"]["share_of_step"] < 1.0 and d["result_gather"]["ms_per_step_without_gather"] > 0
        assert d["cpu_baseline"] == "N=1 line only"
        assert "roofline" in d and d["roofline"]["bound"] == "hbm"
        # each rank keeps to its share of the host's cores
        assert d["config"]["cpu_threads_per_rank"] <= max(1, (os.cpu_count() or n) // n)
        # BASELINE configs[3] on N ranks: recordings across the ranks, decoder streams within a rank, results gathered
        tr = d["transcribe_recordings"]
        assert "error" not in tr, tr
        assert tr["ranks"] == n and len(tr["per_rank_seconds"]) == n and tr["recordings"] == 2 * n
        assert tr["parity_vs_1_stream_on_rank_0"]["max_abs_dt_word_s"] <= 0.02


def test_one_rank_prints_exactly_one_json_line():
    p, lines = run([sys.executable, BENCH, *FAST])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    check_line(d, 1)
    assert d["config"]["result_gather"] == "none"
    assert d["kernel_leg_attempts"] == 1
    assert d["faulted"] is False


def test_gpus_2_launches_itself_under_torchrun():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver uses for N=1)."""
    p, lines = run([sys.executable, BENCH, "--gpus", "2", *FAST])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    check_line(d, 2)
    assert "gather to rank 0" in d["config"]["result_gather"]


def test_driver_form_torchrun_two_ranks():
    p, lines = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29641", BENCH, "--gpus", "2", *FAST])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines
    check_line(json.loads(lines[0]), 2)


@pytest.mark.timeout(900)
def test_eight_ranks_self_launched():
    """The node's shape: eight ranks (gloo here, RCCL on the MI355X node no round has had): one line, rccl_ranks_seen 8,
    per_rank of length 8, the recordings leg gathered from 8 ranks -- self-launched (the driver's torchrun form: the two-rank
    test above; the launcher is the same program at any N)."""
    p, lines = run([sys.executable, BENCH, "--gpus", "8", *FAST], timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    check_line(json.loads(lines[0]), 8)


def test_a_dead_kernel_leg_is_rerun_and_reported():
    p, lines = run([sys.executable, BENCH, *FAST, "--inject-fault", "kernel"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[0])
    check_line(d, 1)
    assert d["kernel_leg_attempts"] == 2
    assert d["faulted"] is True
    assert d["kernel_leg_first_attempt"]["error"] == "signal 6"


def test_children_publish_partial_results(tmp_path):
    """role plumbing: a child that aborts after publishing leaves its partial result for the parent."""
    sys.path.insert(0, ROOT)
    import bench
    argv = sys.argv
    try:
        sys.argv = [BENCH, "--inject-fault", "e2e_fp16"]
        res, err = bench.run_child("e2e", ["--leg", "fp16"], 120)
    finally:
        sys.argv = argv
    assert err == "signal 6"
    assert res == {"marker": "about to abort"}
