"""`bench.py --gpus N` must really run N ranks (one process per GPU): the launcher path of the
script (spawn through torch.distributed.run before anything touches a GPU, shard, all-gather,
max-over-ranks timing, one JSON line from rank 0) exercised on the CPU with the gloo backend and
a stand-in scorer (`--stub`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "2", "--warmup", "1",
                           "--candidates", "16", *extra], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                      # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["world"] == 2
    assert line["all_ranks_hold_all_energies"] is True
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"


def test_gpus_1_is_single_process():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["world"] == 1


def test_rank_count_mismatch_fails():
    # a launcher that started one rank while --gpus asks for two must not print a 1-rank line
    r = _run("--gpus", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rank_that_skips_a_collective_fails_the_run_but_the_line_is_printed():
    """A rank that never joins a collective of the multi-rank end-to-end phase (here: rank 1 skips it) must not
    end in status 0: the watchdog prints the already complete throughput line with an error note, and the run's
    exit status is non-zero (bench.py EXIT_HUNG / EXIT_CLOSE_HUNG), so the driver sees the failure."""
    r = _run("--gpus", "2", env_extra={"ASW_STUB_FAULT_RANK": "1", "ASW_E2E_TIMEOUT_S": "5", "ASW_CLOSE_TIMEOUT_S": "10"})
    assert r.returncode != 0, r.stdout[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["all_ranks_hold_all_energies"] is True
    assert "timed out" in line["e2e_latency"]["error"]
