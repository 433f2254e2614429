"""bench.py --gpus N without a launcher must start N child ranks itself (scripts/cs/pretrain.sh:31,
engine.py:38-46), forward exactly rank 0's JSON line, and refuse to run when fewer devices are
visible.  Exercised here with a stub worker: env / argv / JSON plumbing only, no GPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = [sys.executable, os.path.join(ROOT, "tests", "_bench_stub_worker.py")]


def _run(code, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=120)


_CALL = ("import sys, bench; sys.exit(bench.launch_ranks({n}, ['--gpus', '{n}', '--steps', '3'], "
         "worker={stub!r}, device_count={have}))")


def test_launcher_spawns_n_ranks_and_forwards_rank0_json():
    r = _run(_CALL.format(n=2, stub=STUB, have=2))
    assert r.returncode == 0, r.stderr
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, out                       # ONE JSON line on stdout
    rec = json.loads(out[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["rank"] == 0 and rec["local_rank"] == 0
    assert rec["master"] == "127.0.0.1" and rec["argv"] == ["--gpus", "2", "--steps", "3"]
    assert rec["config"] == {"global_batch": 8, "parallelism": "dp2"}
    # rank 1 really ran, with its own RANK/LOCAL_RANK and the same rendezvous port
    other = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert len(other) == 1 and other[0]["rank"] == 1 and other[0]["local_rank"] == 1
    assert other[0]["port"] == rec["port"]


def test_launcher_refuses_when_too_few_devices():
    r = _run(_CALL.format(n=2, stub=STUB, have=1))
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "only 1 GPU(s) visible" in r.stderr


def test_launcher_propagates_rank_failure_and_stops_the_others():
    t0 = time.time()
    r = _run(_CALL.format(n=2, stub=STUB, have=2), {"STUB_FAIL_RANK": "1", "STUB_HANG_RANK": "0"})
    assert r.returncode == 7 and r.stdout.strip() == ""
    assert time.time() - t0 < 45                    # the hanging rank 0 was terminated, not waited for


def test_launcher_deadline_stops_a_stuck_rank_and_says_which():
    # config 4 has never met RCCL at N > 1: a rank stuck in the rendezvous / a collective must end in a report, not a hang
    call = ("import sys, bench; sys.exit(bench.launch_ranks(2, ['--gpus', '2'], worker={stub!r}, device_count=2, "
            "timeout_s=3.0))").format(stub=STUB)
    t0 = time.time()
    r = _run(call, {"STUB_HANG_RANK": "1"})
    assert r.returncode == 124 and r.stdout.strip() == "", (r.returncode, r.stdout)
    assert time.time() - t0 < 45
    assert "deadline of 3 s passed with rank(s) [1] of 2 still running" in r.stderr
    assert "---- rank 1 stderr (tail) ----" in r.stderr and "entering a collective that never completes" in r.stderr


def test_launcher_rejects_a_wrong_world_size_report():
    r = _run(_CALL.format(n=2, stub=STUB, have=2), {"STUB_RCCL_RANKS": "1"})
    assert r.returncode != 0 and "rccl_ranks=1" in r.stderr


def test_cli_gpus_2_on_a_box_without_two_gpus_exits_nonzero():
    # the real entry point: on this container (and on a 1-GPU box) `bench.py --gpus 2` must not measure 1 GPU
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = dict(os.environ); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "contradicts WORLD_SIZE=2" in (r.stderr + r.stdout)
