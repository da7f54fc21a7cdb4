"""`python bench.py --gpus N` starts its own ranks (mofreak_amd/launch.py): N fresh processes from a parent that makes
no GPU call, rank 0's line is the job's, a failing rank fails the job instead of hanging it."""
import json
import os
import subprocess
import sys
import time

import pytest

from mofreak_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = os.path.join(ROOT, "tests", "helpers", "launch_target.py")


def _run_launcher(world, mode, timeout_s=120):
    code = ("import sys; sys.path.insert(0, %r); from mofreak_amd import launch; "
            "sys.exit(launch.self_launch(%d, %r, [%r], timeout_s=%r))" % (ROOT, world, TARGET, mode, timeout_s))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_rendezvous_and_rank0_speaks_for_the_job(world):
    r = _run_launcher(world, "ok")
    assert r.returncode == 0, r.stderr
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout  # the other ranks' stdout went to stderr
    out = json.loads(lines[0])
    assert out == {"ranks_seen": world, "sum": world * (world + 1) // 2, "launcher": "mofreak_amd.launch"}
    assert "says hello" in r.stderr


def test_a_failing_rank_fails_the_job_and_the_others_are_stopped():
    t0 = time.monotonic()
    r = _run_launcher(2, "fail")
    assert r.returncode == 3
    assert "rank 1 exited with code 3" in r.stderr
    assert time.monotonic() - t0 < 60  # rank 0 was waiting in the rendezvous: stopped, not waited for


def test_a_hung_job_is_stopped_at_the_timeout():
    r = _run_launcher(2, "hang", timeout_s=8)
    assert r.returncode == 124
    assert r.stdout.strip() == ""


def test_rank_environment():
    env = launch.rank_env(1, 4, 29999, base={})
    assert env["RANK"] == env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not launch.launched_by_a_launcher() or "RANK" in os.environ


def test_bench_parent_starts_the_ranks_before_any_gpu_or_torch_import():
    """bench.py --gpus 2 without a launcher around it: the parent must hand over to launch.self_launch straight after
    argument parsing.  No GPU here, so the ranks fail -- what matters is that they were started as ranks (their error
    names the rank), that the parent relays the failure, and that the parent itself never imported torch."""
    probe = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--backend', 'gloo', '--share-device', '--steps', '1', "
             "'--launch-timeout', '120'];\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n"
             "    print('PARENT_EXIT', e.code, 'torch' in sys.modules)\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in launch.RANK_ENV}
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, env=env)
    tail = [x for x in r.stdout.splitlines() if x.startswith("PARENT_EXIT")]
    assert tail, r.stdout + r.stderr
    _, code, torch_in_parent = tail[-1].split()
    assert torch_in_parent == "False"
    assert code != "0"  # no GPU in this container: the ranks cannot run the workload
    assert "[launch] rank" in r.stderr


def test_cpp_launcher_stops_the_other_ranks_when_one_fails(tmp_path):
    """facade_ranks (the libc-only launcher of the C++ route): a rank that ends badly, while the others would sit in a
    collective for ever, ends the job -- the others are stopped and the launcher returns the failing rank's code."""
    import stat
    import subprocess
    import time
    launcher = os.path.join(ROOT, "mofreak_amd", "host", "facade_ranks")
    if not os.path.exists(launcher):
        subprocess.check_call(["make", "-C", os.path.dirname(launcher), "-s", "facade_ranks"])
    prog = tmp_path / "rank.sh"
    marks = tmp_path / "marks"
    marks.mkdir()
    # argv: files-rank <rank> <world> <id_file> <video_dir> <mofreak_dir>
    prog.write_text("#!/bin/sh\necho $$ > %s/pid.$2\nif [ \"$2\" = \"1\" ]; then sleep 0.3; exit 3; fi\nsleep 120\n" % marks)
    prog.chmod(prog.stat().st_mode | stat.S_IEXEC)
    env = dict(os.environ, MOFREAK_RANK_PROGRAM=str(prog))
    t0 = time.monotonic()
    rc = subprocess.run([launcher, "3", str(tmp_path / "videos"), str(tmp_path / "out")], env=env, timeout=60).returncode
    assert rc == 3
    assert time.monotonic() - t0 < 20
    for r in range(3):
        pid = int((marks / f"pid.{r}").read_text())
        assert not os.path.exists(f"/proc/{pid}"), f"rank {r} is still running"
    # and the deadline: all ranks hang -> the launcher gives up and stops them
    prog.write_text("#!/bin/sh\nsleep 120\n")
    env["MOFREAK_RANKS_TIMEOUT_S"] = "1"
    t0 = time.monotonic()
    rc = subprocess.run([launcher, "2", str(tmp_path / "videos"), str(tmp_path / "out2")], env=env, timeout=60).returncode
    assert rc == 124 and time.monotonic() - t0 < 20
