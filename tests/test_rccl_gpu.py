"""The device branch of the N > 1 path on the one-GPU box: a ONE-rank process group of torch.distributed's "nccl" backend
(RCCL on ROCm) in a fresh child process, through harness.gather_rows / run_dataset(on_device=True) /
run_stream_sharded(on_device=True) -- the code a rank of the 8-GPU job runs (main.cpp:854-924 sharded).
Two ranks cannot share one GPU under RCCL; world size 1 runs the same collectives and the same tensors."""
import json
import os
import signal
import subprocess
import sys

import pytest

from mofreak_amd import launch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    return launch.free_port()  # (a port without TIME_WAIT leftovers of the test before)


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    out = tmp_path_factory.mktemp("rccl")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_child.py"), str(out), str(_free_port())],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        log, _ = proc.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # the child is its own process group: by pid, never by pattern
        log, _ = proc.communicate()
        pytest.fail("the RCCL child did not finish in 420 s:\n" + log.decode(errors="replace")[-4000:])
    assert proc.returncode == 0, log.decode(errors="replace")[-4000:]
    with open(out / "report.json") as f:
        rep = json.load(f)
    rep["_dir"] = str(out)
    return rep


def test_the_group_is_rccl_and_the_library_is_mapped(report):
    assert report["backend"] == "nccl" and report["world_size"] == 1
    assert any("rccl" in m for m in report["mapped"]), report["mapped"]
    product = os.path.basename(os.environ.get("MOFREAK_HIP_LIBRARY", "libmofreak_hip.so"))  # (the suite also runs on the bounds-checking build)
    assert any(product == m for m in report["mapped"]), report["mapped"]


def test_gather_rows_on_device_tensors(report):
    assert report["gather_is_cuda"] and report["gather_equal"]


def test_point_to_point_over_rccl(report):
    assert report["p2p_equal"]


def test_run_dataset_device_branch_writes_the_bytes_of_the_plain_run(report):
    d = report["dataset"]
    assert d["distributed"] and d["batched"] and d["rounds"] > 1 and d["total_rows"] > 0
    assert report["dataset_rows_equal"]
    one, two = os.path.join(report["_dir"], "plain"), os.path.join(report["_dir"], "grouped")
    assert sorted(os.listdir(one)) == sorted(os.listdir(two)) and len(os.listdir(one)) == 12
    for name in os.listdir(one):
        with open(os.path.join(one, name), "rb") as a, open(os.path.join(two, name), "rb") as b:
            assert a.read() == b.read(), name


def test_run_stream_sharded_keeps_its_rows_in_hbm(report):
    s = report["stream"]
    assert s["distributed"] and s["rows_in_hbm"] and s["rows_here"] == 18 * 5103
    assert report["stream_rows_equal"]
