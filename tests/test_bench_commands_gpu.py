"""The driver's own command shapes, end to end, on the one GPU of the box: `python bench.py` as a rank under
torch.distributed.run, and with a one-rank RCCL process group (the branch an N > 1 run takes), for the metric's workload and
for the dataset workload with files.  What a first 8-GPU run would trip over -- rendezvous, the device branch of the row gather,
the JSON line's fields -- fails here first."""
import json
import os
import subprocess
import sys

import pytest

from mofreak_amd import launch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MOFREAK_HIP_LIBRARY", None)  # (bench.py measures the in-tree product library and refuses any other)
    p = subprocess.run(cmd, cwd=ROOT, env=env, text=True, capture_output=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_metric_line_as_a_rank_of_torch_distributed_run():
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W, at N = 1: RANK / WORLD_SIZE / MASTER_* come from the launcher."""
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(launch.free_port()), "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--pairs", "32", "--no-cpu-baseline", "--no-detector"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["ranks_seen"] == 1 and d["unit"] == "descriptors/s"
    assert d["process_group"] is None  # (one rank: no group unless --force-dist, below)
    assert d["verified"]["mismatches"] == 0 and d["verified"]["descriptors"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["value"] > 1e8
    assert d["sustained"] is not None and d["sustained"]["seconds"] > 1.5  # the device is kept busy beyond the short timed region


def test_metric_line_with_a_forced_one_rank_group():
    d = _line([sys.executable, "bench.py", "--gpus", "1", "--backend", "nccl", "--force-dist", "--steps", "3", "--warmup", "1", "--pairs", "32",
               "--no-cpu-baseline", "--no-detector"])
    assert d["ranks_seen"] == 1 and d["process_group"] == {"backend": "nccl", "world_size": 1, "forced_at_n1": True}
    assert d["verified"]["mismatches"] == 0 and d["gather_ms"] is not None


@pytest.mark.parametrize("write", [False, True])
def test_dataset_line_with_a_forced_one_rank_group(write):
    cmd = [sys.executable, "bench.py", "--config", "C4", "--clips", "64", "--gpus", "1", "--backend", "nccl", "--force-dist", "--steps", "1"]
    d = _line(cmd + (["--write"] if write else []))
    assert d["unit"] == "clips/s" and d["ranks_seen"] == 1 and d["process_group"]["backend"] == "nccl" and d["value"] > 100
    if write:
        w = d["written"]
        assert w["files"] == 64 and w["files_checked_against_host_formatter"] == 3 and w["text_MB"] > 1
        assert not os.path.exists(w["directory"])  # (the bench removes its files)
    else:
        assert d["written"] is None
