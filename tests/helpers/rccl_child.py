"""One rank of a ONE-rank RCCL process group on cuda:0, started by tests/test_rccl_gpu.py as a fresh process.

Runs the device branches of the N > 1 path -- harness.gather_rows on CUDA tensors, run_dataset(on_device=True),
run_stream_sharded(on_device=True) -- through torch.distributed's "nccl" backend (RCCL on ROCm) exactly as a rank of
an 8-GPU job does, plus a point-to-point self exchange, and leaves what it saw in <out_dir> for the parent to judge.
usage: rccl_child.py <out_dir> <port>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mofreak_amd import api, harness, synth  # noqa: E402


def main(out_dir: str, port: str) -> None:
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    report = {}

    # the reference results, before any process group exists (the plain one-rank path)
    c4 = synth.CONFIGS["C4"]
    lengths = np.minimum(synth.clip_lengths(12, seed=11), 60)
    pool = synth.clip_pool(3, int(lengths.max()), c4["W"], c4["H"])
    clips = [pool[i % 3][: lengths[i]] for i in range(len(lengths))]
    clips[5] = np.ascontiguousarray(synth.synth_stack(17, 400, 300, t0=77))  # another frame size inside the shard
    names = [f"clip{i:02d}.avi" for i in range(len(clips))]
    prov = harness.dense_grid_provider(c4["step"], c4["size"], c4["lo"])
    mo = harness.MoFREAKUtilities(harness.HMDB51, device=0, keypoint_provider=prov)
    plain = harness.run_dataset(clips, names, os.path.join(out_dir, "plain"), mo)
    assert not plain["distributed"]
    c5 = synth.CONFIGS["C5"]
    stream = synth.synth_stack(23, c5["W"], c5["H"])
    mo5 = harness.MoFREAKUtilities(harness.TRECVID, device=0, keypoint_provider=harness.dense_grid_provider(c5["step"], c5["size"], c5["lo"]))
    want_stream = mo5._ctx.extract_stream_host(stream, synth.config_grid("C5"))

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        report["backend"] = dist.get_backend()
        report["world_size"] = dist.get_world_size()
        # 1. the exchange step itself on device tensors
        rng = np.random.default_rng(5)
        rows = rng.integers(0, 256, (1000 + 7) * 32, dtype=np.uint8)
        d = torch.from_numpy(rows).cuda()
        got, counts = harness.gather_rows(d, 1000, dst=0)
        torch.cuda.synchronize()
        report["gather_is_cuda"] = bool(got.is_cuda)
        report["gather_equal"] = bool(counts == [1000] and got.cpu().numpy().tobytes() == rows[: 1000 * 32].tobytes())
        # 2. a point-to-point exchange over RCCL (peer -> root is what ranks 1..7 do): send to and receive from oneself
        src = torch.arange(1 << 16, dtype=torch.int32, device="cuda")
        dst = torch.zeros_like(src)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
            req.wait()
        torch.cuda.synchronize()
        report["p2p_equal"] = bool(torch.equal(src, dst))
        # 3. BASELINE config 4: rows stay in HBM, counts by all_reduce, gather device to device, ONE copy to the host
        res = harness.run_dataset(clips, names, os.path.join(out_dir, "grouped"), mo, on_device=True, batch_bytes=1 << 22)
        report["dataset"] = {k: res[k] for k in ("distributed", "rounds", "batched", "total_rows")}
        report["dataset_rows_equal"] = all(res["rows_per_video"][i].tobytes() == plain["rows_per_video"][i].tobytes() for i in range(len(clips)))
        # 4. one stream, rows kept in HBM up to the gather
        st = harness.run_stream_sharded(stream, mo5, on_device=True, chunk_frames=12)
        report["stream"] = {k: st[k] for k in ("distributed", "rows_in_hbm", "rows_here")}
        report["stream_rows_equal"] = bool(st["rows"].tobytes() == want_stream.tobytes())
        # what the process really mapped
        with open("/proc/self/maps") as f:
            libs = sorted({os.path.basename(line.split()[-1]) for line in f if ".so" in line and ("rccl" in line or "mofreak" in line)})
        report["mapped"] = libs
        dist.barrier()
    finally:
        dist.destroy_process_group()
        mo.close()
        mo5.close()
    with open(os.path.join(out_dir, "report.json"), "w") as f:
        json.dump(report, f)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
