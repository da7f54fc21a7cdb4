"""The N > 1 path on CPU: video sharding and the final row gather with world_size-2 (and 3) gloo groups.

The data path has no collective (videos are independent); the only exchange is gather_rows, which is the
same code for gloo on CPU tensors and RCCL on device tensors."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mofreak_amd import api, harness, launch


def _free_port():
    return launch.free_port()  # (a port without TIME_WAIT leftovers of the test before)


def _rows_for_rank(rank: int, n: int) -> np.ndarray:
    rng = np.random.default_rng(1000 + rank)
    rows = np.zeros(n, api.ROW_DTYPE)
    rows["x"] = rng.integers(0, 1920, n)
    rows["y"] = rng.integers(0, 1080, n)
    rows["frame_number"] = np.sort(rng.integers(4, 400, n))
    rows["scale"] = 12.0
    rows["appearance"] = rng.integers(0, 256, (n, 8))
    rows["motion"] = rng.integers(0, 256, (n, 8))
    return rows


def _worker(rank, world, port, counts, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = _rows_for_rank(rank, counts[rank])
        buf = torch.zeros((counts[rank] + 5) * 32, dtype=torch.uint8)  # capacity > count, like the device buffers
        buf[: counts[rank] * 32] = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy())
        got, cnts = harness.gather_rows(buf, counts[rank], dst=0)
        assert cnts == list(counts)
        if rank == 0:
            np.save(os.path.join(outdir, "gathered.npy"), got.numpy())
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(300, 45), (0, 17), (64, 0), (5, 0, 9)])
def test_gather_rows_equals_rank_order_concatenation(tmp_path, counts):
    world = len(counts)
    mp.spawn(_worker, args=(world, _free_port(), counts, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "gathered.npy"))
    want = np.concatenate([_rows_for_rank(r, counts[r]) for r in range(world)])
    assert got.tobytes() == want.view(np.uint8).tobytes()
    # rank 0 formats text in (rank, frame, keypoint) order: identical to formatting the concatenation
    assert api.format_rows(got.view(api.ROW_DTYPE).reshape(-1)) == b"".join(
        api.format_rows(_rows_for_rank(r, counts[r])) for r in range(world))


def test_shard_videos_lpt():
    rng = np.random.default_rng(0)
    # HMDB51-shaped clip lengths: log-normal, median 80, clamped to [20, 650]
    lengths = np.clip(np.exp(rng.normal(np.log(80), 0.6, 6766)), 20, 650).astype(int)
    for world in (1, 2, 4, 8):
        shards = harness.shard_videos(lengths, world)
        allidx = sorted(i for s in shards for i in s)
        assert allidx == list(range(len(lengths)))           # a partition: every video exactly once
        loads = [int(lengths[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= lengths.max()       # LPT balance
        assert shards == harness.shard_videos(lengths, world)  # deterministic
    assert harness.shard_videos([5, 1], 4) == [[0], [1], [], []]
    assert harness.shard_videos([], 2) == [[], []]


# ------------------------------------------------------------------ run_dataset: shard -> extract -> gather -> ordered files
class _FakeMoFREAK:
    """Stands in for MoFREAKUtilities where there is no GPU: rows are a deterministic function of the clip's bytes, so
    that the sharding, the exchange and rank 0's ordering can be checked against a one-rank run on the CPU."""

    def extract_rows(self, frames):
        T = frames.shape[0]
        n = max(T - 5, 0) * 3
        rows = np.zeros(n, api.ROW_DTYPE)
        rows["frame_number"] = 4 + np.arange(n) // 3
        rows["x"] = frames[:n % 7 + 1].sum() % 1000
        rows["y"] = np.arange(n) % 3
        rows["scale"] = 12.0
        rows["appearance"][:, 0] = frames[0, 0, 0]
        rows["motion"][:, 0] = T % 256
        return rows


def _clips(n=23):
    rng = np.random.default_rng(5)
    lengths = np.clip(np.exp(rng.normal(np.log(20), 0.7, n)), 3, 90).astype(int)  # some shorter than the gap: no rows
    return [rng.integers(0, 256, (t, 4, 6), dtype=np.uint8) for t in lengths], [f"v{i:02d}.avi" for i in range(n)]


def _dataset_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips, names = _clips()
        res = harness.run_dataset(clips, names, outdir, _FakeMoFREAK(), rank=rank, world_size=world)
        assert (rank == 0) == ("rows_per_video" in res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_run_dataset_files_equal_the_one_rank_run(tmp_path, world):
    clips, names = _clips()
    one, many = tmp_path / "one", tmp_path / "many"
    res = harness.run_dataset(clips, names, str(one), _FakeMoFREAK())
    assert res["total_rows"] == sum(max(len(c) - 5, 0) * 3 for c in clips)
    mp.spawn(_dataset_worker, args=(world, _free_port(), str(many)), nprocs=world, join=True)
    assert sorted(os.listdir(one)) == sorted(os.listdir(many)) == sorted(n + ".mofreak" for n in names)
    for n in names:
        assert (one / (n + ".mofreak")).read_bytes() == (many / (n + ".mofreak")).read_bytes()
    assert any((one / (n + ".mofreak")).stat().st_size == 0 for n in names)  # clips shorter than the frame gap


def _ranks_write_worker(rank, world, port, outdir, keep_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips, names = _clips()
        res = harness.run_dataset(clips, names, outdir, _FakeMoFREAK(), rank=rank, world_size=world, write="ranks", keep_rows=keep_rows,
                                  write_threads=2, batch_bytes=1500)  # (several rounds: the text buffers' turn-taking)
        assert res["write"] == "ranks" and res["rounds"] > 1
        mine = harness.shard_videos([len(c) for c in clips], world)[rank]
        assert res["videos_here"] == len(mine)
        assert res["text_bytes_here"] == sum(len(api.format_rows(_FakeMoFREAK().extract_rows(clips[i]))) for i in mine)
        if rank == 0:  # the counts of everybody's videos reach the root either way; the rows only when they are kept
            assert res["total_rows"] == sum(max(len(c) - 5, 0) * 3 for c in clips)
            assert [int(x) for x in res["rows_per_video_counts"]] == [max(len(c) - 5, 0) * 3 for c in clips]
            assert ("rows_per_video" in res) == keep_rows
            if keep_rows:
                for i, c in enumerate(clips):
                    assert res["rows_per_video"][i].tobytes() == _FakeMoFREAK().extract_rows(c).tobytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,keep_rows", [(2, False), (3, False), (2, True)])
def test_files_written_by_their_ranks_equal_the_one_rank_run(tmp_path, world, keep_rows):
    """write="ranks": every rank formats and writes the files of its own videos (here the host formatter: no GPU); without
    keep_rows the only thing that crosses ranks is the row counts."""
    clips, names = _clips()
    one, many = tmp_path / "one", tmp_path / "many"
    harness.run_dataset(clips, names, str(one), _FakeMoFREAK())
    mp.spawn(_ranks_write_worker, args=(world, _free_port(), str(many), keep_rows), nprocs=world, join=True)
    assert sorted(os.listdir(one)) == sorted(os.listdir(many)) == sorted(n + ".mofreak" for n in names)
    for n in names:
        assert (one / (n + ".mofreak")).read_bytes() == (many / (n + ".mofreak")).read_bytes()


def test_compute_mofreak_files_skips_existing(tmp_path):
    class Rec(_FakeMoFREAK):
        def __init__(self):
            self.calls = []

        def computeMoFREAKFromFile(self, video, out, clear):
            self.calls.append(video)
            open(out, "wb").write(b"x")

    vids = []
    for i in range(4):
        p = tmp_path / f"a{i}.npy"
        np.save(p, np.zeros((3 + i, 2, 2), np.uint8))
        vids.append(str(p))
    out = tmp_path / "out"
    m = Rec()
    assert len(harness.compute_mofreak_files(vids, str(out), m)) == 4
    os.remove(out / "a2.npy.mofreak")
    m2 = Rec()
    assert harness.compute_mofreak_files(vids, str(out), m2, skip_existing=True) == [str(out / "a2.npy.mofreak")]
    assert m2.calls == [vids[2]]


def test_a_killed_writer_leaves_no_file_that_resume_would_skip(tmp_path, monkeypatch):
    """The per-video .mofreak file is the checkpoint of a resumed run, and an empty file is a legitimate result, so a
    file must never exist half written: the text goes to <name>.tmp and is renamed when complete.  A writer that dies
    mid-way leaves only the .tmp behind, and skip_existing recomputes that video."""
    target = tmp_path / "v.mofreak"
    calls = []

    def chunks():
        yield b"1 2 3 "
        calls.append("about to die")
        raise KeyboardInterrupt  # the process is killed between two chunks

    with pytest.raises(KeyboardInterrupt):
        harness.write_atomic(str(target), chunks())
    assert calls and not target.exists() and (tmp_path / "v.mofreak.tmp").exists()

    class Rec(_FakeMoFREAK):
        def __init__(self):
            self.calls = []

        def computeMoFREAKFromFile(self, video, out, clear):
            self.calls.append(video)
            harness.write_atomic(out, [b"rows\n"])

    vid = tmp_path / "v.npy"
    np.save(vid, np.zeros((9, 2, 2), np.uint8))
    out = tmp_path / "out"
    os.makedirs(out)
    (out / "v.npy.mofreak.tmp").write_bytes(b"trunc")  # what the killed run left
    m = Rec()
    assert harness.compute_mofreak_files([str(vid)], str(out), m, skip_existing=True) == [str(out / "v.npy.mofreak")]
    assert m.calls == [str(vid)] and (out / "v.npy.mofreak").read_bytes() == b"rows\n"
    assert not (out / "v.npy.mofreak.tmp").exists()
    # a complete file -- even an empty one -- is a finished video
    (out / "v.npy.mofreak").write_bytes(b"")
    m2 = Rec()
    assert harness.compute_mofreak_files([str(vid)], str(out), m2, skip_existing=True) == [] and m2.calls == []


# ------------------------------------------------------------------ one long stream over several ranks (gap-frame halo)
def test_split_stream_pieces_cover_every_processed_frame_once():
    for T, world in [(2005, 8), (90000, 8), (17, 4), (7, 8), (5, 3), (0, 2), (6, 1)]:
        pieces = harness.split_stream(T, world, 5)
        assert len(pieces) == world
        produced = []
        for f0, f1 in pieces:
            if f1 > f0:
                assert f1 - f0 > 5 and 0 <= f0 and f1 <= T  # at least one pair behind the 5-frame halo
                produced += list(range(f0 + 5, f1))           # the frames whose rows this rank produces
        assert produced == list(range(5, T)), (T, world)       # every processed frame exactly once, in rank order
        sizes = [f1 - f0 - 5 for f0, f1 in pieces if f1 > f0]
        assert not sizes or max(sizes) - min(sizes) <= 1


class _FakeCtx:
    class params:
        gap_for_frame_difference = 5

    def extract_stream_pipelined_host(self, frames, kps, chunk_frames=256):
        T = frames.shape[0]
        n = max(T - 5, 0) * len(kps)
        rows = np.zeros(n, api.ROW_DTYPE)
        rows["frame_number"] = 4 + np.arange(n) // len(kps)
        rows["x"] = np.tile(kps[:, 0], max(T - 5, 0))
        # depends on BOTH frames of the pair: a wrong halo shows
        pairs = (frames[5:].reshape(T - 5, -1).sum(1) * 3 + frames[:T - 5].reshape(T - 5, -1).sum(1)) % 251 if T > 5 else np.zeros(0)
        rows["appearance"][:, 0] = np.repeat(pairs, len(kps))
        return rows


class _FakeStreamMoFREAK:
    def __init__(self):
        self._ctx = _FakeCtx()
        self.keypoint_provider = harness.dense_grid_provider(16, 12.0, 38)


def _stream_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = np.random.default_rng(9).integers(0, 256, (43, 96, 128), dtype=np.uint8)
        res = harness.run_stream_sharded(frames, _FakeStreamMoFREAK(), rank, world)
        if rank == 0:
            np.save(os.path.join(outdir, "rows.npy"), res["rows"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stream_rows_equal_the_one_rank_stream(tmp_path, world):
    frames = np.random.default_rng(9).integers(0, 256, (43, 96, 128), dtype=np.uint8)
    want = harness.run_stream_sharded(frames, _FakeStreamMoFREAK())["rows"]
    assert len(want) == 38 * len(harness.dense_grid_provider(16, 12.0, 38)(5, 128, 96)) > 0
    mp.spawn(_stream_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "rows.npy")
    assert got.tobytes() == want.tobytes()


def test_plan_batches_bounds_a_rank_s_rounds():
    """A rank's shard is walked in consecutive batches of at most batch_bytes of frames, at least one video each."""
    nbytes = [10, 30, 5, 100, 1, 1, 50]
    assert harness.plan_batches([0, 1, 2, 3, 4, 5, 6], nbytes, 40) == [[0, 1], [2], [3], [4, 5], [6]]
    assert harness.plan_batches([6, 0], nbytes, 1 << 40) == [[6, 0]]
    assert harness.plan_batches([], nbytes, 40) == []
    assert harness.plan_batches([3], nbytes, 1) == [[3]]  # a video larger than the bound still gets its batch
    rng = np.random.default_rng(1)
    nb = rng.integers(1, 1000, 300).tolist()
    for bound in (1, 999, 5000):
        plan = harness.plan_batches(list(range(300)), nb, bound)
        assert [i for b in plan for i in b] == list(range(300))
        assert all(sum(nb[i] for i in b) <= bound or len(b) == 1 for b in plan)


def test_group_exists_follows_the_process_group():
    assert harness.group_exists() is False
