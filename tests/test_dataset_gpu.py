"""BASELINE config 4 on its own workload (HMDB51-shaped: 320x240 clips of uneven length, dense 16-px grid of 150
size-12 keypoints per pair): harness.run_dataset = shard -> extract -> gather -> rank 0 writes ordered .mofreak text
(main.cpp:854-924).  Needs a GPU; the two-rank case shares device 0 over gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mofreak_amd as M
from mofreak_amd import harness, synth, launch

pytestmark = pytest.mark.gpu

N_CLIPS = 44


def _clips():
    c = synth.CONFIGS["C4"]
    lengths = np.minimum(synth.clip_lengths(N_CLIPS, seed=404), 140)  # seeded log-normal lengths, capped to keep the oracle quick
    pool = synth.clip_pool(4, int(lengths.max()), c["W"], c["H"])
    return [pool[i % 4][: lengths[i]] for i in range(N_CLIPS)], [f"clip{i:03d}.avi" for i in range(N_CLIPS)], lengths


def _mofreak(device=0):
    c = synth.CONFIGS["C4"]
    return harness.MoFREAKUtilities(harness.HMDB51, device=device, keypoint_provider=harness.dense_grid_provider(c["step"], c["size"], c["lo"]))


@pytest.mark.parametrize("workers", [1, 3])
def test_c4_dataset_rows_match_the_oracle_and_files_are_ordered(oracle, tmp_path, workers):
    """workers = 3: three host threads, each with a context of its own, take the clips in turn."""
    clips, names, lengths = _clips()
    assert len(synth.config_grid("C4")) == 150 and lengths.min() >= 20 and len(set(lengths.tolist())) > 20
    mo = _mofreak()
    try:
        res = harness.run_dataset(clips, names, str(tmp_path), mo, workers=workers)
    finally:
        mo.close()
    f = oracle.Freak()
    kps = synth.config_grid("C4")
    total = 0
    for i, clip in enumerate(clips):
        n_pairs = len(clip) - 5
        offs = np.arange(n_pairs + 1, dtype=np.int64) * len(kps)
        want = f.extract_stream(clip, np.tile(kps, (n_pairs, 1)), offs)
        got = res["rows_per_video"][i]
        assert got.tobytes() == want.tobytes(), f"clip {i}"
        assert np.all(np.diff(got["frame_number"]) >= 0) and got["frame_number"].min() == 4
        with open(tmp_path / (names[i] + ".mofreak"), "rb") as fh:
            assert fh.read() == oracle.format_rows(want)
        total += len(want)
    assert res["total_rows"] == total == int(((lengths - 5) * 150).sum())


def test_mixed_frame_sizes_and_bounded_rounds(oracle, tmp_path):
    """A shard whose clips have different frame sizes (HMDB51's widths vary) in rounds of a few MiB: one pipelined call per
    frame size inside a round, rows back in video order, the files of the one-round run; and out_dir with keep_rows=False
    still writes them."""
    clips, names, _ = _clips()
    clips, names = clips[:9], names[:9]
    clips[2] = np.ascontiguousarray(synth.synth_stack(19, 400, 300, t0=31))
    clips[7] = np.ascontiguousarray(synth.synth_stack(11, 352, 288, t0=90))
    mo = _mofreak()
    try:
        one = harness.run_dataset(clips, names, str(tmp_path / "one"), mo)
        rows_one = {i: r.copy() for i, r in one["rows_per_video"].items()}
        many = harness.run_dataset(clips, names, str(tmp_path / "many"), mo, batch_bytes=12 << 20)
        many["rows_per_video"] = {i: r.copy() for i, r in many["rows_per_video"].items()}  # (views into a buffer the next call reuses)
        quiet = harness.run_dataset(clips, names, str(tmp_path / "quiet"), mo, batch_bytes=12 << 20, keep_rows=False)
    finally:
        mo.close()
    assert one["rounds"] == 1 and many["rounds"] > 2 and "rows_per_video" not in quiet
    f = oracle.Freak()
    c = synth.CONFIGS["C4"]
    for i, clip in enumerate(clips):
        kps = synth.dense_grid(clip.shape[2], clip.shape[1], c["step"], c["size"], c["lo"])
        n_pairs = len(clip) - 5
        want = f.extract_stream(clip, np.tile(kps, (n_pairs, 1)), np.arange(n_pairs + 1, dtype=np.int64) * len(kps))
        assert rows_one[i].tobytes() == want.tobytes() == many["rows_per_video"][i].tobytes(), i
        text = oracle.format_rows(want)
        for d in ("one", "many", "quiet"):
            assert (tmp_path / d / (names[i] + ".mofreak")).read_bytes() == text, (d, i)
    assert quiet["total_rows"] == one["total_rows"] == sum(len(r) for r in rows_one.values())


def _free_port():
    return launch.free_port()  # (a port without TIME_WAIT leftovers of the test before)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips, names, _ = _clips()
        mo = _mofreak(0)  # both ranks on device 0: the one-GPU box's rehearsal of one-video-per-GPU
        try:
            res = harness.run_dataset(clips, names, out_dir, mo, rank=rank, world_size=world)
        finally:
            mo.close()
        assert 0 < res["videos_here"] < len(clips)
    finally:
        dist.destroy_process_group()


def test_c4_two_ranks_write_the_bytes_of_one_rank(tmp_path):
    clips, names, _ = _clips()
    one, two = tmp_path / "one", tmp_path / "two"
    mo = _mofreak()
    try:
        harness.run_dataset(clips, names, str(one), mo)
    finally:
        mo.close()
    mp.spawn(_worker, args=(2, _free_port(), str(two)), nprocs=2, join=True)
    for name in names:
        a, b = (one / (name + ".mofreak")).read_bytes(), (two / (name + ".mofreak")).read_bytes()
        assert a == b and len(a) > 0, name
    assert sorted(os.listdir(one)) == sorted(os.listdir(two))


# ------------------------------------------------------------------ one long stream split over ranks (optional)
def _stream_frames():
    c = synth.CONFIGS["C5"]
    return synth.synth_stack(31, c["W"], c["H"])


def _stream_mofreak():
    c = synth.CONFIGS["C5"]
    return harness.MoFREAKUtilities(harness.TRECVID, device=0, keypoint_provider=harness.dense_grid_provider(c["step"], c["size"], c["lo"]))


def _stream_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mo = _stream_mofreak()
        try:
            res = harness.run_stream_sharded(_stream_frames(), mo, rank, world, chunk_frames=12)
        finally:
            mo.close()
        if rank == 0:
            np.save(os.path.join(out_dir, "rows.npy"), res["rows"])
    finally:
        dist.destroy_process_group()


def test_one_stream_over_two_ranks_equals_the_whole_stream(tmp_path):
    """A 720x576 stream cut into two pieces with a 5-frame halo, one piece per rank (both on device 0 here): the gathered
    rows are those of the frame loop over the whole stream, byte for byte."""
    frames = _stream_frames()
    mo = _stream_mofreak()
    try:
        want = mo._ctx.extract_stream_host(frames, synth.config_grid("C5"))
        one = harness.run_stream_sharded(frames, mo)["rows"]
    finally:
        mo.close()
    assert one.tobytes() == want.tobytes() and len(want) == 26 * 5103
    mp.spawn(_stream_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "rows.npy").tobytes() == want.tobytes()


def test_every_rank_writes_its_own_files_with_text_made_on_the_device(oracle, tmp_path):
    """write="ranks": the round's rows stay in HBM, mofreak_format_rows_device makes the text of all the rank's videos in one call,
    the files are written from threads -- the same bytes in the same files as the root-writes-everything route, in several
    rounds of mixed frame sizes and with a clip too short to have rows (an empty file); nothing is kept on the host."""
    clips, names, _ = _clips()
    clips, names = clips[:12], names[:12]
    clips[3] = np.ascontiguousarray(synth.synth_stack(19, 400, 300, t0=31))
    clips[5] = clips[5][:4]  # no pair at all
    mo = _mofreak()
    try:
        root = harness.run_dataset(clips, names, str(tmp_path / "root"), mo, batch_bytes=12 << 20, keep_rows=False)
        ranks = harness.run_dataset(clips, names, str(tmp_path / "ranks"), mo, batch_bytes=12 << 20, keep_rows=False, write="ranks", write_threads=3)
        kept = harness.run_dataset(clips, names, str(tmp_path / "kept"), mo, write="ranks")  # rows still wanted on the host: both
        kept_rows = {i: r.copy() for i, r in kept["rows_per_video"].items()}
    finally:
        mo.close()
    assert ranks["rounds"] > 2 and ranks["write"] == "ranks" and "rows_per_video" not in ranks and ranks["total_rows"] == root["total_rows"]
    f = oracle.Freak()
    c = synth.CONFIGS["C4"]
    total = 0
    for i, clip in enumerate(clips):
        a = open(tmp_path / "root" / (names[i] + ".mofreak"), "rb").read()
        assert open(tmp_path / "ranks" / (names[i] + ".mofreak"), "rb").read() == a, names[i]
        assert open(tmp_path / "kept" / (names[i] + ".mofreak"), "rb").read() == a, names[i]
        assert M.format_rows(kept_rows[i]) == a
        total += len(a)
        if i in (0, 3):
            kps = synth.dense_grid(clip.shape[2], clip.shape[1], c["step"], c["size"], c["lo"])
            n_pairs = len(clip) - 5
            want = f.extract_stream(clip, np.tile(kps, (n_pairs, 1)), np.arange(n_pairs + 1, dtype=np.int64) * len(kps))
            assert a == oracle.format_rows(want)
    assert os.path.getsize(tmp_path / "ranks" / (names[5] + ".mofreak")) == 0
    assert ranks["text_bytes_here"] == total
    assert not [n for n in os.listdir(tmp_path / "ranks") if n.endswith(".tmp")]
