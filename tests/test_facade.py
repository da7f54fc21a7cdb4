"""The C++ MoFREAKUtilities facade (mofreak_amd/host) -- the reference's class interface over the C ABI --
driven through its small command-line driver, and the Python mirror (mofreak_amd.harness)."""
import os
import subprocess

import numpy as np
import pytest

import mofreak_amd as M
from mofreak_amd import harness, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mofreak_amd", "host")
EXE = os.path.join(HOST, "facade_main")


@pytest.fixture(scope="module")
def facade(native_lib):
    from mofreak_amd import build
    build.build_dist()  # facade_main links the row gather over RCCL (include/mofreak_dist.h)
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return EXE


def _random_rows(n, seed):
    rng = np.random.default_rng(seed)
    rows = np.zeros(n, M.ROW_DTYPE)
    rows["x"] = rng.integers(0, 640, n) + rng.choice([0, 0.25, 0.5], n)
    rows["y"] = rng.integers(0, 480, n)
    rows["frame_number"] = np.sort(rng.integers(4, 90, n))
    rows["scale"] = rng.choice(np.float32([12, 8.5, 14.25, 7]), n)
    rows["appearance"] = rng.integers(0, 256, (n, 8))
    rows["motion"] = rng.integers(0, 256, (n, 8))
    return rows


def test_cpp_reader_reverses_and_writer_matches_the_c_abi_formatter(facade, tmp_path):
    """readMoFREAKFeatures pushes rows in reverse file order (reference :1206-1210); the facade's writer uses real
    ostream << float like the reference, and must give the same bytes as mofreak_format_rows."""
    rows = _random_rows(200, 1)
    src, dst = tmp_path / "in.mofreak", tmp_path / "out.mofreak"
    src.write_bytes(M.format_rows(rows))
    out = subprocess.check_output([facade, "roundtrip", str(src), str(dst)], text=True)
    assert "200 features" in out
    assert dst.read_bytes() == M.format_rows(rows[::-1])


def test_python_mirror_io_semantics(native_lib, tmp_path):
    rows = _random_rows(50, 2)
    p = tmp_path / "a.mofreak"
    p.write_bytes(M.format_rows(rows))
    mf = harness.MoFREAKUtilities.__new__(harness.MoFREAKUtilities)  # no GPU context needed for file I/O
    from collections import deque
    mf.features, mf.actions, mf.current_action = deque(), {}, 0
    mf.readMoFREAKFeatures(str(p))
    got = mf.getMoFREAKFeatures()
    assert got.tobytes() == rows[::-1].tobytes()
    mf.readMoFREAKFeatures(str(p), num_to_sample=10)
    assert len(mf.getMoFREAKFeatures()) == 60
    q = tmp_path / "b.mofreak"
    mf.clearFeatures()
    mf.readMoFREAKFeatures(str(p))
    mf.writeMoFREAKFeaturesToFile(str(q))
    assert q.read_bytes() == M.format_rows(rows[::-1])
    mf.setCurrentAction("walk")
    mf.setCurrentAction("run")
    mf.setCurrentAction("walk")
    assert mf.current_action == 0 and mf.actions == {"walk": 0, "run": 1}


@pytest.mark.gpu
def test_cpp_facade_extracts_a_clip_like_the_reference(facade, oracle, tmp_path):
    """BASELINE config 1 as it is specified: one 320x240 clip of 100 frames (95 processed frames, 204 keypoints each),
    DETECT_MOFREAK: .mofreak text identical to the oracle's."""
    c = synth.CONFIGS["C1"]
    T = 100
    fr = synth.synth_stack(T, c["W"], c["H"])
    vid, out = tmp_path / "person01_boxing_d1.npy", tmp_path / "clip.mofreak"
    np.save(vid, fr)
    msg = subprocess.check_output([facade, "extract", str(vid), str(out), str(c["step"]), str(c["size"]), str(c["lo"])], text=True)
    grid = synth.config_grid("C1")
    assert len(grid) == 204
    offs = np.arange(T - 5 + 1, dtype=np.int64) * len(grid)
    want = oracle.Freak().extract_stream(fr, np.tile(grid, (T - 5, 1)), offs)
    assert f"{len(want)} features" in msg and len(want) == (T - 5) * 204
    assert out.read_bytes() == oracle.format_rows(want)


@pytest.mark.gpu
def test_cpp_compute_mofreak_files_layout(facade, oracle, tmp_path):
    """computeMoFREAKFiles (reference main.cpp:854-924): <video> -> <MOFREAK_PATH>[/<action>]/<video>.mofreak"""
    vdir, mdir = tmp_path / "videos", tmp_path / "mofreak"
    (vdir / "walk").mkdir(parents=True)
    (vdir / "run").mkdir()
    mdir.mkdir()
    clips = {"top.npy": synth.synth_stack(8, 160, 120, t0=1), "walk/w1.npy": synth.synth_stack(7, 160, 120, t0=20),
             "run/r1.npy": synth.synth_stack(9, 160, 120, t0=50), "run/short.npy": synth.synth_stack(4, 160, 120, t0=70)}
    for name, fr in clips.items():
        np.save(vdir / name, fr)
    subprocess.check_call([facade, "files", str(vdir), str(mdir)])
    grid = synth.dense_grid(160, 120, 16, 7.0, 23)
    for name, fr in clips.items():
        out = mdir / (name + ".mofreak")
        assert out.exists(), name
        n_pairs = max(len(fr) - 5, 0)
        offs = np.arange(n_pairs + 1, dtype=np.int64) * len(grid)
        want = oracle.Freak().extract_stream(fr, np.tile(grid, (n_pairs, 1)), offs)
        assert out.read_bytes() == oracle.format_rows(want), name
    assert (mdir / "run" / "short.npy.mofreak").read_bytes() == b""  # T <= gap: the reference writes an empty file


@pytest.mark.gpu
def test_cpp_ranks_over_rccl_write_the_files_of_the_plain_walk(facade, tmp_path):
    """facade_ranks 1: the C++ route to the N-GPU run -- a launcher that loads no GPU library starts the rank process(es);
    a rank = MoFREAKUtilities::computeMoFREAKFromFilesSharded over include/mofreak_dist.h: LPT shard, rows left in HBM,
    ncclAllReduce of the per-video counts, ncclAllGather of the ranks' counts, grouped ncclSend / ncclRecv to rank 0 (at one
    rank: a self exchange of 1 MiB in their place), rank 0 writes.  Rounds of a few hundred KiB; mixed frame sizes; an empty
    clip.  The files are those of `facade_main files`, byte for byte."""
    vdir, one, many = tmp_path / "videos", tmp_path / "one", tmp_path / "ranks"
    (vdir / "walk").mkdir(parents=True)
    (vdir / "run").mkdir()
    clips = {"top.npy": synth.synth_stack(8, 160, 120, t0=1), "walk/w1.npy": synth.synth_stack(17, 160, 120, t0=20), "walk/w2.npy": synth.synth_stack(9, 208, 144, t0=5),
             "run/r1.npy": synth.synth_stack(29, 160, 120, t0=50), "run/short.npy": synth.synth_stack(4, 160, 120, t0=70), "run/r2.npy": synth.synth_stack(12, 160, 120, t0=90)}
    for name, fr in clips.items():
        np.save(vdir / name, fr)
    one.mkdir()
    subprocess.check_call([facade, "files", str(vdir), str(one)])
    launcher = os.path.join(os.path.dirname(facade), "facade_ranks")
    # both ways to the files: every rank writes its own videos' files from text made on the device (the default), and rows
    # gathered to rank 0 over RCCL with rank 0 writing
    for gather, out in (("0", many), ("1", tmp_path / "gathered")):
        env = dict(os.environ, MOFREAK_BATCH_BYTES=str(400_000), HSA_ENABLE_IPC_MODE_LEGACY="0", MOFREAK_GATHER_TO_ROOT=gather)
        msg = subprocess.run([launcher, "1", str(vdir), str(out)], env=env, text=True, capture_output=True, timeout=300)
        assert msg.returncode == 0, msg.stdout[-2000:] + msg.stderr[-2000:]
        assert "ranks 1 videos 6" in msg.stdout
        for name in clips:
            a, b = (one / (name + ".mofreak")).read_bytes(), (out / (name + ".mofreak")).read_bytes()
            assert a == b and (len(a) > 0 or name.endswith("short.npy")), (gather, name)
        assert not [f for f in os.listdir(out) if f.startswith(".mofreak_rccl_id")]
        assert not [f for d, _, fs in os.walk(out) for f in fs if f.endswith(".tmp")]


def test_dist_library_exports_its_header_and_the_exchange_logic_runs_on_the_cpu():
    """include/mofreak_dist.h: every declared function is exported by libmofreak_dist.so (looked up with nm: the library links
    /opt/rocm's RCCL and HIP runtime and is not loaded into this Python process), and csrc/dist_gather.h -- the LPT shard and
    the peer-to-root gather that run over RCCL on the GPUs -- passes its self-test over an in-process communicator."""
    import re
    from mofreak_amd import build
    lib = build.build_dist()
    declared = set(re.findall(r"\b(mofreak_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "mofreak_dist.h")).read()))
    declared -= {"mofreak_amd"}
    exported = {line.split()[-1] for line in subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True).splitlines() if " T " in line}
    assert declared and declared <= exported, sorted(declared - exported)
    host = os.path.join(ROOT, "mofreak_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s", "dist_selftest"])
    assert subprocess.check_output([os.path.join(host, "dist_selftest")], text=True).strip() == "ok"


@pytest.mark.gpu
def test_python_mirror_extracts_and_shards(native_lib, oracle, tmp_path):
    vdir = tmp_path / "v"
    vdir.mkdir()
    paths = []
    for i, T in enumerate([12, 7, 9, 6]):
        p = vdir / f"clip{i}.npy"
        np.save(p, synth.synth_stack(T, 160, 120, t0=10 * i))
        paths.append(str(p))
    prov = harness.dense_grid_provider(16, 7.0, 23)
    mf = harness.MoFREAKUtilities(harness.KTH, device=0, keypoint_provider=prov)
    done = []
    for rank in range(2):  # two logical ranks on one device: together they cover every video exactly once
        done += harness.compute_mofreak_files(paths, str(tmp_path / "out"), mf, rank=rank, world_size=2,
                                              costs=[12, 7, 9, 6])
    assert sorted(os.path.basename(d) for d in done) == [f"clip{i}.npy.mofreak" for i in range(4)]
    grid = synth.dense_grid(160, 120, 16, 7.0, 23)
    for i, p in enumerate(paths):
        fr = np.load(p)
        n_pairs = len(fr) - 5
        want = oracle.Freak().extract_stream(fr, np.tile(grid, (n_pairs, 1)), np.arange(n_pairs + 1, dtype=np.int64) * len(grid))
        assert open(os.path.join(tmp_path, "out", f"clip{i}.npy.mofreak"), "rb").read() == oracle.format_rows(want)
    # append semantics: without clearing, the second file also carries the first video's rows (reference :479, :493-496)
    a, b = str(tmp_path / "a.mofreak"), str(tmp_path / "b.mofreak")
    mf.computeMoFREAKFromFile(paths[1], a, False)
    mf.computeMoFREAKFromFile(paths[3], b, True)
    assert open(b, "rb").read().startswith(open(a, "rb").read()) and len(open(b, "rb").read()) > len(open(a, "rb").read())
    one = mf.buildMoFREAKFeature(np.load(paths[0])[5], np.load(paths[0])[0], 80, 64, 7.0)
    d, v = oracle.Freak().extract_pair(np.load(paths[0])[5], np.load(paths[0])[0], np.float32([[80, 64, 7.0]]))
    assert v[0] == 1 and np.array_equal(one[0], d[0, :8]) and np.array_equal(one[1], d[0, 8:])
    mf.close()


@pytest.mark.gpu
def test_long_stream_in_chunks_equals_one_pass(native_lib, oracle):
    """BASELINE config 5 shape in miniature: a stream walked in overlapping chunks (5-frame halo) gives the rows of
    one pass -- per-frame keypoint lists, so the provider's frame index has to survive the chunking too."""
    W, H, T = 176, 144, 41
    fr = synth.synth_stack(T, W, H, t0=3)
    rng = np.random.default_rng(9)
    per_frame = {t: synth.random_keypoints(rng, int(rng.integers(0, 60)), W, H, sizes=(7.0, 9.0, 12.0)) for t in range(T)}
    mf = harness.MoFREAKUtilities(harness.TRECVID, device=0, keypoint_provider=lambda t, w, h: per_frame[t])
    whole = mf.extract_rows(fr)
    for chunk in (6, 7, 13, 40):
        assert mf.extract_rows(fr, chunk_frames=chunk).tobytes() == whole.tobytes(), chunk
    lists = [per_frame[t] for t in range(5, T)]
    offs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
    want = oracle.Freak().extract_stream(fr, np.concatenate(lists), offs)
    assert whole.tobytes() == want.tobytes() and len(whole) > 200
    mf.close()


def test_chop_merge_and_binary_sidecar(native_lib, tmp_path):
    """The reference's file utilities (src/merge_mofreak_files.py): 40k-line chunks numbered from 0; merge in id order from 1."""
    rows = _random_rows(25, 3)
    text = M.format_rows(rows)
    src = tmp_path / "LGW_20071101_E1_CAM1.mpeg.mofreak"
    src.write_bytes(text)
    assert harness.chop_mofreak_file(str(src), str(tmp_path / "chop")) == []  # short files are left alone
    pieces = harness.chop_mofreak_file(str(src), str(tmp_path / "chop"), lines_per_file=10)
    assert [os.path.basename(p) for p in pieces] == [f"LGW_20071101_E1_CAM1.mpeg.{k}.mofreak" for k in range(3)]
    assert b"".join(open(p, "rb").read() for p in pieces) == text
    merged = harness.merge_mofreak_files(str(tmp_path / "chop"), str(tmp_path / "merged"))
    assert [os.path.basename(m) for m in merged] == ["LGW_20071101_E1_CAM1.mpeg.TestSequence.mofreak"]
    lines = text.splitlines(keepends=True)
    assert open(merged[0], "rb").read() == b"".join(lines[10:])  # ids 1 and 2: the script's merge starts at 1, piece 0 is skipped
    assert open(harness.merge_mofreak_files(str(tmp_path / "chop"), str(tmp_path / "m0"), first_id=0)[0], "rb").read() == text
    side = tmp_path / "rows.npy"
    harness.write_rows_binary(str(side), rows)
    assert harness.read_rows_binary(str(side)).tobytes() == rows.tobytes()
    assert M.parse_rows(text).tobytes() == rows.tobytes()


@pytest.mark.gpu
def test_trecvid_shaped_stream_in_chunks(native_lib, oracle):
    """BASELINE config 5 shape: 720x576 frames, dense 8-px grid, walked in 10-frame chunks with the 5-frame halo."""
    c = synth.CONFIGS["C5"]
    W, H, T = c["W"], c["H"], 23
    fr = synth.synth_stack(T, W, H, t0=100)
    grid = synth.config_grid("C5")
    mf = harness.MoFREAKUtilities(harness.TRECVID, device=0,
                                  keypoint_provider=harness.dense_grid_provider(c["step"], c["size"], c["lo"]))
    rows = mf.extract_rows(fr, chunk_frames=10)
    assert rows.tobytes() == mf.extract_rows(fr).tobytes()
    offs = np.arange(T - 5 + 1, dtype=np.int64) * len(grid)
    want = oracle.Freak().extract_stream(fr, np.tile(grid, (T - 5, 1)), offs)
    assert rows.tobytes() == want.tobytes() and len(rows) == (T - 5) * len(grid) > 50000
    assert list(np.unique(rows["frame_number"])) == list(range(4, T - 1))
    mf.close()


@pytest.mark.gpu
def test_cpp_dataset_routes_with_the_brisk_detector(facade, tmp_path):
    """useBriskDetector() in the batched walk (`facade_main files`) and in the N-GPU route (`facade_ranks 1`, both ways to the
    files): the detector runs window by window inside mofreak_compute_clips; the files are those of one
    computeMoFREAKFromFile call per video (`facade_main extract <video> <out> brisk`)."""
    vdir = tmp_path / "videos"
    (vdir / "walk").mkdir(parents=True)
    clips = {"walk/a.npy": synth.moving_objects_stack(12, 256, 192, seed=71), "walk/b.npy": synth.moving_objects_stack(27, 256, 192, seed=72),
             "walk/c.npy": synth.moving_objects_stack(5, 256, 192, seed=73), "walk/d.npy": synth.moving_objects_stack(9, 208, 144, seed=74)}
    for name, fr in clips.items():
        np.save(vdir / name, fr)
    single = tmp_path / "single"
    (single / "walk").mkdir(parents=True)
    for name in clips:
        subprocess.check_call([facade, "extract", str(vdir / name), str(single / (name + ".mofreak")), "brisk"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MOFREAK_USE_BRISK="1", MOFREAK_BATCH_BYTES=str(900_000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = []
    batched = tmp_path / "batched"
    batched.mkdir()
    subprocess.check_call([facade, "files", str(vdir), str(batched)], env=env, stdout=subprocess.DEVNULL)
    outs.append(batched)
    launcher = os.path.join(os.path.dirname(facade), "facade_ranks")
    for gather in ("0", "1"):
        out = tmp_path / f"ranks{gather}"
        msg = subprocess.run([launcher, "1", str(vdir), str(out)], env=dict(env, MOFREAK_GATHER_TO_ROOT=gather), text=True, capture_output=True, timeout=300)
        assert msg.returncode == 0, msg.stdout[-2000:] + msg.stderr[-2000:]
        outs.append(out)
    sizes = 0
    for name in clips:
        want = (single / (name + ".mofreak")).read_bytes()
        sizes += len(want)
        for out in outs:
            assert (out / (name + ".mofreak")).read_bytes() == want, (str(out), name)
    assert sizes > 10000 and (single / "walk/c.npy.mofreak").read_bytes() == b""
