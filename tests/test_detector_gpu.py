"""Parity of the HIP keypoint detector (mofreak_amd/csrc/detect_kernel.hip) with the CPU oracle
(oracle/brisk_oracle.c): pyramid bytes, dense corner scores, and the keypoint list -- coordinates, size, response,
layer AND order, all bit-exact -- through the C ABI."""
import os
import sys

import numpy as np
import pytest

import mofreak_amd as M
import oracle_lib as O
from mofreak_amd import synth
from test_brisk_oracle import C16, _score_maxmin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(native_lib):
    c = M.Context(0)
    yield c
    c.close()


def _oracle_keypoints(img, threshold=30, octaves=3):
    k = O.brisk_detect(img, threshold, octaves)
    return (np.stack([k["x"], k["y"], k["size"]], 1).astype(np.float32).reshape(-1, 3), k["response"].copy(), k["layer"].copy())


def _assert_same_keypoints(got, want, what=""):
    gk, gr, gl = got
    wk, wr, wl = want
    assert len(gk) == len(wk), f"{what}: {len(gk)} keypoints, oracle {len(wk)}"
    assert np.array_equal(gl, wl), what
    assert gk.tobytes() == wk.tobytes(), f"{what}: first difference at {np.nonzero((gk != wk).any(1))[0][:5]}"
    assert gr.tobytes() == wr.tobytes(), what


def _quantised_noise(seed, h, w, levels=4, block=3):
    """Blocky few-level noise: plenty of equal corner scores next to each other, i.e. isMax2D ties."""
    rng = np.random.default_rng(seed)
    small = rng.integers(0, levels, ((h + block - 1) // block, (w + block - 1) // block)) * (255 // (levels - 1))
    return np.kron(small, np.ones((block, block), np.int64))[:h, :w].astype(np.uint8)


@pytest.mark.parametrize("w,h", [(320, 240), (212, 160), (100, 70), (53, 47), (640, 360), (1920, 1080)])
def test_pyramid_and_scores_match_the_oracle(ctx, w, h):
    img = np.random.default_rng(w + h).integers(0, 256, (h, w), dtype=np.uint8)
    img[h // 4:h // 2, w // 4:w // 2] = 230
    got = ctx.brisk_pyramid_host(img, 3)
    b = O.Brisk(img, 3)
    assert len(got) == b.n_layers == 6
    for i, (layer, score, scale, offset) in enumerate(got):
        lw, lh, ls, lo = b.layer_info(i)
        assert layer.shape == (lh, lw) and scale == ls and offset == lo
        assert np.array_equal(layer, b.layer_image(i)), f"layer {i}"
        if lw >= 7 and lh >= 7:
            assert np.array_equal(score, _score_maxmin(layer, C16, 9).astype(np.uint8)), f"scores of layer {i}"
    # spot-check the dense scores against the oracle's bisection (what the reference computes lazily)
    layer, score = got[0][0], got[0][1]
    rng = np.random.default_rng(1)
    for _ in range(200):
        x, y = int(rng.integers(3, w - 3)), int(rng.integers(3, h - 3))
        assert score[y, x] == O.oast_score(layer, x, y, 0)


def test_keypoints_of_moving_objects_match_the_oracle(ctx):
    fr = synth.moving_objects_stack(9, 320, 240)
    cur, prev = fr[5:], fr[:4]
    kps, offs, resp, layer = ctx.detect_pairs_host(cur, prev)
    assert offs[0] == 0 and offs[-1] == len(kps) and len(kps) > 400
    for p in range(4):
        want = _oracle_keypoints(O.absdiff(cur[p], prev[p]))
        s = slice(offs[p], offs[p + 1])
        _assert_same_keypoints((kps[s], resp[s], layer[s]), want, f"pair {p}")
        assert set(np.unique(want[2])) == set(range(6))


@pytest.mark.parametrize("octaves", [0, 1, 2, 3, 4])
def test_every_pyramid_depth(ctx, octaves):
    fr = synth.moving_objects_stack(6, 400, 300, seed=3)
    d = O.absdiff(fr[5], fr[0])
    kps, offs, resp, layer = ctx.detect_pairs_host(d, None, 30, octaves)
    want = _oracle_keypoints(d, 30, octaves)
    _assert_same_keypoints((kps, resp, layer), want, f"octaves {octaves}")
    assert len(kps) > 50


@pytest.mark.parametrize("seed,levels,block,thr", [(1, 4, 3, 30), (2, 3, 2, 30), (3, 2, 4, 20), (4, 6, 5, 40), (5, 4, 1, 30)])
def test_ties_are_broken_like_the_sequential_reference(ctx, seed, levels, block, thr):
    """Equal neighbouring scores: the outcome depends on which score-cache cells the reference has filled by the time
    it reaches a candidate.  The oracle replays that order; the device reproduces it from its touch/status maps."""
    img = _quantised_noise(seed, 180, 240, levels, block)
    b = O.Brisk(img, 3)
    want_all = b.get_keypoints(thr)
    # make sure this input really exercises ties on several layers
    ties = 0
    for i in range(b.n_layers):
        sc = b.layer_scores(i).astype(np.int32)
        pts = b.layer_points(i)
        for x, y in pts:
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
            c = nb[1, 1]
            nb[1, 1] = -1
            ties += int((nb.max() == c))
    assert ties > 20
    kps, offs, resp, layer = ctx.detect_pairs_host(img, None, thr, 3)
    _assert_same_keypoints((kps, resp, layer), _oracle_keypoints(img, thr, 3), f"seed {seed}")
    assert len(want_all) == len(kps)


def test_device_equals_the_independent_sequential_restatement(ctx):
    """The device against tests/golden/brisk_sequential.npz: keypoints of the pure-Python, one-candidate-at-a-time
    restatement written from brisk.cpp alone (tests/helpers/brisk_sequential.py; the oracle is not involved here).
    Tie-heavy inputs, a moving-object difference image at every pyramid depth."""
    import test_brisk_sequential as T
    z = np.load(T.GOLDEN)
    n = 0
    for model, code in (("x87", M.api.FP_X87), ("sse", M.api.FP_SSE)):  # both readings of a float expression
        with M.Context(0, brisk_fp_model=code) as c:
            for name, img, thr, octaves in T.cases():
                kps, offs, resp, layer = c.detect_pairs_host(img, None, thr, octaves)
                w = z[f"{model}/{name}"]
                want = (np.stack([w["x"], w["y"], w["size"]], 1).astype(np.float32).reshape(-1, 3), w["response"].copy(), w["layer"].copy())
                _assert_same_keypoints((kps, resp, layer), want, f"{model}/{name}")
                n += len(w)
    assert n > 40000


@pytest.mark.parametrize("T,W,H", [(21, 320, 240), (25, 400, 300), (60, 320, 240), (72, 212, 160)])
def test_frame_loop_rows_equal_the_three_calls_and_the_oracle(ctx, T, W, H):
    """mofreak_compute_stream = detector -> descriptors -> rows in one call: byte-identical to the three separate calls and to
    the oracle's frame loop (detector -> FREAK + MIP -> rows)."""
    fr = synth.moving_objects_stack(T, W, H, seed=T)
    got = ctx.compute_stream_host(fr)
    kps_d, offs_d, _, _ = ctx.detect_pairs_host(fr[5:], fr[:-5])
    assert got.tobytes() == ctx.extract_stream_host(fr, kps_d, kp_offsets=offs_d).tobytes() and len(got) > 200
    lists = [O.brisk_detect(O.absdiff(fr[t], fr[t - 5])) for t in range(5, T)]
    kps = np.concatenate([np.stack([k["x"], k["y"], k["size"]], 1) for k in lists]).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
    want = O.Freak().extract_stream(fr, kps, offs)
    assert got.tobytes() == want.tobytes()
    assert got["frame_number"].min() == 4 and got["frame_number"].max() == T - 2
    assert ctx.compute_stream_host(fr).tobytes() == want.tobytes()  # again on the same context: its buffers are reused


def test_frame_loop_grows_its_keypoint_buffers(ctx):
    """More keypoints per frame than the loop's keypoint buffer was made for (4096 per pair): reported inside the call, the
    buffer grown once to what the stack needs, same rows as the separate calls."""
    rng = np.random.default_rng(3)
    T, W, H = 22, 640, 480
    fr = rng.integers(0, 2, (T, H // 4, W // 4)).astype(np.uint8) * 200
    fr = np.kron(fr, np.ones((1, 4, 4), np.uint8))  # blocky binary noise: corners everywhere
    ctx.set_detect_capacity(1 << 20)
    try:
        got = ctx.compute_stream_host(fr, capacity=T * 40000)
        kps_d, offs_d, _, _ = ctx.detect_pairs_host(fr[5:], fr[:-5], capacity=T * 40000)
        plain = ctx.extract_stream_host(fr, kps_d, kp_offsets=offs_d)
    finally:
        ctx.set_detect_capacity(0)  # (the default: room by frame size)
    assert len(got) > 17 * 8192 and got.tobytes() == plain.tobytes()


def test_frame_loop_in_several_detector_batches(ctx):
    """mofreak_compute_stream walks a stack in detector batches (as many pairs as the detector's workspace takes at once);
    with 4 M candidates reserved per pair a batch is a dozen pairs: four batches here, frame numbers and rows running on,
    each batch's integral images read from the difference planes the detector has just left.  Same rows as the three calls."""
    T, W, H = 40, 160, 120
    fr = synth.moving_objects_stack(T, W, H, seed=9)
    kps_d, offs_d, _, _ = ctx.detect_pairs_host(fr[5:], fr[:-5])
    plain = ctx.extract_stream_host(fr, kps_d, kp_offsets=offs_d)
    ctx.set_detect_capacity(1 << 22)
    try:
        got = ctx.compute_stream_host(fr)
        with pytest.raises(M.MoFREAKError) as e:  # a rows buffer that is too small: the size a retry needs comes back
            n_rows, n_kp = ctx.compute_stream(fr, T, W, H, np.zeros(10, M.api.ROW_DTYPE), capacity=10)
        assert e.value.code == M.api.ERR_CAPACITY and str(len(plain)) in str(e.value)
    finally:
        ctx.set_detect_capacity(0)  # (the default: room by frame size)
    assert len(got) > 200 and got.tobytes() == plain.tobytes()
    assert len(set(got["frame_number"].tolist())) > 20  # rows from every batch


def test_rows_wider_than_one_chunk(ctx):
    """Layer rows longer than the 2048 pixels the candidate kernel takes per chunk (and than one pyramid block row):
    keypoints next to the chunk boundary see their neighbours across it."""
    rng = np.random.default_rng(11)
    img = _quantised_noise(3, 96, 4400, levels=5, block=2)
    img[:, 2040:2060] = rng.integers(0, 255, (96, 20), dtype=np.uint8)   # busy columns around x = 2048
    img[:, 4090:4104] = rng.integers(0, 255, (96, 14), dtype=np.uint8)   # and around the half-resolution layer's (x = 4096)
    ctx.set_detect_capacity(1 << 20)  # this much noise has more corner candidates than the default reserve
    try:
        kps, offs, resp, layer = ctx.detect_pairs_host(img, None, 30, 2)
    finally:
        ctx.set_detect_capacity(0)  # (the default: room by frame size)
    _assert_same_keypoints((kps, resp, layer), _oracle_keypoints(img, 30, 2), "4400-wide")
    near = np.abs(kps[:, 0] - 2048) < 6
    assert near.any(), "no keypoint next to the chunk boundary: the input does not test it"


def test_full_hd_pair_and_capacity_errors(ctx):
    fr = synth.moving_objects_stack(6, 1920, 1080)
    d = O.absdiff(fr[5], fr[0])
    kps, offs, resp, layer = ctx.detect_pairs_host(fr[5], fr[0])
    _assert_same_keypoints((kps, resp, layer), _oracle_keypoints(d), "1080p")
    assert len(kps) > 3000
    with pytest.raises(M.MoFREAKError) as e:
        ctx.detect_pairs_host(fr[5], fr[0], capacity=100)
    assert e.value.code == M.api.ERR_CAPACITY
    ctx.set_detect_capacity(256)
    with pytest.raises(M.MoFREAKError) as e:
        ctx.detect_pairs_host(fr[5], fr[0])
    assert e.value.code == M.api.ERR_CAPACITY
    ctx.set_detect_capacity(0)  # (the default: room by frame size)
    again = ctx.detect_pairs_host(fr[5], fr[0])
    assert again[0].tobytes() == kps.tobytes()


def test_candidate_room_follows_the_frames_and_grows():
    """The default: room for an eighth of the frame's pixels as candidates per pair (at least 4096), grown four-fold by a call that
    meets more.  Blocky few-level noise on a small frame has several times that: the call runs again by itself and gives the
    oracle's keypoints -- the first time, the second time (the room is kept), and for a quiet image afterwards; and a number named
    by the caller is the room outright: too small a one is an error, not a retry."""
    noisy = _quantised_noise(5, 150, 200, levels=3, block=2)
    quiet = O.absdiff(*synth.moving_objects_stack(6, 200, 150, seed=4)[[5, 0]])
    want_noisy, want_quiet = _oracle_keypoints(noisy), _oracle_keypoints(quiet)
    b = O.Brisk(noisy, 3)
    b.get_keypoints(30)
    assert sum(len(b.layer_points(l)) for l in range(b.n_layers)) > 2 * 4096  # (the input does ask for more room than the start)
    with M.Context(0) as fresh:
        for turn in range(2):
            kps, offs, resp, layer = fresh.detect_pairs_host(noisy, None)
            _assert_same_keypoints((kps, resp, layer), want_noisy, f"noisy, turn {turn}")
        kps, offs, resp, layer = fresh.detect_pairs_host(quiet, None)
        _assert_same_keypoints((kps, resp, layer), want_quiet, "quiet")
        fresh.set_detect_capacity(512)
        with pytest.raises(M.MoFREAKError) as e:
            fresh.detect_pairs_host(noisy, None)
        assert e.value.code == M.api.ERR_CAPACITY
        fresh.set_detect_capacity(0)
        kps, offs, resp, layer = fresh.detect_pairs_host(noisy, None)
        _assert_same_keypoints((kps, resp, layer), want_noisy, "noisy, automatic again")


def test_detect_then_describe_gives_the_reference_rows(ctx, oracle):
    """The two halves of computeMoFREAKFromFile's loop body together: detector output feeds the descriptor path through
    CSR offsets; rows equal the oracle's for the oracle's own keypoints."""
    fr = synth.moving_objects_stack(11, 352, 288, seed=11)
    T = len(fr)
    cur, prev = fr[5:], fr[:T - 5]
    kps, offs, _, _ = ctx.detect_pairs_host(cur, prev)
    rows = ctx.extract_stream_host(fr, kps, kp_offsets=offs)
    lists = [_oracle_keypoints(O.absdiff(cur[p], prev[p]))[0] for p in range(T - 5)]
    woffs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
    assert np.array_equal(offs, woffs)
    want = oracle.Freak().extract_stream(fr, np.concatenate(lists), woffs)
    assert rows.tobytes() == want.tobytes() and len(rows) > 300
    # keypoints near the border are detected but erased by FREAK's border rule: fewer rows than keypoints
    assert len(rows) < len(kps)


def test_empty_and_flat_inputs(ctx):
    flat = np.full((2, 120, 160), 77, np.uint8)
    kps, offs, _, _ = ctx.detect_pairs_host(flat, flat)
    assert len(kps) == 0 and np.array_equal(offs, [0, 0, 0])
    tiny = np.random.default_rng(0).integers(0, 256, (1, 20, 24), dtype=np.uint8)
    kps, offs, resp, layer = ctx.detect_pairs_host(tiny, None)
    _assert_same_keypoints((kps, resp, layer), _oracle_keypoints(tiny[0]), "tiny")


def test_compute_stream_is_the_whole_frame_loop(ctx, oracle, tmp_path):
    """mofreak_compute_stream == detector + descriptors + compaction: the rows of the .mofreak file the reference
    would write for the clip, through the C ABI, the Python mirror and the C++ facade."""
    import os
    import subprocess
    from mofreak_amd import harness
    fr = synth.moving_objects_stack(12, 320, 240, seed=5)
    T = len(fr)
    lists = [_oracle_keypoints(O.absdiff(fr[t], fr[t - 5]))[0] for t in range(5, T)]
    offs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
    want = oracle.Freak().extract_stream(fr, np.concatenate(lists), offs)
    rows = ctx.compute_stream_host(fr)
    assert rows.tobytes() == want.tobytes() and len(rows) > 100
    assert len(ctx.compute_stream_host(fr[:5])) == 0 and len(ctx.compute_stream_host(fr, capacity=16)) == len(rows)
    mf = harness.MoFREAKUtilities(harness.KTH, device=0, keypoint_provider="brisk")
    assert mf.extract_rows(fr).tobytes() == want.tobytes()
    assert mf.extract_rows(fr, chunk_frames=8).tobytes() == want.tobytes()
    mf.close()
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mofreak_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    vid, out = tmp_path / "person01_walking_d1.npy", tmp_path / "clip.mofreak"
    np.save(vid, fr)
    subprocess.check_call([os.path.join(host, "facade_main"), "extract", str(vid), str(out), "brisk"])
    assert out.read_bytes() == oracle.format_rows(want)


def test_many_pairs_in_several_batches(ctx):
    """The detector works through a call in batches sized from its workspace; a large candidate reservation forces one
    pair per batch here, and the CSR offsets / running totals have to carry across batches."""
    fr = synth.moving_objects_stack(10, 240, 176, seed=21)
    cur, prev = fr[5:], fr[:5]
    one = ctx.detect_pairs_host(cur, prev)
    ctx.set_detect_capacity(1 << 24)
    try:
        many = ctx.detect_pairs_host(cur, prev)
    finally:
        ctx.set_detect_capacity(0)  # (the default: room by frame size)
    for a, b in zip(one, many):
        assert a.tobytes() == b.tobytes()
    assert len(one[1]) == 6 and np.all(np.diff(one[1]) > 0)
    for p in range(5):
        want = _oracle_keypoints(O.absdiff(cur[p], prev[p]))
        s = slice(one[1][p], one[1][p + 1])
        _assert_same_keypoints((one[0][s], one[2][s], one[3][s]), want, f"pair {p}")


def test_device_buffers_strides_and_thresholds(ctx):
    """Device-resident frames with padded rows and a pair stride that skips frames; other thresholds than the reference's 30."""
    import torch
    W, H, n = 200, 150, 3
    fr = synth.moving_objects_stack(5 + 2 * n, W, H, seed=31)
    stride, fstride = 224, 224 * 160  # padded rows, padded frames
    buf = torch.zeros((5 + 2 * n, 160, stride), dtype=torch.uint8)
    buf[:, :H, :W] = torch.from_numpy(fr)
    d = buf.cuda()
    cap = 1 << 16
    kps = torch.zeros((cap, 3), dtype=torch.float32, device="cuda")
    offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    resp = torch.zeros(cap, dtype=torch.float32, device="cuda")
    layer = torch.zeros(cap, dtype=torch.int32, device="cuda")
    for thr in (30, 12, 70):
        torch.cuda.synchronize()
        # pairs (5, 0), (7, 2), (9, 4): every other frame
        total = ctx.detect_pairs(d[5:], d, W, H, n, kps, offs, threshold=thr, octaves=3, out_response=resp, out_layer=layer,
                                 capacity=cap, row_stride=stride, pair_stride=2 * fstride)
        o = offs.cpu().numpy()
        assert o[-1] == total
        for p in range(n):
            want = _oracle_keypoints(O.absdiff(fr[5 + 2 * p], fr[2 * p]), thr, 3)
            s = slice(o[p], o[p + 1])
            _assert_same_keypoints((kps[s].cpu().numpy(), resp[s].cpu().numpy(), layer[s].cpu().numpy()), want, f"thr {thr} pair {p}")
    with pytest.raises(M.MoFREAKError):
        ctx.detect_pairs(d[5:], d, W, H, n, kps, offs, threshold=0, capacity=cap, row_stride=stride, pair_stride=fstride)
    with pytest.raises(M.MoFREAKError):
        ctx.detect_pairs(d[5:], d, W, H, n, kps, offs, octaves=5, capacity=cap, row_stride=stride, pair_stride=fstride)


def test_frame_at_a_time_stream_equals_the_stack_calls(ctx, oracle):
    """mofreak_stream_*: the reference's loop shape (one decoded frame per iteration against a queue of the last five).
    Rows per pushed frame, concatenated, equal the whole-stack calls -- with the detector on gray frames, and with
    caller keypoints on BGR frames (BGR2GRAY happens on the way into the ring)."""
    fr = synth.moving_objects_stack(11, 256, 192, seed=41)
    want = ctx.compute_stream_host(fr)
    with ctx.open_stream(256, 192, use_detector=True) as st:
        parts = [st.push(f) for f in fr]
        assert st.frames == len(fr)
    assert all(len(p) == 0 for p in parts[:5]) and sum(len(p) for p in parts) == len(want) > 50
    assert np.concatenate(parts).tobytes() == want.tobytes()
    # BGR frames, keypoints from the caller
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (9, 120, 160, 3), dtype=np.uint8)
    bgr[:, 30:90, 40:120] //= 4
    gray = ctx.bgr_to_gray_host(bgr)
    per_frame = [synth.random_keypoints(rng, 40, 160, 120, sizes=(7.0, 9.0, 12.0)) for _ in range(9)]
    offs = np.concatenate([[0], np.cumsum([len(k) for k in per_frame[5:]])]).astype(np.int64)
    want = oracle.Freak().extract_stream(gray, np.concatenate(per_frame[5:]), offs)
    with ctx.open_stream(160, 120, use_detector=False) as st:
        parts = [st.push(bgr[t], per_frame[t]) for t in range(9)]
    assert np.concatenate(parts).tobytes() == want.tobytes() and len(want) > 20


@pytest.mark.parametrize("w,h", [(7, 7), (6, 6), (33, 9), (9, 40), (53, 47), (2, 2)])
def test_tiny_images_and_every_depth(ctx, w, h):
    """Layers that shrink to nothing (no scored region, zero-sized planes) must neither crash nor differ from the oracle."""
    img = np.random.default_rng(w * 100 + h).integers(0, 256, (h, w), dtype=np.uint8)
    for octaves in (0, 1, 3, 4):
        kps, offs, resp, layer = ctx.detect_pairs_host(img, None, 30, octaves)
        _assert_same_keypoints((kps, resp, layer), _oracle_keypoints(img, 30, octaves), f"{w}x{h} octaves {octaves}")


def test_other_frame_gaps(native_lib, oracle):
    """gap_for_frame_difference is a parameter (the reference hard-codes 5, MoFREAKUtilities.cpp:378): pairing, labels, the
    frame ring and the whole-loop call all follow it."""
    fr = synth.moving_objects_stack(9, 224, 160, seed=51)
    for gap in (1, 3):
        c = M.Context(0, gap_for_frame_difference=gap)
        T = len(fr)
        lists = [_oracle_keypoints(O.absdiff(fr[t], fr[t - gap]))[0] for t in range(gap, T)]
        offs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
        want = oracle.Freak().extract_stream(fr, np.concatenate(lists), offs, gap=gap)
        assert c.compute_stream_host(fr).tobytes() == want.tobytes() and len(want) > 50
        with c.open_stream(224, 160) as st:
            got = np.concatenate([st.push(f) for f in fr])
        assert got.tobytes() == want.tobytes()
        assert want["frame_number"].min() == gap - 1
        c.close()


def test_tie_chains_when_the_waiting_list_overflows():
    """det_tie_kernel keeps a pair's ties (6144) and a layer's waiting ties (2048) in LDS lists, reads the ties from global
    memory and scans the whole layer when a list overflows.  The debug library (libmofreak_hip_debug.so) is the same source
    with lists of 64 and 8 (and 16 chunks of candidates per prologue thread instead of 512), so tie-heavy images take the
    fallback paths on every layer; it also decides every tie both on whole rows and cell by cell and reports a difference:
    same keypoints as the oracle."""
    import subprocess
    from mofreak_amd import build
    build.build_native(debug=True)  # rebuilt whenever a source is newer than it (same check as the product build)
    here = os.path.dirname(os.path.abspath(__file__))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mofreak_amd as M
import test_detector_gpu as T
assert M.api.load().mofreak_build_flags() == 1, "not the debug build"
with M.Context(0) as ctx:
    for seed, levels, block, thr in ((1, 4, 3, 30), (5, 4, 1, 30), (7, 3, 2, 25)):
        img = T._quantised_noise(seed, 300, 400, levels, block)
        kps, offs, resp, layer = ctx.detect_pairs_host(img, None, thr, 3)
        T._assert_same_keypoints((kps, resp, layer), T._oracle_keypoints(img, thr, 3), "seed %%d" %% seed)
        assert len(kps) > 100
print("overflow path ok")
""" % (os.path.dirname(here), here)
    env = dict(os.environ, MOFREAK_HIP_LIBRARY=build.DEBUG_LIB_PATH)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "overflow path ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_growing_frame_sizes_in_one_context_keep_the_tie_maps_clean():
    """The tie logic's two bookkeeping maps are all zero between calls: every call takes back the bytes it set, and a map
    that has to grow is cleared (a grown buffer can come back from the allocator at its old address, stale bytes behind
    its old end: the fuzz tool found that once in some ten thousand calls).  The debug library checks at the start of
    every detector call that both maps are zero and fails the call if not: a sequence of calls whose frames and pair
    counts grow and shrink, other calls in between, against the oracle."""
    import subprocess
    from mofreak_amd import build
    build.build_native(debug=True)
    here = os.path.dirname(os.path.abspath(__file__))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mofreak_amd as M
import oracle_lib as O
from mofreak_amd import synth
assert M.api.load().mofreak_build_flags() == 1, "not the debug build"
rng = np.random.default_rng(11)
with M.Context(0) as ctx:
    for i, (W, H, n_pairs, octaves) in enumerate([(160, 120, 1, 3), (320, 240, 2, 3), (161, 120, 1, 2), (640, 360, 1, 3), (320, 240, 3, 1), (640, 480, 2, 3),
                                                  (160, 120, 3, 0), (736, 480, 1, 3), (640, 360, 3, 3), (1024, 576, 1, 3)]):
        fr = synth.moving_objects_stack(5 + n_pairs, W, H, seed=100 + i)
        k, offs, resp, layer = ctx.detect_pairs_host(fr[5:], fr[:n_pairs], 30, octaves)
        for p in range(n_pairs):
            want = O.brisk_detect(O.absdiff(fr[5 + p], fr[p]), 30, octaves)
            wk = np.stack([want["x"], want["y"], want["size"]], 1).astype(np.float32).reshape(-1, 3)
            assert k[offs[p]:offs[p + 1]].tobytes() == wk.tobytes() and resp[offs[p]:offs[p + 1]].tobytes() == want["response"].tobytes(), (i, p)
        frames = rng.integers(0, 256, (6, 97 + 40 * i, 131 + 60 * i), dtype=np.uint8)  # the staging buffers move as well
        ctx.extract_pairs_host(frames[5:], frames[:1], synth.random_keypoints(rng, 50, frames.shape[2], frames.shape[1], sizes=(8.4, 12.0)))
print("maps stay clean")
""" % (os.path.dirname(here), here)
    env = dict(os.environ, MOFREAK_HIP_LIBRARY=build.DEBUG_LIB_PATH)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "maps stay clean" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_detector_in_the_pipelined_routes(ctx, oracle, tmp_path):
    """The reference's own keypoint source where the fast routes are: mofreak_compute_clips (many clips, one pipelined pass, the
    detector window by window), chunk pushes into a detector stream, and run_dataset's batched mode with the "brisk" provider --
    all the rows of one mofreak_compute_stream call per clip (themselves checked against the oracle elsewhere in this file)."""
    import torch
    from mofreak_amd import harness
    clips = [synth.moving_objects_stack(T, 256, 192, seed=60 + i) for i, T in enumerate((9, 23, 4, 14, 6, 31))]
    want = [ctx.compute_stream_host(c) for c in clips]
    assert sum(len(w) for w in want) > 400 and len(want[2]) == 0
    # one pass, small windows (a clip spans several, windows span several clips)
    for chunk in (0, 9, 13):
        rows, offs, n_kp = ctx.compute_clips(clips, chunk_frames=chunk)
        assert n_kp >= len(rows) > 0
        for i, w in enumerate(want):
            assert rows[offs[i]:offs[i + 1]].tobytes() == w.tobytes(), (chunk, i)
    # rows left on the device, too small a buffer reported with the size a retry needs
    buf = torch.empty(sum(len(w) for w in want) * 32, dtype=torch.uint8, device="cuda")
    n, offs, _ = ctx.compute_clips(clips, rows_out=buf, chunk_frames=11)
    assert n == sum(len(w) for w in want) and buf.cpu().numpy().view(M.ROW_DTYPE).tobytes() == np.concatenate(want).tobytes()
    with pytest.raises(M.MoFREAKError) as e:
        ctx.compute_clips(clips, rows_out=np.zeros(10, M.ROW_DTYPE))
    assert e.value.code == M.api.ERR_CAPACITY
    # the oracle on one of them, end to end
    f = oracle.Freak()
    lists = [_oracle_keypoints(O.absdiff(clips[3][t], clips[3][t - 5]))[0] for t in range(5, len(clips[3]))]
    offs3 = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
    assert want[3].tobytes() == f.extract_stream(clips[3], np.concatenate(lists), offs3).tobytes()
    # chunk pushes of any length into a detector stream, mixed with single pushes
    long = np.concatenate([clips[1], clips[5]])
    whole = ctx.compute_stream_host(long)
    with ctx.open_stream(256, 192, use_detector=True) as st:
        parts = [st.push_frames(long[:3]), st.push(long[3]), st.push_frames(long[4:19], chunk_frames=8), st.push_frames(long[19:20]), st.push_frames(long[20:])]
        assert st.frames == len(long)
    assert np.concatenate(parts).tobytes() == whole.tobytes()
    # the dataset walk: batched, rows kept; and every rank writing its files from device-made text
    names = [f"clip{i}.avi" for i in range(len(clips))]
    mo = harness.MoFREAKUtilities(harness.HMDB51, device=0, keypoint_provider="brisk")
    try:
        res = harness.run_dataset(clips, names, str(tmp_path / "root"), mo, batch_bytes=3 << 20)
        kept = {i: r.copy() for i, r in res["rows_per_video"].items()}
        res2 = harness.run_dataset(clips, names, str(tmp_path / "ranks"), mo, batch_bytes=3 << 20, keep_rows=False, write="ranks")
    finally:
        mo.close()
    assert res["batched"] and res["rounds"] > 1 and res2["write"] == "ranks"
    for i, w in enumerate(want):
        assert kept[i].tobytes() == w.tobytes(), i
        text = M.format_rows(w)
        assert open(tmp_path / "root" / (names[i] + ".mofreak"), "rb").read() == text
        assert open(tmp_path / "ranks" / (names[i] + ".mofreak"), "rb").read() == text
